#!/bin/bash
# How often does a streaming-kernel parity test fail? (the slot a sequence lands in depends on timing: a rare interleaving shows as a
# rare failure). Usage: scripts/gpu_flake_hunt.sh <out> <pytest -k expression> <runs> <variants: tree or build/<name> ...>
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/$1; K=$2; N=$3; shift; shift; shift
for V in "$@"; do
  if [ $V = tree ]; then unset OPAL_LIB_PATH; else export OPAL_LIB_PATH=$GRAFT_REPO_ROOT/build/$V/libopal.so; fi
  fails=0
  for i in $(seq 1 $N); do
    timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "$K" > /tmp/flake.log 2>&1 || { fails=$((fails+1)); grep -m3 "Mismatched\|stream ties\|Max absolute" /tmp/flake.log >> $OUT.details; }
  done
  echo "$V: $fails failures in $N runs of -k '$K'" | tee -a $OUT
done
