#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5l; mkdir -p $O
run() { echo "# $*" >> $O/e2e_pre.txt; env "$@" timeout 300 python3 scripts/gpu_api_e2e.py 2>/dev/null | grep searchType >> $O/e2e_pre.txt; }
for r in 1 2; do
  run A=quiet
  run PRE=5
  run PRE=5 OPAL_HIP_AUX_HIGH_PRIO=1
  run PRE=5 OPAL_HIP_NO_ZERO_COPY=1
  run PRE=5 OPAL_HIP_SLICE_STREAMS=serial
done
cat $O/e2e_pre.txt
