#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5x
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "ties" > $OUT/ties.log 2>&1
tail -8 $OUT/ties.log | cut -c1-250
