#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5w
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
OPAL_TEST_SEEDS=4000 timeout 2400 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "randomized_differential" > $OUT/stress_seeds4000.log 2>&1
tail -4 $OUT/stress_seeds4000.log | cut -c1-250
