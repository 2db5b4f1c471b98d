#!/bin/bash
# the randomized differential tests (100 seeds) and the ties tests on the build in the tree
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5x
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
OPAL_TEST_SEEDS=100 timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "ties or randomized" > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log | cut -c1-200
