#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "same_database or api_context or c4_eight" 2>&1 | tail -15 > $OUT/pytest_subset.log
python3 scripts/gpu_api_e2e.py > $OUT/e2e.txt 2>&1
OPAL_HIP_DEBUG=1 python3 scripts/gpu_api_e2e.py 2>&1 | grep "fill:\|opalSearchDatabase:\|shard:" | tail -24 > $OUT/e2e_debug.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest_gpu.log
