#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $OUT/pytest_gpu.log
bash scripts/bench_configs.sh > $OUT/bench_configs.txt 2> $OUT/bench_configs.err
OPAL_HIP_DEBUG=1 python3 scripts/gpu_api_e2e.py 2>&1 | grep -v "plan width\|one launch" | tail -30 > $OUT/e2e_debug.txt
