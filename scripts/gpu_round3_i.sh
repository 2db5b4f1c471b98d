#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $OUT/pytest_gpu.log
python3 scripts/gpu_api_e2e.py > $OUT/e2e.txt 2>&1
OPAL_HIP_DEBUG=1 python3 scripts/gpu_api_e2e.py 2>&1 | grep "order:\|fill: order\|rejected\|opalSearchDatabase:" | tail -8 > $OUT/e2e_debug.txt
OPAL_HIP_DEBUG=1 python3 bench.py --workload sprot --query-fasta tests/golden/data/query_O74807.fasta --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | grep "launch\|one launch" | tail -8 > $OUT/sprot_q110_plan.txt
