#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $OUT/pytest_gpu.log
python3 scripts/gpu_api_e2e.py > $OUT/e2e.txt 2>&1
OPAL_HIP_DEBUG=1 python3 scripts/gpu_api_e2e.py 2>&1 | grep "order:\|fill: order\|rejected\|opalSearchDatabase:" | tail -6 > $OUT/e2e_debug.txt
python3 bench.py --workload sprot > $OUT/bench_sprot.json 2> $OUT/bench_sprot.err
python3 bench.py --workload sprot --query-fasta tests/golden/data/query_O74807.fasta --no-cpu-baseline --no-e2e --steps 20 --warmup 3 > $OUT/bench_sprot_q110.json 2>/dev/null
