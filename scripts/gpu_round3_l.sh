#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3n
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 scripts/gpu_api_e2e.py > $OUT/e2e.txt 2>&1
NSEQ=625000 QLEN=513 python3 scripts/gpu_api_e2e.py > $OUT/e2e_625k.txt 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest_gpu.log
for w in "--workload c2" "--workload c4 --total-seqs 625000" "--workload c5 --total-seqs 100000"; do python3 bench.py $w --no-cpu-baseline --steps 10 --warmup 2 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], d['roofline']['kernel_ms'], d['e2e'])"; done > $OUT/bench_e2e.txt 2>&1
