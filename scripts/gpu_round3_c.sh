#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
OPAL_HIP_DEBUG=1 python3 scripts/gpu_api_e2e.py 2>&1 | grep -v "plan width" | tail -120 > $OUT/e2e_debug.txt
python3 scripts/gpu_api_e2e.py > $OUT/e2e.txt 2>&1
OPAL_HIP_SYNC_FILL=1 python3 scripts/gpu_api_e2e.py > $OUT/e2e_syncfill.txt 2>&1
python3 scripts/gpu_timeline.py > $OUT/timeline_c2.txt 2>&1
UNIFORM=350 python3 scripts/gpu_timeline.py > $OUT/timeline_uniform350.txt 2>&1
UNIFORM=350 NSEQ=32768 python3 scripts/gpu_timeline.py > $OUT/timeline_uniform350_1round.txt 2>&1
python3 bench.py --workload sprot --no-cpu-baseline --no-e2e > $OUT/bench_sprot.json 2> $OUT/bench_sprot.err
