#!/bin/bash
# round 5, second half of the final pass: SW + end on the published shape with the final plan rules, PMC profiles of the configs whose
# kernels changed, the full planner sweep, the whole GPU suite, bench.py's drop-in call with a single OpenMP thread
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5r
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
ST=1 MODE=SW timeout 300 python3 scripts/gpu_sprot_queries.py > $OUT/sprot_queries_end_sw.txt 2>&1
OMP_NUM_THREADS=1 timeout 300 python3 bench.py --no-cpu-baseline > $OUT/bench_omp1.json 2> $OUT/bench_omp1.err
timeout 300 python3 bench.py --no-cpu-baseline > $OUT/bench_omp_default.json 2> $OUT/bench_omp_default.err
timeout 1500 python3 -m pytest tests -q -m gpu --timeout 600 > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
timeout 900 python3 scripts/gpu_planner_sweep.py > $OUT/planner_sweep.txt 2>&1
tail -25 $OUT/planner_sweep.txt | cut -c1-250
timeout 900 bash scripts/prof_round5.sh r5r/prof c2 > $OUT/prof_c2.log 2>&1
QUICK=1 timeout 1200 bash scripts/prof_round5.sh r5r/prof c2end c5end sprotend > $OUT/prof_quick.log 2>&1
ls $OUT $OUT/prof
