#!/bin/bash
# Round 5, final measurement pass on the GPU box (run from the repo root through gpurun): what profiles/r05_* holds.
#   scripts/gpu_measure_round5.sh [with-tests]
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5final
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ "$1" = with-tests ]; then timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $OUT/pytest_gpu.log; fi
if [ "$1" = with-tests ]; then OPAL_TEST_SEEDS=2000 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k randomized_differential_round5 2>&1 | tail -4 > $OUT/stress_seeds2000.log; fi
timeout 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
timeout 900 python3 bench.py --workload sprot > $OUT/bench_sprot.json 2> $OUT/bench_sprot.err
timeout 300 python3 scripts/gpu_sprot_queries.py > $OUT/sprot_queries.txt 2>&1
OPAL_HIP_SWAP=0 timeout 300 python3 scripts/gpu_sprot_queries.py O74807 P19930 > $OUT/sprot_queries_noswap.txt 2>&1
for m in SW HW OV NW; do ST=1 MODE=$m timeout 300 python3 scripts/gpu_sprot_queries.py; done > $OUT/sprot_queries_end.txt 2>&1
for m in SW HW OV NW; do OPAL_HIP_SWAP=0 ST=1 MODE=$m REPS=3 timeout 300 python3 scripts/gpu_sprot_queries.py O74807 P19930 P18080; done > $OUT/sprot_queries_end_noswap.txt 2>&1
timeout 300 python3 scripts/gpu_api_e2e.py > $OUT/e2e.txt 2>&1
timeout 1500 bash scripts/bench_configs.sh > $OUT/bench_configs.txt 2> $OUT/bench_configs.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_steady -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > $OUT/c2_steady.json 2> $OUT/c2_steady.err )
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_driver -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $OUT/c2_driver_prof.json 2> $OUT/c2_driver_prof.err )
for Q in 1000 2000; do for m in SW HW NW OV; do ST=1 MODE=$m REPS=3 timeout 600 python3 scripts/gpu_sprot_queries.py $Q; done; done > $OUT/sprot_long_queries.txt 2>&1
timeout 600 python3 scripts/cold/cold_process.py > $OUT/cold_process.txt 2>&1
find $OUT -name "*kernel_trace.csv" -size +4M -delete
ls -la $OUT
