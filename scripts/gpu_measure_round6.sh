#!/bin/bash
# Round-6 measurement session on one box: parity of what changed last, PMC / trace profiles of every configuration, bench lines,
# the published-shape queries. Usage (GPU box, repo root): scripts/gpu_measure_round6.sh <outdir-under-gpurun_out>
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/$1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_pair.py -x -q -k "global or stream or pipeline or three_sequences or result_reuse or pair or every_wave" > $OUT/pytest_subset.log 2>&1
grep -v RCCL $OUT/pytest_subset.log | tail -3
scripts/bench_configs.sh > $OUT/bench_configs.txt 2> $OUT/bench_configs.err
for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_$i.json 2> $OUT/bench_driver_$i.err; done
timeout 300 python3 scripts/gpu_sprot_queries.py > $OUT/sprot_queries.txt 2>&1
for m in SW HW OV NW; do ST=1 MODE=$m timeout 300 python3 scripts/gpu_sprot_queries.py; done > $OUT/sprot_queries_end.txt 2>&1
for Q in 1000 2000; do for m in SW HW NW OV; do ST=1 MODE=$m REPS=3 timeout 600 python3 scripts/gpu_sprot_queries.py $Q; done; done > $OUT/sprot_long_queries.txt 2>&1
scripts/prof_round6.sh $1/prof > $OUT/prof.log 2>&1
ls $OUT $OUT/prof | head -80
