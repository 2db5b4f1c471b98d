#!/bin/bash
# Round 3, first measurement pass on the GPU box (run from the repo root through gpurun).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/pytest_gpu.log
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python3 bench.py --workload sprot > $OUT/bench_sprot.json 2> $OUT/bench_sprot.err
python3 scripts/gpu_warm_curve.py > $OUT/warm_curve.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_steady -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > $OUT/c2_steady.json 2> $OUT/c2_steady.err
find $OUT -name "*kernel_trace.csv" -size +4M -delete
ls -la $OUT
