#!/bin/bash
# Round 3, final measurement pass on the GPU box (run from the repo root through gpurun): what profiles/r03_* holds.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3final
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $OUT/pytest_gpu.log
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python3 bench.py --workload sprot > $OUT/bench_sprot.json 2> $OUT/bench_sprot.err
python3 scripts/gpu_api_e2e.py > $OUT/e2e.txt 2>&1
bash scripts/bench_configs.sh > $OUT/bench_configs.txt 2> $OUT/bench_configs.err
OPAL_HIP_FORCE_WIDTH=32 python3 bench.py --seqs 100000 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $OUT/bench_w32_100k.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_steady -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > $OUT/c2_steady.json 2> $OUT/c2_steady.err )
find $OUT -name "*kernel_trace.csv" -size +4M -delete
ls -la $OUT
