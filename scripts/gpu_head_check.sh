#!/bin/bash
# Head-of-round sanity on the GPU box: smoke(), the whole -m gpu suite, the default bench line and the driver-style one.
OUT=$GRAFT_REPO_ROOT/gpurun_out/head
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $OUT/pytest_gpu.log
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
