#!/bin/bash
# Final check of round 6 on one box: the GPU suite, the planner sweep on the final tree, the randomized differential tests on 4 000
# seeds, the suite again with fallback paths and the queue's other modes switched on, a default and a driver-style bench line.
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r6final; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; grep -v RCCL $OUT/pytest_gpu.log | tail -3
python3 scripts/gpu_planner_sweep.py > $OUT/planner_sweep.txt 2> $OUT/planner_sweep.err; tail -20 $OUT/planner_sweep.txt | cut -c1-160
OPAL_TEST_SEEDS=4000 timeout 2400 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "randomized_differential" > $OUT/stress_seeds4000.log 2>&1; tail -3 $OUT/stress_seeds4000.log | cut -c1-200
OPAL_HIP_PIPELINE=0 OPAL_HIP_NO_UBER=1 OPAL_HIP_NO_ZERO_COPY=1 OPAL_HIP_NO_STREAM_HEAD=1 OPAL_HIP_PAIR=0 timeout 1200 python3 -m pytest tests -m gpu -q --timeout 600 -k "not overlap_and_keep" 2>&1 | grep -v RCCL | tail -8 > $OUT/toggles_a.log
OPAL_HIP_PIPELINE=2 OPAL_HIP_SWAP=0 OPAL_HIP_WAVES_PER_BLOCK=8 timeout 1200 python3 -m pytest tests -m gpu -q --timeout 600 -k "not swap and not overlap_and_keep" 2>&1 | grep -v RCCL | tail -8 > $OUT/toggles_b.log
OPAL_HIP_PIPE_PRIO=1 OPAL_HIP_NO_STREAM_CHUNKS=1 OPAL_HIP_WAVES_PER_BLOCK=2 OPAL_HIP_DEVICE_ORDER=0 timeout 1200 python3 -m pytest tests -m gpu -q --timeout 600 -k "not row_chunks" 2>&1 | grep -v RCCL | tail -8 > $OUT/toggles_c.log
for f in $OUT/toggles_*.log; do echo "== $f"; tail -6 $f | cut -c1-220; done
/usr/bin/time -v python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; grep -E "Elapsed|Maximum resident" $OUT/bench_default.err
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python3 -c "
import json
for f in ('bench_default','bench_driver'):
    d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], r['kernel_ms'], r['launch_ms'], r['frac'], d['verified'], d['e2e']['ms'], d['also']['c4']['value'], d['also']['c5']['value'])
"
