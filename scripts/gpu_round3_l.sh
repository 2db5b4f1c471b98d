#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python3 bench.py --workload sprot --no-e2e --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sprot', d['value'], d['roofline']['kernel_ms'], {k:v['gpu_kernel_ms'] for k,v in d['published_shape'].items()})"; done > $OUT/sprot3.txt 2>&1
for m in NW HW NW HW; do python3 bench.py --workload c3 --mode $m --no-e2e --no-cpu-baseline --steps 50 --warmup 5 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['roofline']['kernel_ms'])"; done > $OUT/c3.txt 2>&1
OPAL_HIP_DEBUG=1 python3 bench.py --workload sprot --no-e2e --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep "launch\|streaming" | head -8 > $OUT/sprot_plan.txt
