#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_longtail.py -q -x -k "chunks or longer_than or large_alphabet or longtail or long_tail" 2>&1 | tail -5
line() {
  python3 bench.py --workload c5 --total-seqs 100000 --no-e2e --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$* noside=$OPAL_HIP_NO_SIDE_PREFIX', 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['achieved'], 'frac', r['frac'], 'verified', d['verified'])"
}
for m in NW HW OV; do line --qlen 3000 --mode $m --search-type 1; OPAL_HIP_NO_SIDE_PREFIX=1 line --qlen 3000 --mode $m --search-type 1; done
line --qlen 6000 --mode NW --search-type 1
