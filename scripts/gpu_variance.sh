#!/bin/bash
# Run-to-run spread of the searches that put class-kernel launches on a side stream beside persistent streaming launches.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
line() {
  python3 bench.py --workload c5 --total-seqs 100000 --no-e2e --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$* noside=$OPAL_HIP_NO_SIDE_PREFIX', 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['achieved'], 'frac', r['frac'], 'verified', d['verified'])"
}
for r in 1 2 3 4; do for m in NW OV; do line --qlen 3000 --mode $m --search-type 1; done; done
for r in 1 2; do OPAL_HIP_NO_SIDE_PREFIX=1 line --qlen 3000 --mode NW --search-type 1; done
