#!/bin/bash
# Width-18 kernels (2-byte profile entries) after a change of their row blocks: parity of what runs on them,
# then the C5 lines (streaming kernels at width 18) and the chunked 3000-aa lines.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "stream or narrow or chunks or w18 or 18" 2>&1 | tail -4
line() {
  python3 bench.py --no-e2e --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$*', 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['achieved'], 'frac', r['frac'], 'clock', r.get('clock_ghz'), 'verified', d['verified'])"
}
line --workload c5 --total-seqs 100000
line --workload c5 --total-seqs 100000 --search-type 1
for m in NW HW OV; do line --workload c5 --total-seqs 100000 --search-type 1 --mode $m; done
for m in SW NW OV; do line --workload c5 --total-seqs 100000 --qlen 3000 --search-type 1 --mode $m; done
line --workload c5 --total-seqs 100000 --qlen 2500
