#!/bin/bash
# Row chunks of queries beyond one wave: parity of every chunk path, then the 3000-aa lines (streaming chunks by default,
# OPAL_HIP_NO_STREAM_CHUNKS=1 = the class kernels' chunks).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_longtail.py -q -x -k "chunks or longer_than or large_alphabet or longtail or long_tail" 2>&1 | tail -15
line() {
  python3 bench.py --workload c5 --total-seqs 100000 --no-e2e --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$* nostream=$OPAL_HIP_NO_STREAM_CHUNKS', 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['achieved'], 'frac', r['frac'], 'verified', d['verified'])"
}
for m in NW HW OV; do line --qlen 3000 --mode $m --search-type 1; OPAL_HIP_NO_STREAM_CHUNKS=1 line --qlen 3000 --mode $m --search-type 1; done
line --qlen 6000 --mode NW --search-type 1
line --qlen 2600 --mode OV --search-type 1
OPAL_HIP_DEBUG=1 python3 bench.py --workload c5 --total-seqs 100000 --no-e2e --no-cpu-baseline --steps 1 --warmup 0 --qlen 3000 --mode NW --search-type 1 2>&1 | grep -E "opal\] (streaming|chunk [0-9])" | head -12
