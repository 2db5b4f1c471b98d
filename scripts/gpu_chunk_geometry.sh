#!/bin/bash
# Queries beyond one wave: parity of the chunk paths, then ms per search of the streaming chunks (SW score) and of the
# class-kernel chunks (SW with end locations, NW, HW, OV).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "chunks or longer_than or large_alphabet" 2>&1 | tail -5
line() {
  python3 bench.py --workload c5 --total-seqs 100000 --no-e2e --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$* width=$OPAL_HIP_CHUNK_WIDTH', 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['achieved'], 'frac', r['frac'], 'verified', d['verified'])"
}
for q in 2600 3000 6000; do line --qlen $q; done
for m in SW NW HW OV; do line --qlen 3000 --mode $m --search-type 1; done
for m in NW HW OV; do line --qlen 3000 --mode $m --search-type 0; done
OPAL_HIP_CHUNK_WIDTH=18 line --qlen 3000 --mode HW --search-type 1
OPAL_HIP_CHUNK_WIDTH=17 line --qlen 3000 --mode NW --search-type 1
OPAL_HIP_CHUNK_WIDTH=17 line --qlen 3000 --mode OV --search-type 1
OPAL_HIP_CHUNK_WIDTH=17 line --qlen 3000 --mode SW --search-type 1
OPAL_HIP_DEBUG=1 python3 bench.py --workload c5 --total-seqs 100000 --no-e2e --no-cpu-baseline --steps 1 --warmup 0 --qlen 3000 --mode NW --search-type 1 2>&1 | grep -E "opal\] chunk [0-9]" | head -12
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
