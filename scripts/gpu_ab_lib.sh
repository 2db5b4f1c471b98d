#!/bin/bash
# A/B of two builds of the library on the same box: the in-tree one against build/ab/libopal.so (OPAL_LIB_PATH), interleaved;
# then the drop-in call with and without the zero-copy search, and the whole GPU suite.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
line() {
  python3 bench.py --no-e2e --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('lib=${OPAL_LIB_PATH:-tree} $*', 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['achieved'], 'frac', r['frac'], 'clock', r.get('clock_ghz'), 'verified', d['verified'])"
}
for r in 1 2 3; do
  line --workload c2 --steps 100 --warmup 10
  OPAL_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/libopal.so line --workload c2 --steps 100 --warmup 10
done
for r in 1 2; do
echo "zero copy:"; python3 scripts/gpu_api_e2e.py 2>&1 | grep opalSearch
echo "upload:"; OPAL_HIP_NO_ZERO_COPY=1 python3 scripts/gpu_api_e2e.py 2>&1 | grep opalSearch
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
