#!/bin/bash
# A/B of two builds of the library on the same box: the in-tree one against build/ab/libopal.so (OPAL_LIB_PATH), interleaved;
# the published shape with and without the in-launch head; the whole GPU suite.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
line() {
  python3 bench.py --no-e2e --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('lib=${OPAL_LIB_PATH:-tree} nohead=$OPAL_HIP_NO_STREAM_HEAD $*', 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['achieved'], 'frac', r['frac'], 'verified', d['verified'], ' '.join('%s %.3f' % (k, v['gpu_kernel_ms']) for k,v in d.get('published_shape', {}).items()))"
}
for r in 1 2; do
  line --workload c4 --total-seqs 625000 --steps 10 --warmup 3
  OPAL_LIB_PATH=$GRAFT_REPO_ROOT/build/ab/libopal.so line --workload c4 --total-seqs 625000 --steps 10 --warmup 3
done
for r in 1 2; do
  line --workload sprot
  OPAL_HIP_NO_STREAM_HEAD=1 line --workload sprot
done
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
