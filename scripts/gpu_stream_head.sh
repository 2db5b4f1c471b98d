#!/bin/bash
# Databases with a long tail: the longest sequences on the class body as the first workgroups of the streaming launch
# (OPAL_HIP_NO_STREAM_HEAD=1: class kernels only), interleaved processes.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
line() {
  python3 bench.py --workload sprot --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('sprot nohead=$OPAL_HIP_NO_STREAM_HEAD blocks=$OPAL_HIP_STREAM_BLOCKS value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['achieved'], 'frac', r['frac'], 'verified', d['verified'], ' '.join('%s %.3f' % (k, v['gpu_kernel_ms']) for k,v in d['published_shape'].items()))"
}
for r in 1 2 3; do
  line
  OPAL_HIP_STREAM_BLOCKS=256 line
  OPAL_HIP_NO_STREAM_HEAD=1 line
done
