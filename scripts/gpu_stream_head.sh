#!/bin/bash
# Databases with a long tail: the longest sequences on the class body as the first workgroups of the streaming launch
# (OPAL_HIP_NO_STREAM_HEAD=1: class kernels only; OPAL_HIP_STREAM=1 + OPAL_HIP_STREAM_HEAD_ITEMS=n: forced for every query).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
line() {
  python3 bench.py --workload sprot --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('sprot nohead=$OPAL_HIP_NO_STREAM_HEAD force=$OPAL_HIP_STREAM head=$OPAL_HIP_STREAM_HEAD_ITEMS value', d['value'], 'ms/step', d['ms_per_step'], 'verified', d['verified'], ' '.join('%s %.3f' % (k, v['gpu_kernel_ms']) for k,v in d['published_shape'].items()))"
}
line
OPAL_HIP_STREAM=1 OPAL_HIP_STREAM_HEAD_ITEMS=128 line
OPAL_HIP_STREAM=1 OPAL_HIP_STREAM_HEAD_ITEMS=512 line
OPAL_HIP_NO_STREAM_HEAD=1 line
