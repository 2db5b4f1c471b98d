#!/bin/bash
# Databases with a long tail: the longest sequences on the class kernels beside the streamed bulk (OPAL_HIP_NO_STREAM_HEAD=1: off).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for b in "" 240 224 208; do
  OPAL_HIP_STREAM_BLOCKS=$b python3 bench.py --workload sprot --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('sprot blocks=$b value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['achieved'], 'frac', r['frac'], 'verified', d['verified'], ' '.join('%s %.3f' % (k, v['gpu_kernel_ms']) for k,v in d['published_shape'].items()))"
done
OPAL_HIP_NO_STREAM_HEAD=1 python3 bench.py --workload sprot --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('sprot nohead value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['achieved'], 'frac', r['frac'], 'verified', d['verified'], ' '.join('%s %.3f' % (k, v['gpu_kernel_ms']) for k,v in d['published_shape'].items()))"
