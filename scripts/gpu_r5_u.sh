#!/bin/bash
# the GPU suite with the fallback paths switched on (as profiles/r04_pytest_gpu_toggles.txt)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5u
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
OPAL_HIP_NO_UBER=1 OPAL_HIP_NO_ZERO_COPY=1 OPAL_HIP_NO_SIDE_PREFIX=1 OPAL_HIP_NO_STREAM_HEAD=1 OPAL_HIP_PAIR=0 timeout 1200 python3 -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -12 > $OUT/toggles_a.log
OPAL_HIP_SWAP=0 OPAL_HIP_WAVES_PER_BLOCK=8 timeout 1200 python3 -m pytest tests -m gpu -q --timeout 600 -k "not swap" 2>&1 | tail -12 > $OUT/toggles_b.log
OPAL_HIP_NO_STREAM_CHUNKS=1 OPAL_HIP_WAVES_PER_BLOCK=2 OPAL_HIP_DEVICE_ORDER=0 timeout 1200 python3 -m pytest tests -m gpu -q --timeout 600 -k "not row_chunks" 2>&1 | tail -12 > $OUT/toggles_c.log
for f in $OUT/toggles_*.log; do echo "== $f"; tail -8 $f | cut -c1-220; done
