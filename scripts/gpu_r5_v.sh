#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
OPAL_HIP_NO_UBER=1 OPAL_HIP_NO_ZERO_COPY=1 OPAL_HIP_NO_SIDE_PREFIX=1 OPAL_HIP_NO_STREAM_HEAD=1 OPAL_HIP_PAIR=0 timeout 900 python3 -m pytest tests/test_gpu_longtail.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -k "longtail_search_score_and_end or sprot_shape" 2>&1 | grep -E "^E  |FAILED|passed|failed|assert" | head -60 > $OUT/a.log
OPAL_HIP_NO_UBER=1 OPAL_HIP_NO_ZERO_COPY=1 OPAL_HIP_NO_SIDE_PREFIX=1 OPAL_HIP_NO_STREAM_HEAD=1 OPAL_HIP_PAIR=0 timeout 900 python3 -m pytest tests -m gpu -q --timeout 600 --co -q 2>/dev/null | tail -2 >> $OUT/a.log
OPAL_HIP_SWAP=0 OPAL_HIP_WAVES_PER_BLOCK=8 timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -k "sprot_shape" 2>&1 | grep -E "^E  |FAILED|passed|failed|assert" | head -40 > $OUT/b.log
cat $OUT/a.log $OUT/b.log | cut -c1-260
