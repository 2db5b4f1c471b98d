#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5d; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "api or reuse or ragged or ordering or context" > $O/api_tests.log 2>&1; tail -3 $O/api_tests.log
timeout 1200 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "same_database or context_reuse or c2_score" > $O/full_tests.log 2>&1; tail -3 $O/full_tests.log
run() { echo "# $*" >> $O/variants.txt; env "$@" timeout 300 python3 scripts/gpu_api_e2e.py 2>/dev/null | grep searchType >> $O/variants.txt; }
for r in 1 2; do
  run A=default
  run OPAL_HIP_NO_ZERO_COPY=1
  run OPAL_HOST_THREADS=32
  run OPAL_HOST_THREADS=20
done
echo "# NSEQ=625000 QLEN=513" >> $O/variants.txt
NSEQ=625000 QLEN=513 timeout 300 python3 scripts/gpu_api_e2e.py 2>/dev/null | grep searchType >> $O/variants.txt
OPAL_HIP_DEBUG=1 timeout 300 python3 scripts/gpu_api_e2e.py > $O/e2e_debug.txt 2>&1
cat $O/variants.txt; grep "fill: order\|order on the device" $O/e2e_debug.txt | sed -n 3,6p
timeout 600 python3 scripts/gpu_clock_timeline.py > $O/clock_timeline.txt 2>&1; cat $O/clock_timeline.txt
