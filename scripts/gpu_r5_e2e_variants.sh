#!/bin/bash
# cold opalSearchDatabase call on C2 under the scheduler's toggles (device ordering on), two rounds interleaved
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5c; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "api or reuse or ragged or where_the_ordering or context" > $O/api_tests.log 2>&1; tail -3 $O/api_tests.log
timeout 1200 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "same_database or context_reuse or c2_" > $O/full_tests.log 2>&1; tail -3 $O/full_tests.log
run() { echo "# $*" >> $O/variants.txt; env "$@" timeout 300 python3 scripts/gpu_api_e2e.py 2>/dev/null | grep searchType >> $O/variants.txt; }
for r in 1 2; do
  run A=default
  run OPAL_HIP_NO_ZERO_COPY=1
  run OPAL_HIP_ZC_FROM_SLICE=1
  run OPAL_HIP_ZC_FROM_SLICE=2
  run OPAL_HIP_SLICES=2,6
  run OPAL_HIP_SLICES=2,6 OPAL_HIP_NO_ZERO_COPY=1
  run OPAL_HIP_SLICES=2,6 OPAL_HIP_ZC_FROM_SLICE=2
  run OPAL_HOST_THREADS=24
  run OPAL_HOST_THREADS=48
done
OPAL_HIP_DEBUG=1 timeout 300 python3 scripts/gpu_api_e2e.py > $O/e2e_debug.txt 2>&1
cat $O/variants.txt; grep "fill: order" $O/e2e_debug.txt | head -4
