#!/bin/bash
# cold opalSearchDatabase call on C2 under the scheduler's toggles (device ordering on), two rounds interleaved
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5b; mkdir -p $O
run() { echo "# $*" >> $O/variants.txt; env "$@" timeout 300 python3 scripts/gpu_api_e2e.py 2>/dev/null | grep searchType >> $O/variants.txt; }
for r in 1 2; do
  run A=default
  run OPAL_HIP_NO_ZERO_COPY=1
  run OPAL_HIP_NO_SLICES=1
  run OPAL_HIP_SLICES=1,2,4
  run OPAL_HIP_SLICES=2,6
  run OPAL_HIP_SLICES=1,2,3,4
  run OPAL_HIP_NO_ZERO_COPY=1 OPAL_HIP_SLICES=1,2,3,4
  run OPAL_HOST_THREADS=32
  run OPAL_HOST_THREADS=16
done
cat $O/variants.txt
