#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for how in serial alternate separate; do
  echo "== streams $how"; OPAL_HIP_SLICE_STREAMS=$how python3 scripts/gpu_api_e2e.py 2>&1 | tail -2
  for s in "1,2,3,4" "1,4" "1,2,2,3,4"; do echo "streams $how slices $s"; OPAL_HIP_SLICE_STREAMS=$how OPAL_HIP_SLICES=$s python3 scripts/gpu_api_e2e.py 2>&1 | tail -2; done
done > $OUT/e2e_streams.txt 2>&1
OPAL_HIP_DEBUG=1 python3 scripts/gpu_api_e2e.py 2>&1 | grep -v "plan width\|one launch" | tail -40 > $OUT/e2e_debug.txt
