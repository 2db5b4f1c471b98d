#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "multi_gpu_handle or api_context or two_rank" 2>&1 | tail -5 > $OUT/pytest_subset.log
OPAL_HIP_DEBUG=1 python3 scripts/gpu_api_e2e.py 2>&1 | grep -v "plan width" | tail -70 > $OUT/e2e_debug.txt
python3 scripts/gpu_api_e2e.py > $OUT/e2e.txt 2>&1
for s in "1,16" "1,4" "1,2,4" "2,15" "1,3,5,4" "1,8"; do echo "slices $s"; OPAL_HIP_SLICES=$s python3 scripts/gpu_api_e2e.py 2>&1 | tail -2; done > $OUT/e2e_slices.txt 2>&1
echo "no slices" >> $OUT/e2e_slices.txt; OPAL_HIP_NO_SLICES=1 python3 scripts/gpu_api_e2e.py 2>&1 | tail -2 >> $OUT/e2e_slices.txt
for g in "" 16 4; do echo "lanes '$g'"; OPAL_HIP_LANES_PER_SEQ=$g python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['clock_ghz'], d['verified'])"; done > $OUT/lanes.txt 2>&1
