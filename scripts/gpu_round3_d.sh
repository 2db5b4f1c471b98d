#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
OPAL_HIP_DEBUG=1 python3 scripts/gpu_api_e2e.py 2>&1 | grep -v "plan width" | tail -60 > $OUT/e2e_debug.txt
python3 scripts/gpu_api_e2e.py > $OUT/e2e.txt 2>&1
for s in "1,2,4" "3,6" "4" "2,2,3,4"; do echo "slices $s"; OPAL_HIP_SLICES=$s python3 scripts/gpu_api_e2e.py 2>&1 | tail -2; done > $OUT/e2e_slices.txt 2>&1
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "api or reuse or dropin or empty or golden" 2>&1 | tail -5 > $OUT/pytest_subset.log
for t in "0,5" "1,5" "1,6" "1,4" "1,3" "2,5" "2,6" "2,4"; do echo "tune $t"; OPAL_HIP_TUNE=$t python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-e2e | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['clock_ghz'], d['verified'])"; done > $OUT/tune.txt 2>&1
OPAL_HIP_TUNE=1,5 python3 scripts/gpu_timeline.py > $OUT/timeline_c2_tune15.txt 2>&1
OPAL_HIP_TUNE=2,5 python3 scripts/gpu_timeline.py > $OUT/timeline_c2_tune25.txt 2>&1
OPAL_HIP_TUNE=1,5 UNIFORM=350 python3 scripts/gpu_timeline.py > $OUT/timeline_u350_tune15.txt 2>&1
