#!/bin/bash
# Round 3: after the one-byte residue layout and the pipelined opalSearchDatabase (gather/upload beside the search).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $OUT/pytest_gpu.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver.json 2> $OUT/bench_driver.err
OPAL_HIP_DEBUG=1 python3 scripts/gpu_api_e2e.py > $OUT/e2e_debug.txt 2>&1
for s in "2,3,5" "1,2,4,8" "3,6" "4" "17"; do echo "slices $s"; OPAL_HIP_SLICES=$s python3 scripts/gpu_api_e2e.py 2>&1 | tail -3; done > $OUT/e2e_slices.txt 2>&1
echo "no slices"; OPAL_HIP_NO_SLICES=1 python3 scripts/gpu_api_e2e.py 2>&1 | tail -3 >> $OUT/e2e_slices.txt
echo "sync fill"; OPAL_HIP_SYNC_FILL=1 python3 scripts/gpu_api_e2e.py 2>&1 | tail -3 >> $OUT/e2e_slices.txt
ls -la $OUT
