#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5x
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "ties or streaming or row_chunks or head" > $OUT/stream_tests.log 2>&1
tail -3 $OUT/stream_tests.log | cut -c1-250
for c in 11 0 4 14; do CASES=$c TAG=new timeout 300 python3 scripts/gpu_r5_abl.py 2>&1 | grep "^new"; done | tee $OUT/abl.txt
for c in 11 0; do CASES=$c TAG=new timeout 300 python3 scripts/gpu_r5_abl.py 2>&1 | grep "^new"; done | tee -a $OUT/abl.txt
