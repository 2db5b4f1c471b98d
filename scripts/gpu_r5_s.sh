#!/bin/bash
# same-box A/B of the SW end-cell code: the library as it was at the start of the day (build_abl/libopal_r5start.so) against the
# current one, alternating; the whole GPU suite; the randomized differential tests with many seeds
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5s
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
for round in 1 2; do
  for v in old new; do
    if [ $v = old ]; then export OPAL_LIB_PATH=$PWD/build_abl/libopal_r5start.so; else unset OPAL_LIB_PATH; fi
    ST=1 MODE=SW timeout 300 python3 scripts/gpu_sprot_queries.py 2>&1 | grep "^SW" | sed "s/^/$v (round $round) /"
    CASES=11 TAG=$v timeout 200 python3 scripts/gpu_r5_abl.py 2>&1 | grep "^$v"
    CASES=12 TAG=$v timeout 200 python3 scripts/gpu_r5_abl.py 2>&1 | grep "^$v"
    CASES=13 TAG=$v timeout 200 python3 scripts/gpu_r5_abl.py 2>&1 | grep "^$v"
  done
done > $OUT/ab_sprot_end3.txt 2>&1
unset OPAL_LIB_PATH
cat $OUT/ab_sprot_end3.txt | cut -c1-200
timeout 1500 python3 -m pytest tests -q -m gpu --timeout 600 > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log


for m in SW; do ST=1 MODE=$m timeout 300 python3 scripts/gpu_sprot_queries.py; done > $OUT/sprot_queries_end_sw_final.txt 2>&1
grep "^SW" $OUT/sprot_queries_end_sw_final.txt
