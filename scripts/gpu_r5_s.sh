#!/bin/bash
# same-box A/B of the SW end-cell code: the library as it was at the start of the day (build_abl/libopal_r5start.so) against the
# current one, alternating; then the randomized differential tests with many seeds
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5s
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
for round in 1 2; do
  for v in old new; do
    if [ $v = old ]; then export OPAL_LIB_PATH=$PWD/build_abl/libopal_r5start.so; else unset OPAL_LIB_PATH; fi
    echo "== $v (round $round)"
    OPAL_HIP_STREAM=0 ST=1 MODE=SW timeout 300 python3 scripts/gpu_sprot_queries.py 2>&1 | grep "^SW" | sed "s/^/$v class  /"
    ST=1 MODE=SW timeout 300 python3 scripts/gpu_sprot_queries.py 2>&1 | grep "^SW" | sed "s/^/$v natural /"
  done
done > $OUT/ab_sprot_end.txt 2>&1
unset OPAL_LIB_PATH
cat $OUT/ab_sprot_end.txt
OPAL_TEST_SEEDS=500 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "randomized_differential" > $OUT/stress_seeds500.log 2>&1
tail -3 $OUT/stress_seeds500.log
