#!/bin/bash
mkdir -p gpurun_out/r5p
export PYTHONUNBUFFERED=1
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/r5p/$name.log 2>&1; echo "$name rc=$?" | tee -a gpurun_out/r5p/summary.txt; }
run swap 600 python3 -m pytest tests/test_gpu_swap.py -v -x -m gpu
run fullsize 1500 python3 -m pytest tests/test_gpu_fullsize.py -v -m gpu --timeout 600
run pair_planner 900 python3 -m pytest tests/test_gpu_pair.py tests/test_gpu_planner.py -v -m gpu --timeout 600
for f in gpurun_out/r5p/*.log; do echo "== $f"; grep -E "FAILED|Timeout|passed|failed" $f | tail -8 | cut -c1-250; done
