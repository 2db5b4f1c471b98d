#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5j; mkdir -p $O
timeout 600 python3 scripts/gpu_r5_pairprobe.py 2>&1 | grep "Q=2000\|Q=300"
timeout 1800 python3 -m pytest tests/test_gpu_longtail.py tests/test_gpu_swap.py tests/test_gpu_pair.py -q -m gpu > $O/tests.log 2>&1; tail -5 $O/tests.log
SHAPES=sprot NS=100000 QS=300,513,1000,2000 MODES=HW,OV TYPES=1 timeout 900 python3 scripts/gpu_planner_sweep.py 2>&1 | grep -v "^#  " | cut -c1-280
