#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5h; mkdir -p $O
timeout 900 python3 scripts/gpu_r5_f4.py > $O/f4.txt 2>&1; grep "kernel ms" $O/f4.txt | cut -c1-200
WHAT=f4 OPAL_HIP_NO_STARTED_WAIT=1 timeout 900 python3 scripts/gpu_r5_f4.py 2>&1 | grep "Q=300.*{}: kernel ms" | sed 's/^/no started-wait: /'
NS=2000,20000 QS=1700,2000,2600,4100 timeout 1500 python3 scripts/gpu_planner_sweep.py > $O/sweep_partial.txt 2>&1; grep "MISS\|MISMATCH\|^# " $O/sweep_partial.txt | cut -c1-250
SHAPES=sprot NS=100000,541762 QS=300,1000,1400,1700 timeout 1500 python3 scripts/gpu_planner_sweep.py > $O/sweep_sprot.txt 2>&1; grep -v "^#  " $O/sweep_sprot.txt | cut -c1-250
for r in 1 2; do timeout 300 python3 scripts/gpu_api_e2e.py 2>/dev/null | grep searchType; NSEQ=625000 QLEN=513 timeout 300 python3 scripts/gpu_api_e2e.py 2>/dev/null | grep searchType; done
timeout 2400 python3 -m pytest tests/test_gpu_swap.py tests/test_gpu_longtail.py -x -q -m gpu > $O/tests_swap_tail.log 2>&1; tail -3 $O/tests_swap_tail.log
timeout 2400 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream" > $O/tests_stream.log 2>&1; tail -3 $O/tests_stream.log
