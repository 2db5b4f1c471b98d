#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5g; mkdir -p $O
timeout 600 python3 scripts/gpu_r5_mismatch2.py 2>&1 | grep "^n=" | head -6
timeout 900 python3 scripts/gpu_r5_f4.py > $O/f4.txt 2>&1; grep -v "plan width\|amdgpu.ids" $O/f4.txt | cut -c1-250
NS=2000,20000 QS=1700,2000,2600,4100 timeout 1500 python3 scripts/gpu_planner_sweep.py > $O/sweep_partial.txt 2>&1; grep "MISS\|MISMATCH\|^# " $O/sweep_partial.txt | cut -c1-250
