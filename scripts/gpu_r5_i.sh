#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5i; mkdir -p $O
timeout 1800 python3 -m pytest tests/test_gpu_pair.py -x -q -m gpu > $O/tests_pair.log 2>&1; tail -5 $O/tests_pair.log | cut -c1-250
SHAPES=sprot NS=100000,541762 QS=300,513,1000,2000 MODES=HW,NW,OV TYPES=1 timeout 1500 python3 scripts/gpu_planner_sweep.py > $O/sweep_pair.txt 2>&1; grep -v "^#  " $O/sweep_pair.txt | cut -c1-300
for P in 0 -1; do
 echo "OPAL_HIP_PAIR=$P"
 if [ $P = 0 ]; then export OPAL_HIP_PAIR=0; else unset OPAL_HIP_PAIR; fi
 python3 scripts/gpu_r5_pairprobe.py
done
