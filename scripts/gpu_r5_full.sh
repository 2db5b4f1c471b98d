#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5k; mkdir -p $O
bash scripts/gpu_r5_l.sh > $O/e2e_pre.log 2>&1; cat $O/e2e_pre.log | tail -22
timeout 3400 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log | cut -c1-300
