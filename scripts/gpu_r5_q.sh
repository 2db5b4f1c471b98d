#!/bin/bash
mkdir -p gpurun_out/r5q
export PYTHONUNBUFFERED=1
TAG=new timeout 600 python3 scripts/gpu_r5_abl.py > gpurun_out/r5q/abl.txt 2>&1
grep "^new" gpurun_out/r5q/abl.txt
timeout 1500 python3 -m pytest tests -q -m gpu --timeout 600 > gpurun_out/r5q/pytest_gpu.log 2>&1
tail -5 gpurun_out/r5q/pytest_gpu.log
