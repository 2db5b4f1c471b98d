#!/bin/bash
# ablation of the SW + end code: the shipped library, then the three measurement builds
mkdir -p gpurun_out/r5m
{
TAG=base python3 scripts/gpu_r5_abl.py
for n in 1 2 3; do TAG=abl$n OPAL_LIB_PATH=$PWD/build_abl/libopal_abl$n.so python3 scripts/gpu_r5_abl.py; done
TAG=base python3 scripts/gpu_r5_abl.py
} > gpurun_out/r5m/abl.txt 2>&1
tail -60 gpurun_out/r5m/abl.txt
