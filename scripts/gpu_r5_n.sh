#!/bin/bash
mkdir -p gpurun_out/r5n
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_swap.py tests/test_gpu_fullsize.py tests/test_gpu_longtail.py -q -m gpu -x -k "end or ties or sw or SW or every_sequence or stream or sprot" > gpurun_out/r5n/pytest.log 2>&1
tail -15 gpurun_out/r5n/pytest.log
TAG=new python3 scripts/gpu_r5_abl.py > gpurun_out/r5n/abl.txt 2>&1
grep "^new" gpurun_out/r5n/abl.txt
