#!/bin/bash
# localise a hang: every step under its own timeout, unbuffered, names before results
mkdir -p gpurun_out/r5o
export PYTHONUNBUFFERED=1
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/r5o/$name.log 2>&1; echo "$name rc=$?" | tee -a gpurun_out/r5o/summary.txt; }
run ties 400 python3 -m pytest tests/test_gpu_parity.py -v -x -m gpu -k "ties_in_the_streaming"
run streamsw 400 python3 -m pytest tests/test_gpu_parity.py -v -x -m gpu -k "streaming_sw_score_kernels"
run chunks 400 python3 -m pytest tests/test_gpu_parity.py -v -x -m gpu -k "row_chunks"
CASES=2 TAG=new run abl_natural 200 python3 scripts/gpu_r5_abl.py
run classties 300 python3 -m pytest tests/test_gpu_parity.py -v -x -m gpu -k "ties_in_every_wave"
run parity_rest 600 python3 -m pytest tests/test_gpu_parity.py -v -x -m gpu -k "not ties and not streaming_sw_score and not row_chunks"
run longtail 400 python3 -m pytest tests/test_gpu_longtail.py -v -x -m gpu
TAG=new run abl 300 python3 scripts/gpu_r5_abl.py
for f in gpurun_out/r5o/*.log; do echo "== $f"; tail -4 $f | cut -c1-250; done
