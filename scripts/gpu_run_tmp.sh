mkdir -p gpurun_out/r4f
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_longtail.py -x -q -m gpu -k "streaming or stream or c5_long or longtail_search" > gpurun_out/r4f/parity.log 2>&1; tail -3 gpurun_out/r4f/parity.log
CONFIGS="c5nw c5ov c5" ROUNDS=2 scripts/gpu_ab_variants.sh r4f/ab_glstream.txt base
