#!/bin/bash
# round 5, first GPU call: the new tests, then the cold drop-in call with the ordering on the host / on the device
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5a; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "device_ordering or where_the_ordering" > $O/order_tests.log 2>&1; tail -5 $O/order_tests.log
timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "every_sequence_vs_reference or sprot_shape_full_size" > $O/ref_tests.log 2>&1; tail -8 $O/ref_tests.log
for V in 0 1 0 1; do
  echo "# OPAL_HIP_DEVICE_ORDER=$V" >> $O/e2e.txt
  OPAL_HIP_DEVICE_ORDER=$V timeout 300 python3 scripts/gpu_api_e2e.py >> $O/e2e.txt 2>&1
done
echo "# NSEQ=625000 QLEN=513" >> $O/e2e.txt
for V in 0 1; do
  echo "# OPAL_HIP_DEVICE_ORDER=$V" >> $O/e2e.txt
  NSEQ=625000 QLEN=513 OPAL_HIP_DEVICE_ORDER=$V timeout 300 python3 scripts/gpu_api_e2e.py >> $O/e2e.txt 2>&1
done
OPAL_HIP_DEBUG=1 OPAL_HIP_DEVICE_ORDER=1 timeout 300 python3 scripts/gpu_api_e2e.py > $O/e2e_debug.txt 2>&1
cat $O/e2e.txt
grep -n "order on the device" $O/e2e_debug.txt | head -5
