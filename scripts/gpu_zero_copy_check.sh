#!/bin/bash
# Zero-copy search of a new database (kernels read the pinned staging buffer and write the device copy) against the
# upload path, on the plain opalSearchDatabase entry; then the parity suites that go through that entry.
OUT=$GRAFT_REPO_ROOT/gpurun_out/zc
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for r in 1 2; do
python3 scripts/gpu_api_e2e.py > $OUT/e2e_zero_copy_$r.txt 2>&1
OPAL_HIP_NO_ZERO_COPY=1 python3 scripts/gpu_api_e2e.py > $OUT/e2e_upload_$r.txt 2>&1
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $OUT/pytest_gpu.log
