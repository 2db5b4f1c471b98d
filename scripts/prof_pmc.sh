#!/bin/bash
# PMC passes for the DP kernel on the uniform-length database (no tail): run on the GPU box.
# usage: scripts/prof_pmc.sh <outdir-under-gpurun_out> [G]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
G=${2:-16}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export C2=0 UNIFORM=1 GLIST=$G
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/scripts/gpu_perf.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 $GRAFT_REPO_ROOT/scripts/gpu_perf.py > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc2 -- python3 $GRAFT_REPO_ROOT/scripts/gpu_perf.py > $OUT/pmc2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_IFETCH --output-format csv -d $OUT/pmc3 -- python3 $GRAFT_REPO_ROOT/scripts/gpu_perf.py > $OUT/pmc3.log 2>&1
find $OUT -name "*.csv" | head -30
