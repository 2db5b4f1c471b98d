#!/bin/bash
# Round-6 profiles: rocprofv3 kernel stats and SQ / HBM PMC passes of every BASELINE.json config.
# Run on the GPU box from the repo root: scripts/prof_round6.sh <outdir-under-gpurun_out> [configs...]
# The program goes directly after `--` (no env/bash hop: the profiler initialises the GPU before exec).
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
CFGS=${@:-"c2 c2end c2end31 c3nw c3hw c3ov c4 c4end c5 c5end c5nw c5hw c5ov c5x3 w32"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
COMMON="--no-cpu-baseline --no-e2e --no-also --steps 5 --warmup 2 --settle-ms 0"  # (no settle phase: summarize_prof.py counts 12 launches per run)
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
SQ2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
for c in $CFGS; do
  unset OPAL_HIP_FORCE_WIDTH
  case $c in
    c2)    ARGS="--workload c2" ;;
    c2end) ARGS="--workload c3 --mode SW" ;;
    c3nw)  ARGS="--workload c3 --mode NW" ;;
    c3hw)  ARGS="--workload c3 --mode HW" ;;
    c3ov)  ARGS="--workload c3 --mode OV" ;;
    c4)    ARGS="--workload c4 --total-seqs 625000" ;;
    c5)    ARGS="--workload c5 --total-seqs 100000" ;;
    c5end) ARGS="--workload c5 --total-seqs 100000 --search-type 1" ;;
    c5nw)  ARGS="--workload c5 --total-seqs 100000 --search-type 1 --mode NW" ;;
    c5hw)  ARGS="--workload c5 --total-seqs 100000 --search-type 1 --mode HW" ;;
    c5ov)  ARGS="--workload c5 --total-seqs 100000 --search-type 1 --mode OV" ;;
    w32)   ARGS="--workload c2 --seqs 100000 --steps 3"; export OPAL_HIP_FORCE_WIDTH=32 ;;
    c5x3)  ARGS="--workload c5x3" ;;
    c2end31) ARGS="--workload c3 --mode SW --gap-open 3" ;;
    q3000) ARGS="--workload c5 --total-seqs 100000 --qlen 3000" ;;
    q3000nw) ARGS="--workload c5 --total-seqs 100000 --qlen 3000 --mode NW --search-type 1" ;;
    q3000end) ARGS="--workload c5 --total-seqs 100000 --qlen 3000 --search-type 1" ;;
    sprot) ARGS="--workload sprot" ;;
    sprotend) ARGS="--workload sprot --search-type 1" ;;
    c4end) ARGS="--workload c4 --total-seqs 625000 --search-type 1" ;;
    sprot110) ARGS="--workload sprot --query-fasta $GRAFT_REPO_ROOT/tests/golden/data/query_O74807.fasta" ;;
    sprot195) ARGS="--workload sprot --query-fasta $GRAFT_REPO_ROOT/tests/golden/data/query_P19930.fasta" ;;
    sprot390) ARGS="--workload sprot --query-fasta $GRAFT_REPO_ROOT/tests/golden/data/query_Q3ZAI3.fasta" ;;
  esac
  D=$OUT/$c; mkdir -p $D
  timeout 600 python3 $B $ARGS $COMMON > $D/bench.json 2> $D/bench.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $B $ARGS $COMMON > $D/trace.log 2>&1
  if [ -z "$QUICK" ]; then :; fi
  timeout 600 rocprofv3 --pmc $SQ1 --output-format csv -d $D/sq1 -- python3 $B $ARGS $COMMON > $D/sq1.log 2>&1
  timeout 600 rocprofv3 --pmc $SQ2 --output-format csv -d $D/sq2 -- python3 $B $ARGS $COMMON > $D/sq2.log 2>&1
  if [ -z "$QUICK" ]; then
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $D/grbm -- python3 $B $ARGS $COMMON > $D/grbm.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 $B $ARGS $COMMON > $D/fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 $B $ARGS $COMMON > $D/write.log 2>&1
  fi
  python3 $GRAFT_REPO_ROOT/scripts/summarize_prof.py $D $OUT/r06_$c.json $c > $D/summary.log 2>&1
  python3 $GRAFT_REPO_ROOT/scripts/trace_period.py $D/trace >> $OUT/r06_trace_periods.txt 2>&1
  cp $(ls $D/trace/*/*kernel_stats.csv | head -1) $OUT/r06_${c}_kernel_stats.csv 2>/dev/null
  # keep the merged output small: the raw per-dispatch CSVs can be tens of MB
  find $D -name "*_counter_collection.csv" -size +2M -delete
  find $D -name "*kernel_trace.csv" -size +2M -delete
done
ls -la $OUT
