#!/bin/bash
# Queued searches on two streams (OPAL_HIP_PIPELINE=1, the default) against one stream (=0), interleaved on one box:
#   scripts/gpu_ab_pipeline.sh <outfile-under-gpurun_out>   [CONFIGS="c2 c3sw ..." ROUNDS=2 STEPS=30 MODES="1 0"]
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/$1; shift; mkdir -p $(dirname $OUT)
CONFIGS=${CONFIGS:-"c2 c3sw c3nw c3hw c3ov"}
line() {
  python3 bench.py --no-e2e --no-cpu-baseline --steps ${STEPS:-30} --warmup 5 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('pipeline=%-2s %-6s' % ('$P', '$C'), 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'launch_ms', r.get('launch_ms'), 'frac', r['frac'], 'frac_per_launch', r.get('frac_per_launch'), 'clock', r.get('clock_ghz'), 'verified', d['verified'])"
}
for r in $(seq 1 ${ROUNDS:-2}); do
 for C in $CONFIGS; do
  case $C in
    c2) ARGS="--workload c2" ;;
    c2_31) ARGS="--workload c2 --gap-open 3 --gap-ext 1" ;;
    c2_1m) ARGS="--workload c2 --seqs 1000000" ;;
    c3sw) ARGS="--workload c3 --mode SW" ;;
    c3nw) ARGS="--workload c3 --mode NW" ;;
    c3hw) ARGS="--workload c3 --mode HW" ;;
    c3ov) ARGS="--workload c3 --mode OV" ;;
    c5) ARGS="--workload c5 --total-seqs 100000" ;;
    c5end) ARGS="--workload c5 --total-seqs 100000 --search-type 1" ;;
    c5nw) ARGS="--workload c5 --total-seqs 100000 --search-type 1 --mode NW" ;;
    c5hw) ARGS="--workload c5 --total-seqs 100000 --search-type 1 --mode HW" ;;
    c5ov) ARGS="--workload c5 --total-seqs 100000 --search-type 1 --mode OV" ;;
    c4) ARGS="--workload c4 --total-seqs 625000 --steps 10" ;;
    c4end) ARGS="--workload c4 --total-seqs 625000 --search-type 1 --steps 10" ;;
    sprot) ARGS="--workload sprot --steps 8" ;;
  esac
  for P in ${MODES:-1 0}; do
    export OPAL_HIP_PIPELINE=$P
    line $ARGS
  done
 done
done | tee $OUT
