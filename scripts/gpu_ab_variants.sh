#!/bin/bash
# A/B of library builds on one box (scripts/build_variant.sh): the in-tree library and every build/<name>/libopal.so given,
# interleaved, over the C2 / C3 configurations (kernel GCUPS of bench.py, 30 steps).
#   scripts/gpu_ab_variants.sh <outfile-under-gpurun_out> <variant names...>   [CONFIGS="c2 c3sw c3nw c3hw c3ov" ROUNDS=2]
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/$1; shift; mkdir -p $(dirname $OUT)
CONFIGS=${CONFIGS:-"c2 c3sw c3nw c3hw c3ov"}
line() {
  python3 bench.py --no-e2e --no-cpu-baseline --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-10s %-6s' % ('$V', '$C'), 'value', d['value'], 'kernel_ms', r['kernel_ms'], 'kernel_gcups', r['achieved'], 'frac', r['frac'], 'verified', d['verified'])"
}
for r in $(seq 1 ${ROUNDS:-2}); do
 for C in $CONFIGS; do
  case $C in
    c2) ARGS="--workload c2" ;;
    c3sw) ARGS="--workload c3 --mode SW" ;;
    c3nw) ARGS="--workload c3 --mode NW" ;;
    c3hw) ARGS="--workload c3 --mode HW" ;;
    c3ov) ARGS="--workload c3 --mode OV" ;;
    c5) ARGS="--workload c5 --total-seqs 100000" ;;
    c5nw) ARGS="--workload c5 --total-seqs 100000 --search-type 1 --mode NW" ;;
    c5ov) ARGS="--workload c5 --total-seqs 100000 --search-type 1 --mode OV" ;;
    c4) ARGS="--workload c4 --total-seqs 625000 --steps 10" ;;
    c4end) ARGS="--workload c4 --total-seqs 625000 --search-type 1 --steps 10" ;;
    sprotend) ARGS="--workload sprot --search-type 1 --steps 8" ;;
    c3sw31) ARGS="--workload c3 --mode SW --gap-open 3 --gap-ext 1" ;;
    sprot) ARGS="--workload sprot --steps 8" ;;
  esac
  for V in tree "$@"; do
    if [ $V = tree ]; then unset OPAL_LIB_PATH; else export OPAL_LIB_PATH=$GRAFT_REPO_ROOT/build/$V/libopal.so; fi
    line $ARGS
  done
 done
done | tee $OUT
