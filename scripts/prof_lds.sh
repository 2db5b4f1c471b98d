#!/bin/bash
# One PMC pass (LDS cycles and bank conflicts, wave waits) per configuration and library build, summarised per kernel:
#   scripts/prof_lds.sh <outfile-under-gpurun_out> "<configs>" <variant names...>      (tree = the in-tree library)
# The program goes directly after `--` (no env/bash hop: the profiler initialises the GPU before exec).
OUTF=$GRAFT_REPO_ROOT/gpurun_out/$1; CFGS=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
COMMON="--no-cpu-baseline --no-e2e --steps 5 --warmup 2 --settle-ms 0"
PMC="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES"
for c in $CFGS; do
  case $c in
    c2)    ARGS="--workload c2" ;;
    c2end) ARGS="--workload c3 --mode SW" ;;
    c3nw)  ARGS="--workload c3 --mode NW" ;;
    c3hw)  ARGS="--workload c3 --mode HW" ;;
    c3ov)  ARGS="--workload c3 --mode OV" ;;
    c4)    ARGS="--workload c4 --total-seqs 625000" ;;
    c4end) ARGS="--workload c4 --total-seqs 625000 --search-type 1" ;;
    c5)    ARGS="--workload c5 --total-seqs 100000" ;;
    c5end) ARGS="--workload c5 --total-seqs 100000 --search-type 1" ;;
    c5nw)  ARGS="--workload c5 --total-seqs 100000 --search-type 1 --mode NW" ;;
    c5hw)  ARGS="--workload c5 --total-seqs 100000 --search-type 1 --mode HW" ;;
    c5ov)  ARGS="--workload c5 --total-seqs 100000 --search-type 1 --mode OV" ;;
  esac
  for V in "$@"; do
    if [ $V = tree ]; then unset OPAL_LIB_PATH; else export OPAL_LIB_PATH=$GRAFT_REPO_ROOT/build/$V/libopal.so; fi
    D=/tmp/prof_lds/$c/$V; rm -rf $D; mkdir -p $D
    timeout 600 rocprofv3 --pmc $PMC --output-format csv -d $D -- python3 $B $ARGS $COMMON > $D/log 2>&1
    python3 - $D $c $V <<'PY' >> $OUTF
import collections, csv, glob, sys
d, c, v = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "opal_dp" in k or "opal_sw_stream" in k or "opal_gl_stream" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    e = {n: sum(x) / len(x) for n, x in cs.items()}
    if not e.get("SQ_INSTS_LDS"): continue
    print(f"{c:6s} {v:8s} {k[5:48]:43s} launches {max(len(x) for x in cs.values()):3d}  LDS cycles/instr {e['SQ_LDS_IDX_ACTIVE'] / e['SQ_INSTS_LDS']:5.2f}  conflict share {e['SQ_LDS_BANK_CONFLICT'] / max(e['SQ_LDS_IDX_ACTIVE'], 1):.3f}"
          f"  WAIT_ANY {e['SQ_WAIT_ANY'] / e['SQ_WAVE_CYCLES']:.3f}  WAIT_INST_ANY {e['SQ_WAIT_INST_ANY'] / e['SQ_WAVE_CYCLES']:.3f}  VALU instr {e['SQ_INSTS_VALU']:.4g}  busy {e['SQ_BUSY_CYCLES']:.4g}")
PY
  done
done
cat $OUTF
