#!/bin/bash
# Kernel durations of the C2 search under different tail-class policies (rocprofv3 kernel trace).
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
for cfg in default lone50; do
  unset OPAL_HIP_TAIL_MODEL OPAL_HIP_TAIL_FRACTION
  if [ $cfg = lone50 ]; then export OPAL_HIP_TAIL_MODEL=lone OPAL_HIP_TAIL_FRACTION=0.5; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$cfg -- python3 $B --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $OUT/$cfg.log 2>&1
  f=$(ls $OUT/$cfg/*/*kernel_trace.csv | head -1)
  python3 - $f <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "opal_dp" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 7
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print(r["Kernel_Name"][19:38], "grid", r["Grid_Size_X"], "wg", r["Workgroup_Size_X"], "lds", r["LDS_Block_Size"], "start", int(r["Start_Timestamp"]) - t0, "dur", int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
PY
done
