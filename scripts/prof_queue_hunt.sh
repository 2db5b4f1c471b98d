#!/bin/bash
# Hunt for the intermittent slow C2 search: run bench.py under the kernel trace several times and print, for the last
# step of each run, when every DP kernel started and how long it ran.
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/run$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $OUT/run$i.log 2>&1
  f=$(ls $OUT/run$i/*/*kernel_trace.csv | head -1)
  python3 - $f $OUT/run$i.log <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "opal_dp" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-3:]
t0 = min(int(r["Start_Timestamp"]) for r in last)
line = [l for l in open(sys.argv[2]) if l.startswith("{")]
km = json.loads(line[-1])["roofline"]["kernel_ms"] if line else None
print("kernel_ms", km, " | ".join(f"S={r['Kernel_Name'].split(',')[3].split('>')[0].strip()} q{r['Queue_Id']} start {int(r['Start_Timestamp']) - t0} dur {int(r['End_Timestamp']) - int(r['Start_Timestamp'])}" for r in last))
PY
  rm -rf $OUT/run$i
done
