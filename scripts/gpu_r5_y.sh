#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5y
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 600 python3 bench.py --no-cpu-baseline > $OUT/bench_default_$i.json 2> $OUT/err_$i.txt; done
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver.json 2>> $OUT/err_d.txt
for f in $OUT/bench_*.json; do python3 - "$f" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1].split("/")[-1], "steps", d["steps"], "value", d["value"], "ms", d["ms_per_step"], "kernel_ms", r["kernel_ms"], "clock", r["clock_ghz"], "frac", r["frac"], "e2e", (d.get("e2e") or {}).get("ms"))
P
done
