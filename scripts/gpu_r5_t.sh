#!/bin/bash
# final build: the driver-style line once more, and the full C4 / C5 databases (5 M sequences) on ONE GPU
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5t
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
timeout 900 python3 bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_c4_5M.json 2> $OUT/bench_c4_5M.err
timeout 900 python3 bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_c5_5M.json 2> $OUT/bench_c5_5M.err
timeout 900 python3 bench.py --workload c4 --search-type 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_c4_5M_end.json 2> $OUT/bench_c4_5M_end.err
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
tail -2 $OUT/smoke.log
for f in $OUT/*.json; do python3 - "$f" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1].split("/")[-1], "value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "verified", d["verified"], "e2e", (d.get("e2e") or {}).get("ms"))
P
done
