#!/bin/bash
# the last build of the round: the whole GPU suite, smoke(), the default and the driver-style bench lines
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5z
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -x -q -m gpu --timeout 600 > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
for f in $OUT/bench_*.json; do python3 - "$f" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1].split("/")[-1], "value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], "verified", d["verified"], "e2e", (d.get("e2e") or {}).get("ms"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
P
done
