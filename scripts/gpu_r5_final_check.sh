#!/bin/bash
# the last build of the round: the whole GPU suite, smoke(), the driver-style bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5z
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -x -q -m gpu --timeout 600 > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python3 - $OUT/bench_driver.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("driver-style", "value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "verified", d["verified"], "e2e", (d.get("e2e") or {}).get("ms"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
P
