#!/bin/bash
# bench.py's N = 1 loop with two searches queued on the handle (default) against one at a time (--in-flight 1): one 30-step line per config.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash scripts/bench_configs.sh > gpurun_out/bench_configs_inflight2.txt 2>/dev/null
run() {
  python3 bench.py "$@" --in-flight 1 --no-cpu-baseline --no-e2e --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('in-flight 1: $*'.ljust(70), 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['achieved'], 'frac', r['frac'], 'clock', r.get('clock_ghz'), 'verified', d['verified'])"
}
run --workload c2
run --workload c3 --mode SW
run --workload c3 --mode NW
run --workload c3 --mode HW
run --workload c3 --mode OV
run --workload c4 --total-seqs 625000
run --workload c5 --total-seqs 100000 --search-type 1 --mode NW
