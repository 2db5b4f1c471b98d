#!/bin/bash
# One 30-step bench line per configuration (value, ms per step, kernel GCUPS, fraction of the VALU-issue bound, verified).
# Run on the GPU box from the repo root: scripts/bench_configs.sh > gpurun_out/bench_configs.txt
run() {
  python3 bench.py "$@" --no-cpu-baseline --no-e2e --steps 30 --warmup 5 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$*'.ljust(62), 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['achieved'], 'frac', r['frac'], 'verified', d['verified'])"
}
run --workload c2
run --workload c2 --gap-open 3 --gap-ext 1
run --workload c2 --seqs 1000000
run --workload c3 --mode SW
run --workload c3 --mode NW
run --workload c3 --mode HW
run --workload c3 --mode OV
run --workload c4 --total-seqs 625000
run --workload c5 --total-seqs 100000
run --workload c5 --total-seqs 100000 --search-type 1
run --workload c5 --total-seqs 100000 --search-type 1 --mode NW
run --workload c5 --total-seqs 100000 --search-type 1 --mode HW
run --workload c5 --total-seqs 100000 --search-type 1 --mode OV
run --workload c5 --total-seqs 100000 --qlen 1000
run --workload c5 --total-seqs 100000 --qlen 3000
