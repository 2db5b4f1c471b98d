#!/bin/bash
# One 30-step bench line per configuration (value, ms per step, kernel GCUPS, fraction of the VALU-issue bound at 2.4 GHz and
# at the measured clock, verified; for the lines run with the drop-in call: its end-to-end time).
# Run on the GPU box from the repo root: scripts/bench_configs.sh > gpurun_out/bench_configs.txt
run() {
  python3 bench.py "$@" --no-cpu-baseline --steps 30 --warmup 5 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; e=d.get('e2e')
print('$*'.ljust(70), 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['achieved'], 'frac', r['frac'], 'clock', r.get('clock_ghz'), 'frac@clock', r.get('frac_at_measured_clock'), 'verified', d['verified'], ('e2e ms %s (same database again %s)' % (e['ms'], e.get('same_database_ms'))) if e else '')"
}
run --workload c2
run --workload c2 --gap-open 3 --gap-ext 1 --no-e2e
run --workload c2 --seqs 1000000 --no-e2e
run --workload c3 --mode SW
run --workload c3 --mode NW
run --workload c3 --mode HW
run --workload c3 --mode OV
run --workload c4 --total-seqs 625000
run --workload c5 --total-seqs 100000
run --workload c5 --total-seqs 100000 --search-type 1 --no-e2e
run --workload c5 --total-seqs 100000 --search-type 1 --mode NW --no-e2e
run --workload c5 --total-seqs 100000 --search-type 1 --mode HW --no-e2e
run --workload c5 --total-seqs 100000 --search-type 1 --mode OV --no-e2e
run --workload c5 --total-seqs 625000
run --workload c5 --total-seqs 100000 --qlen 1000 --no-e2e
run --workload c5 --total-seqs 100000 --qlen 3000 --no-e2e
run --workload c5 --total-seqs 100000 --qlen 3000 --search-type 1 --no-e2e
run --workload c5 --total-seqs 100000 --qlen 3000 --search-type 1 --mode NW --no-e2e
run --workload c5 --total-seqs 100000 --qlen 6000 --no-e2e
run --workload sprot --no-e2e
run --workload c3 --mode SW --gap-open 3 --no-e2e
run --workload c5x3 --no-e2e
