#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5e; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "rccl or multi_gpu_handle" > $O/rccl_test.log 2>&1; tail -4 $O/rccl_test.log
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python3 -c "
import json; d=json.load(open('$O/bench_driver.json')); r=d['roofline']; print('driver-style', d['value'], d['ms_per_step'], 'kernel', r['kernel_ms'], 'frac', r['frac'], 'clock', r['clock_ghz'], r['frac_at_measured_clock'], 'e2e', d.get('e2e',{}).get('ms'), d.get('e2e',{}).get('same_database_ms'), 'verified', d['verified'], 'settle', d['settle']['steps'])"
timeout 900 python3 bench.py --steps 20 --warmup 5 --settle-ms 0 --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('no settle   ', d['value'], d['ms_per_step'], 'kernel', r['kernel_ms'], 'frac', r['frac'], 'clock', r['clock_ghz'])"
timeout 900 python3 scripts/cold/cold_process.py > $O/cold_process.txt 2>&1; cat $O/cold_process.txt
timeout 1500 python3 -m pytest tests/test_gpu_planner.py -x -q -m gpu -s > $O/planner_quick.log 2>&1; tail -90 $O/planner_quick.log
