#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5f; mkdir -p $O
timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('driver-style', d['value'], d['ms_per_step'], 'kernel', r['kernel_ms'], 'frac', r['frac'], 'clock', r['clock_ghz'], 'e2e', d.get('e2e',{}).get('ms'), d.get('e2e',{}).get('same_database_ms'))"
timeout 300 python3 scripts/gpu_api_e2e.py 2>/dev/null | grep searchType
timeout 900 python3 scripts/cold/cold_process.py > $O/cold_process.txt 2>&1; cat $O/cold_process.txt
OPAL_HIP_DEBUG=1 NSEQ=625000 QLEN=513 timeout 300 python3 scripts/gpu_api_e2e.py > $O/e2e_c4_debug.txt 2>&1; grep -n "order on the device\|fill: order\|slice\|shard:\|opalSearchDatabase:" $O/e2e_c4_debug.txt | sed -n 20,50p
timeout 2400 python3 scripts/gpu_planner_sweep.py > $O/planner_full.txt 2>&1; tail -40 $O/planner_full.txt
