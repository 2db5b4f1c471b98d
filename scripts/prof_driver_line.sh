#!/bin/bash
# rocprofv3 --kernel-trace --stats of the driver's own command (python3 bench.py --steps 20 --warmup 5): the kernel statistics and,
# from the trace, the average duration and the average period of the dominant kernel's launches, beside the line the run printed.
# Usage (GPU box): scripts/prof_driver_line.sh <outdir-under-gpurun_out>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-also > $OUT/bench_line.json 2> $OUT/bench.err
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
python3 $GRAFT_REPO_ROOT/scripts/trace_period.py $OUT/trace > $OUT/periods.txt
find $OUT/trace -name "*.csv" -size +2M -delete
cat $OUT/periods.txt; head -4 $OUT/kernel_stats.csv | cut -c1-200
python3 -c "
import json
d=json.loads(open('$OUT/bench_line.json').read().strip().splitlines()[-1]); r=d['roofline']
print('line: value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'launch_ms', r['launch_ms'], 'frac', r['frac'], 'frac_per_launch', r['frac_per_launch'])"
