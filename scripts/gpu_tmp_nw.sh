#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/nwq
OPAL_HIP_DEBUG=1 python3 bench.py --workload c5 --total-seqs 100000 --no-e2e --no-cpu-baseline --steps 1 --warmup 1 --qlen 3000 --mode NW --search-type 1 2>&1 | grep -E "opal\]" | grep -v "chunk plan" | head -30
OPAL_HIP_DEBUG=1 python3 bench.py --workload c5 --total-seqs 100000 --no-e2e --no-cpu-baseline --steps 1 --warmup 1 --qlen 3000 --mode NW --search-type 1 2>&1 | grep -E "chunk plan" | sort -u | head -40
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/nwq/trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload c5 --total-seqs 100000 --no-e2e --no-cpu-baseline --steps 3 --warmup 1 --qlen 3000 --mode NW --search-type 1 > /dev/null 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/nwq/trace/*/*kernel_stats.csv | cut -c1-200 | head
find $GRAFT_REPO_ROOT/gpurun_out/nwq -name "*kernel_trace.csv" -delete
