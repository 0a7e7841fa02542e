#!/bin/bash
# Runs on the GPU box: bench line + rocprofv3 --kernel-trace --stats summaries into gpurun_out/refresh/ (copied to profiles/).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
echo "bench done: $(cut -c1-160 $O/bench.json)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $R/bench.py --steps 3 --warmup 4 --no-cpu-baseline --no-secondary > /tmp/p1.log 2>&1
T=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1); S=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1)
python3 $R/tools/prof_summary.py $T > $O/step_kernel_summary.txt
head -41 $S > $O/kernel_stats_top40.csv
echo "trace done"
