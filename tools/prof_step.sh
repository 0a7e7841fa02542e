#!/bin/bash
# Runs on the GPU box: rocprofv3 --kernel-trace --stats of 3 replayed steps -> per-step kernel summary + top-40 stats in gpurun_out/$1/
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $R/bench.py --steps 3 --warmup 4 --no-cpu-baseline --no-secondary > /tmp/p1.log 2>&1 || { tail -20 /tmp/p1.log; exit 1; }
T=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1); S=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1)
python3 $R/tools/prof_summary.py $T --top 200 > $O/step_kernel_summary.txt
head -41 $S > $O/kernel_stats_top40.csv
head -60 $O/step_kernel_summary.txt | cut -c1-200
