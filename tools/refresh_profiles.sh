#!/bin/bash
# Runs on the GPU box: regenerates the judged measurement artefacts into gpurun_out/refresh/ (copied to profiles/ afterwards).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
echo "bench done: $(cut -c1-160 $O/bench.json)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $R/bench.py --steps 3 --warmup 4 --no-cpu-baseline > /tmp/p1.log 2>&1
T=$(find /tmp/p1 -name "*kernel_trace.csv" | head -1); S=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1)
python3 $R/tools/prof_summary.py $T > $O/step_kernel_summary.txt
head -41 $S > $O/kernel_stats_top40.csv
echo "trace done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p2 -- python3 $R/bench.py --eager --steps 2 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
echo "fetch pass done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p3 -- python3 $R/bench.py --eager --steps 2 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
echo "write pass done"
python3 $R/tools/pmc_traffic.py $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $(find /tmp/p3 -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json $O/pmc_fetch_write.txt > /dev/null
echo "pmc done"
