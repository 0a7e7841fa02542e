#!/bin/bash
# Runs on the GPU box: the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only with --kernel-trace) over two
# eager steps, reduced to per-launch HBM traffic by tools/pmc_traffic.py into gpurun_out/refresh/.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p2 -- python3 $R/tools/pmc_step.py > $O/pmc_fetch.log 2>&1
echo "fetch pass done"
timeout -k 10 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p3 -- python3 $R/tools/pmc_step.py > $O/pmc_write.log 2>&1
echo "write pass done"
python3 $R/tools/pmc_traffic.py $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $(find /tmp/p3 -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json $O/pmc_fetch_write.txt
