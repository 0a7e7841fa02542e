#!/bin/bash
# Runs on the GPU box: SQ counters of the weight-gradient kernels on the microbench shapes (rocprofv3 --pmc with --kernel-trace only).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmcw}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm2
UBW_ONLY=${2:-} timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pm2 -- python3 $R/tools/ubench_wgrad.py > $O/pmc.log 2>&1
C=$(find /tmp/pm2 -name "*counter_collection.csv" | head -1)
T=$(find /tmp/pm2 -name "*kernel_trace.csv" | head -1)
python3 - "$C" "$T" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if "wgrad" not in r["Kernel_Name"]:
        continue
    k = (r["Kernel_Name"][:70], r["Grid_Size"], r["LDS_Block_Size"] if "LDS_Block_Size" in r else "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    n = d.get("SQ_WAVE_CYCLES", 1)
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v:16.0f}  ({v / n:6.3f} of SQ_WAVE_CYCLES)")
PY
