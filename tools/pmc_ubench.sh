#!/bin/bash
# Runs on the GPU box: SQ counters of the ring kernel on one microbench shape (rocprofv3 --pmc with --kernel-trace only).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm1
UB_ONLY=${2:-P11.L5} VITS_RING=2 timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pm1 -- python3 $R/tools/ubench_conv.py > $O/pmc.log 2>&1
C=$(find /tmp/pm1 -name "*counter_collection.csv" | head -1)
python3 - "$C" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k, d in agg.items():
    if "ring" in k or "conv1d" in k:
        n = d.get("SQ_WAVE_CYCLES", 1)
        print(k)
        for c, v in sorted(d.items()):
            print(f"   {c:28s} {v:16.0f}  ({v / n:6.3f} of SQ_WAVE_CYCLES)")
PY
