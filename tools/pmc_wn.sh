#!/bin/bash
# Runs on the GPU box: SQ counters of the one-launch WaveNet layer kernels on the step's shape (rocprofv3 --pmc with --kernel-trace only).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmcwn}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
            "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_BUSY_CYCLES"; do
  rm -rf /tmp/pm3
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm3 -- python3 $R/tools/ubench_wn.py 16 500 192 5 4 > $O/pmc.log 2>&1
  C=$(find /tmp/pm3 -name "*counter_collection.csv" | head -1)
  python3 - "$C" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    if "wn_layer" not in r["Kernel_Name"] and "wgrad_batch" not in r["Kernel_Name"]:
        continue
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, d in agg.items():
    n = d.get("SQ_WAVE_CYCLES", 1)
    print(k, "launches", cnt[k])
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v / max(cnt[k],1):16.0f} per launch ({v / n:6.3f} of SQ_WAVE_CYCLES)")
PY
done
