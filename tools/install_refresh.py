#!/usr/bin/env python3
"""Copy gpurun_out/refresh/* (made on the GPU box by tools/refresh_bench.sh and tools/refresh_pmc.sh) into profiles/."""
import json, os, shutil
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src, dst = os.path.join(R, "gpurun_out", "refresh"), os.path.join(R, "profiles")
for a, b in [("bench.json", "r01_bench_c2_bf16_1gpu.json"), ("kernel_stats_top40.csv", "r01_rocprofv3_kernel_stats_top40.csv"),
             ("step_kernel_summary.txt", "r01_step_kernel_summary_final.txt"), ("pmc_fetch_write.txt", "r01_pmc_fetch_write.txt")]:
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))
p = os.path.join(src, "pmc_traffic.json")
if os.path.exists(p):
    d = json.load(open(p))
    d["_comment"] = ("HBM traffic per launch from rocprofv3 PMC passes (profiles/r01_pmc_fetch_write.txt; tools/refresh_pmc.sh): bytes = "
                     "(2*FETCH_SIZE + WRITE_SIZE)*1024, FETCH doubled per the gfx950 correction; mean over all dispatches of the entry "
                     "point's kernels in two eager steps (tools/pmc_step.py)")
    json.dump(d, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
b = json.load(open(os.path.join(dst, "r01_bench_c2_bf16_1gpu.json")))
print({k: v for k, v in b.items() if k not in ("roofline", "config")})
r = b["roofline"]
print({k: r[k] for k in r if k not in ("all_kernels", "note")})
