"""Which kernels follow a device memset in a rocprofv3 kernel trace?  (memset nodes are not reliable under graph replay on
this stack — see DESIGN.md; every such site must go.)  usage: memset_sites.py <trace dir> [marker substring]"""
import csv, sys, glob, os, collections
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "mas_kernel"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
lo, hi = (idx[-2], idx[-1]) if len(idx) >= 2 else (0, len(rows))
cnt = collections.Counter()
prev = collections.Counter()
for i in range(lo, hi):
    if "fillBuffer" in rows[i]["Kernel_Name"] and i + 1 < hi:
        n = rows[i + 1]
        p = rows[i - 1]
        cnt[(n["Kernel_Name"][:140], n["Grid_Size_X"], n["Grid_Size_Y"], n["Workgroup_Size_X"])] += 1
        prev[p["Kernel_Name"][:100]] += 1
print("kernels in step:", hi - lo, "memsets:", sum(cnt.values()))
for k, v in cnt.most_common():
    print(f"{v:4d}  grid={k[1]},{k[2]} wg={k[3]}  {k[0]}")
print("--- kernels right before the memsets")
for k, v in prev.most_common(25):
    print(f"{v:4d}  {k}")
