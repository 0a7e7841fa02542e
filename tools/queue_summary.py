#!/usr/bin/env python3
"""Per hardware queue: launches and busy time in the steady-state window of a rocprofv3 kernel trace (how the runtime mapped a
replayed hipGraph's branches onto queues).  python tools/queue_summary.py kernel_trace.csv [steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady-state window: the last `steps` occurrences of the alignment kernel delimit the steps
marks = [i for i, r in enumerate(rows) if "mas_kernel" in r["Kernel_Name"]]
lo = marks[-steps - 1] if len(marks) > steps else 0
hi = marks[-1]
win = rows[lo:hi]
t0, t1 = int(win[0]["Start_Timestamp"]), int(win[-1]["End_Timestamp"])
print(f"window: {len(win)} launches, {(t1 - t0) / 1e6 / steps:.3f} ms/step, {len(win) / steps:.0f} launches/step")
q = collections.defaultdict(lambda: [0, 0])
for r in win:
    k = (r.get("Queue_Id"), r.get("Stream_Id"))
    q[k][0] += 1; q[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, busy) in sorted(q.items(), key=lambda kv: -kv[1][1]):
    print(f"  queue {k[0]} stream {k[1]}: {n / steps:7.0f} launches/step  {busy / 1e6 / steps:7.3f} ms busy/step")
# gaps on the busiest queue
bq = max(q, key=lambda k: q[k][1])
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in win if (r.get("Queue_Id"), r.get("Stream_Id")) == bq]
gaps = [b[0] - a[1] for a, b in zip(ev, ev[1:])]
import statistics
print(f"  busiest queue: median gap {statistics.median(gaps) / 1e3:.2f} us, mean {statistics.mean(gaps) / 1e3:.2f} us, gaps > 20 us: {sum(g > 20000 for g in gaps) / steps:.0f}/step totalling {sum(g for g in gaps if g > 20000) / 1e6 / steps:.3f} ms/step")
# the kernels of each queue
import re
for k in sorted(q, key=lambda k: -q[k][1]):
    agg = collections.defaultdict(lambda: [0, 0])
    for r in win:
        if (r.get("Queue_Id"), r.get("Stream_Id")) == k:
            name = re.sub(r"^void |\(anonymous namespace\)::", "", r["Kernel_Name"])[:110]
            agg[name][0] += 1; agg[name][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print(f"-- queue {k[0]} stream {k[1]}: top kernels (launches/step, ms/step, avg us)")
    for name, (n, busy) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"   {n / steps:6.0f} {busy / 1e6 / steps:7.3f} {busy / 1e3 / n:7.1f}  {name}")
