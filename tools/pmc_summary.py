#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel name.  usage: pmc_summary.py <counter_collection.csv> [substring ...]"""
import collections, csv, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
pats = sys.argv[2:]
for name, ctrs in sorted(rows.items(), key=lambda kv: -sum(len(v) for v in kv[1].values())):
    if pats and not any(p in name for p in pats):
        continue
    for c, vals in ctrs.items():
        print(f"{c:12s} n={len(vals):6d} mean={sum(vals)/len(vals):14.2f} sum={sum(vals):16.1f}  {name[:100]}")
