#!/usr/bin/env python3
"""HBM traffic per C-ABI launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of
`bench.py --eager`).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 1/2 of the bytes of wide
coalesced reads, so read bytes = 2*FETCH_SIZE*1024; WRITE_SIZE is in KB.
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <out.txt>"""
import collections, csv, json, sys

# entry point -> (kernels that count as one launch each, helper kernels whose bytes are added to those launches)
GROUPS = {"vits_conv1d_cl": (("conv1d_cl_kernel", "conv1d_flat_kernel", "conv1d_ring_kernel"), ()),
          "vits_conv1d_cl_wgrad": (("wgrad_kernel",), ("reduce_slabs", "reduce_pending_kernel")),
          "vits_mas_f32": (("mas_kernel",), ("mas_zero_kernel",))}


def load(path, counter):
    per = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"_comment": "HBM traffic per launch from rocprofv3 PMC passes (see the .txt beside this file): bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, "
                   "FETCH doubled per the gfx950 correction; mean over all dispatches of the entry point's kernels in bench.py --eager"}
lines = ["# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, bench.py --eager), per-dispatch means in KB.",
         "# gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads => read bytes = 2*FETCH_SIZE*1024."]
for name, ctr in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
    for k, v in sorted(ctr.items(), key=lambda kv: -len(kv[1])):
        if any(p in k for pats in GROUPS.values() for grp in pats for p in grp):
            lines.append(f"{name:12s} n={len(v):6d} mean={sum(v)/len(v):14.2f} sum={sum(v):16.1f}  {k[:110]}")
for entry, (pats, extra) in GROUPS.items():
    fv = [x for k, v in fetch.items() if any(p in k for p in pats) for x in v]
    wv = [x for k, v in write.items() if any(p in k for p in pats) for x in v]
    fx = sum(x for k, v in fetch.items() if any(p in k for p in extra) for x in v)
    wx = sum(x for k, v in write.items() if any(p in k for p in extra) for x in v)
    if fv and wv:
        fm, wm = (sum(fv) + fx) / len(fv), (sum(wv) + wx) / len(wv)
        out[entry] = {"dispatches": len(fv), "fetch_kb_mean": round(fm, 2), "write_kb_mean": round(wm, 2),
                      "bytes_per_launch": int((2 * fm + wm) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
open(sys.argv[4], "w").write("\n".join(lines) + "\n")
print(json.dumps(out, indent=1))
