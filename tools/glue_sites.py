#!/usr/bin/env python3
"""Where do the aten (non-library) device launches of one fine-tune step come from?  CPU-side torch profiler, ops that are
not views, attributed to the innermost frame inside this package (backward ops through their forward op's sequence number)."""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from importlib import import_module
P = import_module("personalized_text-to-speech_amd"); cfgs = import_module("personalized_text-to-speech_amd.configs"); tr = import_module("personalized_text-to-speech_amd.train")
cfg_name, batch_size, t_y_range = cfgs.WORKLOADS["C2"]
hps = cfgs.get(cfg_name)
ft = tr.FineTuner(hps, "cuda:0", amp=True)
batch = tr.synthetic_batch(hps, batch_size, t_y_range, "cuda:0")
for _ in range(2):
    ft.step(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ft.step(batch)
    torch.cuda.synchronize()
PKG = "personalized_text-to-speech_amd"
def site(stack):
    fr = [s for s in stack if PKG in s]
    return fr[0].split(PKG + "/")[-1][:60] if fr else None
launching = lambda e: any(k.device_type is not None for k in e.kernels) if hasattr(e, "kernels") else False
fwd_site = {}
by_site = collections.Counter(); by_site_t = collections.Counter()
events = [e for e in prof.events()]
for e in events:
    if e.sequence_nr is not None and e.sequence_nr >= 0 and e.stack and not e.name.startswith("autograd::engine") and "Backward" not in e.name:
        s = site(e.stack)
        if s and e.sequence_nr not in fwd_site:
            fwd_site[e.sequence_nr] = s
for e in events:
    nk = len(e.kernels) if hasattr(e, "kernels") else 0
    if nk == 0 or not e.name.startswith("aten::"):
        continue
    if e.cpu_children and any(len(getattr(c, "kernels", [])) for c in e.cpu_children):
        continue                                   # count the innermost launching op only
    s = site(e.stack) if e.stack else None
    tag = "fwd"
    if s is None and e.sequence_nr is not None and e.sequence_nr in fwd_site:
        s, tag = fwd_site[e.sequence_nr], "bwd-of"
    if s is None:
        s = "(autograd/other)"
    by_site[(tag, s)] += nk
    by_site_t[(tag, s)] += sum(k.duration for k in e.kernels)
tot = sum(by_site.values())
print("aten launches attributed:", tot)
for (tag, s), c in by_site.most_common(70):
    print(f"{c:5d} {by_site_t[(tag, s)]/1e3:8.3f} ms  {tag:7s} {s}")
