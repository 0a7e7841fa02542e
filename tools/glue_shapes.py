#!/usr/bin/env python3
"""aten device launches of one fine-tune step grouped by (op, input shapes): which tensors does the element-wise glue touch?"""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    ft.step(batch)
    torch.cuda.synchronize()
cnt = collections.Counter(); tim = collections.Counter()
for e in prof.events():
    nk = len(getattr(e, "kernels", []))
    if nk == 0 or not e.name.startswith("aten::"):
        continue
    if any(len(getattr(c, "kernels", [])) for c in e.cpu_children):
        continue
    shp = str([s for s in e.input_shapes if s])[:90]
    cnt[(e.name, shp)] += nk; tim[(e.name, shp)] += sum(k.duration for k in e.kernels)
print("launches", sum(cnt.values()), "ms", sum(tim.values()) / 1e3)
byop = collections.Counter(); byop_t = collections.Counter()
for (n, s), c in cnt.items():
    byop[n] += c; byop_t[n] += tim[(n, s)]
print("--- by op")
for n, c in byop.most_common(25):
    print(f"{c:5d} {byop_t[n]/1e3:7.3f} ms  {n}")
print("--- by op and shapes")
for (n, s), c in cnt.most_common(90):
    print(f"{c:5d} {tim[(n, s)]/1e3:7.3f} ms  {n:28s} {s}")
