#!/usr/bin/env python3
"""aten / library device launches of one eager fine-tune step attributed to the package source line that issued them
(backward launches are attributed to the line of the forward op whose autograd node they belong to, via sequence numbers)."""
import collections, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from importlib import import_module
from torch.profiler import profile, ProfilerActivity
P = import_module("personalized_text-to-speech_amd"); cfgs = import_module("personalized_text-to-speech_amd.configs"); tr = import_module("personalized_text-to-speech_amd.train")
cfg_name, batch_size, t_y_range = cfgs.WORKLOADS["C2"]
hps = cfgs.get(cfg_name)
ft = tr.FineTuner(hps, "cuda:0", amp=True)
batch = tr.synthetic_batch(hps, batch_size, t_y_range, "cuda:0")
for _ in range(2):
    ft.step(batch)
torch.cuda.synchronize()
# a profiler range per module call: forward launches are attributed to the innermost module
names = {}
for root, tag in ((ft.net_g, "G"), (ft.net_d, "D")):
    for n, m in root.named_modules():
        names[id(m)] = f"{tag}.{n}:{type(m).__name__}"
ranges = {}
def pre(m, args):
    r = torch.profiler.record_function("M:" + names.get(id(m), type(m).__name__)); r.__enter__(); ranges.setdefault(id(m), []).append(r)
def post(m, args, out):
    ranges[id(m)].pop().__exit__(None, None, None)
torch.nn.modules.module.register_module_forward_pre_hook(pre)
torch.nn.modules.module.register_module_forward_hook(post, always_call=True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    ft.step(batch)
    torch.cuda.synchronize()
evs = list(prof.events())
PKG = "personalized_text-to-speech_amd"

import re
def site(e):
    p = e.cpu_parent
    while p is not None:
        if p.name.startswith("M:"):
            return re.sub(r"\.\d+", ".N", p.name[2:])[:70]          # layers of a stack share a line
        p = p.cpu_parent
    return None

fwd_site = {}
for e in evs:
    if e.sequence_nr is not None and e.sequence_nr >= 0 and not e.name.startswith("autograd::engine") and "Backward" not in e.name:
        s = site(e)
        if s and e.sequence_nr not in fwd_site:
            fwd_site[e.sequence_nr] = s
OWN = ("anonymous namespace", "_GLOBAL__N_")
cnt = collections.Counter(); tim = collections.Counter(); ops = collections.defaultdict(collections.Counter)
for e in evs:
    ks = getattr(e, "kernels", [])
    if not ks or any(len(getattr(c, "kernels", [])) for c in e.cpu_children):
        continue
    ks = [k for k in ks if not (any(s in k.name for s in OWN) and "at::native" not in k.name)]
    if not ks:
        continue
    s, phase = site(e), "fwd"
    if s is None:
        p = e
        while p is not None:
            if p.name.startswith("autograd::engine::evaluate_function"):
                s = fwd_site.get(p.sequence_nr); phase = "bwd"
                if s is None:
                    s = p.name.split(": ")[-1]
                break
            p = p.cpu_parent
    key = (s or "?", phase)
    cnt[key] += len(ks); tim[key] += sum(k.duration for k in ks); ops[key][e.name] += len(ks)
print("aten/library launches", sum(cnt.values()), "ms", sum(tim.values()) / 1e3)
byfile = collections.Counter(); byfile_t = collections.Counter()
for (s, ph), c in cnt.items():
    f = s.split("(")[0]; byfile[f] += c; byfile_t[f] += tim[(s, ph)]
for f, c in byfile.most_common(20):
    print(f"{c:5d} {byfile_t[f]/1e3:7.3f} ms  {f}")
print("--- by site")
for key, c in cnt.most_common(int(os.environ.get("TOPN", "110"))):
    top = ", ".join(f"{n.replace('aten::','')}x{k}" for n, k in ops[key].most_common(4))
    print(f"{c:5d} {tim[key]/1e3:7.3f} ms  {key[1]} {key[0]:72s} {top}")
