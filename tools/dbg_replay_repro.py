#!/usr/bin/env python3
"""Which tensors differ between two replays of the captured step from the same state?  (verify_replay's diagnosis tool.)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from importlib import import_module
P = import_module("personalized_text-to-speech_amd"); cfgs = import_module("personalized_text-to-speech_amd.configs"); tr = import_module("personalized_text-to-speech_amd.train")
cfg_name, batch_size, t_y_range = cfgs.WORKLOADS[os.environ.get("WL", "C2")]
hps = cfgs.get(cfg_name)
X = os.environ.get("EXPERIMENT", "")
if "nodefer" in X:                      # side-lane convolutions reduce their slabs immediately
    WA = import_module("personalized_text-to-speech_amd.weight_arena")
    main_defer = WA.WeightArena.defer
    WA.WeightArena.defer = property(lambda self: None)
if "freshws" in X:                      # no persistent scratch: a fresh buffer per call
    K = P.kernels
    K.workspace = lambda nbytes, device: torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
    import_module("personalized_text-to-speech_amd.rowops").K = K
if "sync" in X:                         # every launch of the C library followed by a device-wide barrier on the host side
    pass
ft = tr.FineTuner(hps, "cuda:0", amp=os.environ.get("FP32") != "1")
if os.environ.get("BRANCHES") is not None:
    ft.side_branches = frozenset(b for b in os.environ["BRANCHES"].split(",") if b)
batch = tr.synthetic_batch(hps, batch_size, t_y_range, "cuda:0")
ft.capture(batch, warmup=3, verify=False)
ts = ft._state_tensors()
snap = [t.detach().clone() for t in ts]
rng = torch.cuda.get_rng_state(ft.device)
named = [("G." + k, p) for k, p in ft.net_g.named_parameters()] + [("D." + k, p) for k, p in ft.net_d.named_parameters()]

def run(fn):
    with torch.no_grad():
        for t, s in zip(ts, snap):
            t.copy_(s)
    torch.cuda.set_rng_state(rng, ft.device)
    out = fn()
    torch.cuda.synchronize()
    res = {"out." + k: v.detach().clone() for k, v in out.items()}
    for k, p in named:
        res["param." + k] = p.detach().clone()
        if p.grad is not None:
            res["grad." + k] = p.grad.detach().clone()
    return res

NR = int(os.environ.get("NREP", "3"))
runs = [run(ft.replay) for _ in range(NR)] + [run(lambda: ft.step(batch))]
for i, name in [(j, f"replay{j+1}") for j in range(1, NR)] + [(NR, "eager")]:
    diff = []
    for k in runs[0]:
        if k in runs[i] and not torch.equal(runs[0][k], runs[i][k]):
            a, b = runs[0][k].double(), runs[i][k].double()
            diff.append((float((a - b).abs().max() / (b.abs().max() + 1e-30)), k))
    diff.sort(reverse=True)
    print(f"--- replay1 vs {name}: {len(diff)} of {len(runs[0])} tensors differ")
    kinds = {}
    for d, k in diff:
        kk = k.split(".")[0] + "." + k.split(".")[1]
        kinds[kk] = kinds.get(kk, 0) + 1
    print("   by kind:", kinds)
    for d, k in diff[:25]:
        print(f"   {d:.3e}  {k}")
    if diff and os.environ.get("DETAIL"):
        k = diff[0][1]
        a, b = runs[0][k].double().flatten(), runs[i][k].double().flatten()
        nz = (a != b).nonzero().flatten()
        print(f"   detail {k}: shape {tuple(runs[0][k].shape)}, {nz.numel()} of {a.numel()} elements differ; first idx {nz[:12].tolist()}")
        print("      a:", [f"{v:.4e}" for v in a[nz[:8]].tolist()]); print("      b:", [f"{v:.4e}" for v in b[nz[:8]].tolist()])
        for kk in [d[1] for d in diff if "convs_sep.2.bias" in d[1] or "pre.weight" in d[1] or "pre.bias" in d[1]][:3]:
            a2, b2 = runs[0][kk].double().flatten(), runs[i][kk].double().flatten()
            print(f"   detail {kk}: {(a2 != b2).sum().item()} of {a2.numel()} differ, max rel {float(((a2-b2).abs()/(b2.abs()+1e-30)).max()):.2e}")
    outs = {k: (float(runs[0][k]), float(runs[i][k])) for k in runs[0] if k.startswith("out.")}
    print("   out:", {k: v for k, v in outs.items() if v[0] != v[1]})
