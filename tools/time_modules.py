#!/usr/bin/env python3
"""GPU time of each sub-module of the fine-tune step (forward and backward separately), bf16
autocast, workload C2 — to decide what to fuse next.  Eager launches, CUDA events, median of n."""
import os, sys, statistics
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from importlib import import_module
P = import_module("personalized_text-to-speech_amd"); cfgs = import_module("personalized_text-to-speech_amd.configs"); tr = import_module("personalized_text-to-speech_amd.train")
hps = cfgs.get("modified_finetune_speaker"); dev = "cuda:0"
ft = tr.FineTuner(hps, dev, amp=True)
batch = tr.synthetic_batch(hps, 16, (200, 500), dev)
x, xl, spec, sl, y, yl, sid = batch
g = ft.net_g; d = ft.net_d
ac = lambda: torch.autocast("cuda", dtype=torch.bfloat16)

def timeit(name, fwd, n=7):
    fs, bs = [], []
    for i in range(n + 2):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        torch.cuda.synchronize(); e[0].record()
        with ac():
            out = fwd()
        outs = [o for o in (out if isinstance(out, (tuple, list)) else [out]) if torch.is_tensor(o) and o.requires_grad]
        loss = sum(o.float().pow(2).mean() for o in outs)
        e[1].record()
        loss.backward()
        e[2].record(); torch.cuda.synchronize()
        if i >= 2:
            fs.append(e[0].elapsed_time(e[1])); bs.append(e[1].elapsed_time(e[2]))
    print(f"{name:28s} fwd {statistics.median(fs):8.2f} ms   bwd {statistics.median(bs):8.2f} ms", flush=True)

with torch.no_grad(), ac():
    h, m_p, logs_p, x_mask = g.enc_p(x, xl); gg = g.emb_g(sid).unsqueeze(-1)
    z, m_q, logs_q, y_mask = g.enc_q(spec, sl, g=gg); z_p = g.flow(z, y_mask, g=gg)
    nc = g.neg_cent(z_p, m_p, logs_p); am = (x_mask.unsqueeze(2) * y_mask.unsqueeze(-1)).squeeze(1)
    attn = P.kernels.maximum_path(nc, am).unsqueeze(1); w = attn.sum(2)
    zs, ids = P.commons.rand_slice_segments(z, sl, 32)
    o = g.dec(zs, g=gg)
h, gg, z, zs, w, o = (t.detach().float() for t in (h, gg, z, zs, w, o))
yr = torch.randn_like(o)
timeit("enc_p (text encoder)", lambda: g.enc_p(x, xl)[:3])
timeit("enc_q (posterior, WN16)", lambda: g.enc_q(spec, sl, g=gg.requires_grad_())[:3])
timeit("flow (4 coupling, WN4)", lambda: g.flow(z.requires_grad_(), y_mask, g=gg))
timeit("dp (stochastic duration)", lambda: g.dp(h.requires_grad_(), x_mask, w, g=gg))
timeit("dec (HiFi-GAN, HIP)", lambda: g.dec(zs.requires_grad_(), g=gg))
def dfw():
    rs, gs, fr, fg = d(yr, o.requires_grad_())
    return list(gs) + [f for fm in fg for f in fm]
timeit("D (MPD: real+fake batch)", dfw)
timeit("neg_cent + MAS", lambda: P.kernels.maximum_path(g.neg_cent(z_p, m_p, logs_p), am).sum() * h.requires_grad_().sum())
