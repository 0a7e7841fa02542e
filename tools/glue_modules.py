#!/usr/bin/env python3
"""Per sub-module of the fine-tune step (bf16, workload C2): device launches and time of one forward + backward, split into
this repo's kernels and aten/library kernels, with the most frequent aten ops — to decide which glue to fuse next."""
import collections, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from importlib import import_module
from torch.profiler import profile, ProfilerActivity
P = import_module("personalized_text-to-speech_amd"); cfgs = import_module("personalized_text-to-speech_amd.configs"); tr = import_module("personalized_text-to-speech_amd.train")
from importlib import import_module as im
wa = im("personalized_text-to-speech_amd.weight_arena")
hps = cfgs.get("modified_finetune_speaker"); dev = "cuda:0"
ft = tr.FineTuner(hps, dev, amp=True)
batch = tr.synthetic_batch(hps, 16, (200, 500), dev)
x, xl, spec, sl, y, yl, sid = batch
g = ft.net_g; d = ft.net_d
ac = lambda: torch.autocast("cuda", dtype=torch.bfloat16)
with torch.no_grad(), ac(), g._scope():
    h, m_p, logs_p, x_mask = g.enc_p(x, xl); gg = g.emb_g(sid).unsqueeze(-1)
    z, m_q, logs_q, y_mask = g.enc_q(spec, sl, g=gg); z_p = g.flow(z, y_mask, g=gg)
    nc = g.neg_cent(z_p, m_p, logs_p); am = (x_mask.unsqueeze(2) * y_mask.unsqueeze(-1)).squeeze(1)
    attn = P.kernels.maximum_path(nc, am).unsqueeze(1); w = attn.sum(2)
    zs, ids = P.commons.rand_slice_segments(z, sl, 32)
    o = g.dec(zs, g=gg)
h, gg, z, zs, w, o = (t.detach().float() for t in (h, gg, z, zs, w, o))
yr = torch.randn_like(o)
OWN = ("anonymous namespace", "_GLOBAL__N_")

def run(name, fwd, scope=True):
    def once():
        with ac():
            if scope:
                with g._scope():
                    out = fwd()
            else:
                out = fwd()
        outs = [t for t in (out if isinstance(out, (tuple, list)) else [out]) if torch.is_tensor(t) and t.requires_grad]
        sum(t.float().pow(2).mean() for t in outs).backward()
    for _ in range(2):
        once()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        once(); torch.cuda.synchronize()
    own = [0, 0.0]; oth = [0, 0.0]; ops = collections.Counter(); opt = collections.Counter()
    for e in prof.events():
        ks = getattr(e, "kernels", [])
        if not ks or any(len(getattr(c, "kernels", [])) for c in e.cpu_children):
            continue
        for k in ks:
            tgt = own if any(s in k.name for s in OWN) and "at::native" not in k.name else oth
            tgt[0] += 1; tgt[1] += k.duration
            if tgt is oth:
                key = (e.name, str([s for s in e.input_shapes if s])[:70]); ops[key] += 1; opt[key] += k.duration
    print(f"== {name}: own {own[0]} launches {own[1]/1e3:.2f} ms | aten/library {oth[0]} launches {oth[1]/1e3:.2f} ms", flush=True)
    for key, c in ops.most_common(int(os.environ.get("TOPN", "14"))):
        print(f"     {c:4d} {opt[key]/1e3:6.3f} ms  {key[0]:26s} {key[1]}")

run("enc_p (text encoder)", lambda: g.enc_p(x, xl)[:3])
run("enc_q (posterior, WN16)", lambda: g.enc_q(spec, sl, g=gg.requires_grad_())[:3])
run("flow (4 coupling, WN4)", lambda: g.flow(z.requires_grad_(), y_mask, g=gg))
run("dp (stochastic duration)", lambda: g.dp(h.requires_grad_(), x_mask, w, g=gg))
run("dec (HiFi-GAN)", lambda: g.dec(zs.requires_grad_(), g=gg))
def dfw():
    rs, gs, fr, fg = d(yr, o.requires_grad_())
    return list(gs) + [f for fm in fg for f in fm]
run("D (MPD: real+fake batch)", dfw, scope=False)
def mel():
    m = P.mel_processing.mel_spectrogram_torch(o.requires_grad_().squeeze(1), hps.data.filter_length, hps.data.n_mel_channels, hps.data.sampling_rate, hps.data.hop_length, hps.data.win_length, hps.data.mel_fmin, hps.data.mel_fmax)
    return m
run("mel of y_hat", mel, scope=False)
