#!/usr/bin/env python3
"""Generate tests/golden/model_tiny.npz and tests/golden/ops.npz by RUNNING THE REFERENCE
(/root/reference, imported read-only, CPU fp32) — build container only.

What is stored is data only: a tiny-config state_dict (seeded random init), inputs, the noise
tensors the reference drew (captured by wrapping torch.randn/randn_like/rand), and its outputs.
No reference source is copied.  The reference's `monotonic_align` package ships only Windows
binaries, so a stand-in module with the same `maximum_path(neg_cent, mask)` contract is put in
sys.modules; it calls the reference's own Cython routine compiled into oracle/_ref/.
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from oracle import mas as omas  # noqa: E402

captured = {}


def _install_monotonic_align():
    m = types.ModuleType("monotonic_align")

    def maximum_path(neg_cent, mask):
        nc = neg_cent.data.cpu().numpy().astype(np.float32)
        t_ys = mask.sum(1)[:, 0].data.cpu().numpy().astype(np.int32)
        t_xs = mask.sum(2)[:, 0].data.cpu().numpy().astype(np.int32)
        captured["neg_cent"] = nc.copy()
        path = omas.mas_reference(nc, t_ys, t_xs)
        return torch.from_numpy(path).to(device=neg_cent.device, dtype=neg_cent.dtype)

    m.maximum_path = maximum_path
    sys.modules["monotonic_align"] = m


class NoiseTap:
    """Records every tensor drawn through torch.randn / randn_like / rand while active."""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self._orig = (torch.randn, torch.randn_like, torch.rand)
        tap = self

        def randn(*a, **k):
            t = tap._orig[0](*a, **k); tap.draws.append(t.detach().clone()); return t

        def randn_like(*a, **k):
            t = tap._orig[1](*a, **k); tap.draws.append(t.detach().clone()); return t

        def rand(*a, **k):
            t = tap._orig[2](*a, **k); tap.draws.append(t.detach().clone()); return t

        torch.randn, torch.randn_like, torch.rand = randn, randn_like, rand
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like, torch.rand = self._orig


TINY = dict(
    n_vocab=20, spec_channels=17, segment_size=8, n_speakers=3,
    model=dict(inter_channels=16, hidden_channels=16, filter_channels=32, n_heads=2, n_layers=2, kernel_size=3,
               p_dropout=0.1, resblock="1", resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5]],
               upsample_rates=[4, 4], upsample_initial_channel=32, upsample_kernel_sizes=[8, 8], gin_channels=8))


def np_(t):
    return t.detach().cpu().numpy()


def main():
    _install_monotonic_align()
    sys.path.insert(0, "/root/reference")
    import commons, transforms, modules, attentions, losses, models  # noqa: E401

    out = {}
    ops = {}
    torch.manual_seed(1234)
    cfg = TINY
    net = models.SynthesizerTrn(cfg["n_vocab"], cfg["spec_channels"], cfg["segment_size"],
                                n_speakers=cfg["n_speakers"], **cfg["model"])
    # the zero-initialised output projections (post / proj of the flows) would hide everything
    # behind them: perturb all parameters a little so every path carries signal.
    with torch.no_grad():
        for p in net.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    net.eval()
    out["config"] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    for k, v in net.state_dict().items():
        out["sd/" + k] = np_(v)

    B = 2
    x_len = torch.tensor([11, 7]); y_len = torch.tensor([24, 17])
    x = torch.randint(1, cfg["n_vocab"], (B, 11)); x[1, 7:] = 0
    spec = torch.rand(B, cfg["spec_channels"], 24) * 2; spec[1, :, 17:] = 0
    sid = torch.tensor([0, 2])
    out.update({"in/x": np_(x), "in/x_lengths": np_(x_len), "in/spec": np_(spec), "in/spec_lengths": np_(y_len), "in/sid": np_(sid)})

    # ---- forward (train graph), with gradients of a scalar probe wrt a few parameters
    with NoiseTap() as tap:
        o, l_length, attn, ids_slice, x_mask, y_mask, (z, z_p, m_p, logs_p, m_q, logs_q) = net(x, x_len, spec, y_len, sid)
    for i, d in enumerate(tap.draws):
        out[f"fwd/noise{i}"] = np_(d)
    out["fwd/n_noise"] = np.array(len(tap.draws))
    out["fwd/neg_cent"] = captured["neg_cent"]
    for name, t in dict(o=o, l_length=l_length, attn=attn, ids_slice=ids_slice, x_mask=x_mask, y_mask=y_mask, z=z, z_p=z_p,
                        m_p=m_p, logs_p=logs_p, m_q=m_q, logs_q=logs_q).items():
        out["fwd/" + name] = np_(t)
    probe = o.pow(2).mean() + l_length.sum() + losses.kl_loss(z_p, logs_q, m_p, logs_p, y_mask)
    probe.backward()
    out["fwd/probe"] = np_(probe)
    for k in ["enc_p.emb.weight", "enc_p.encoder.attn_layers.0.emb_rel_k", "enc_p.encoder.attn_layers.1.conv_q.weight",
              "enc_q.enc.in_layers.3.weight_v", "enc_q.enc.in_layers.3.weight_g", "flow.flows.2.post.weight",
              "dp.flows.1.proj.weight", "dp.post_flows.3.convs.convs_sep.1.weight", "dec.ups.1.weight_v",
              "dec.ups.1.weight_g", "dec.resblocks.2.convs1.1.weight_v", "dec.conv_post.weight", "emb_g.weight",
              "dec.cond.weight", "enc_q.pre.weight"]:
        out["fwd/grad/" + k] = np_(dict(net.named_parameters())[k].grad)

    # ---- infer
    with torch.no_grad(), NoiseTap() as tap:
        o_i, attn_i, y_mask_i, (z_i, z_p_i, m_p_i, logs_p_i) = net.infer(x, x_len, sid, noise_scale=0.667, length_scale=1.1, noise_scale_w=0.8)
    for i, d in enumerate(tap.draws):
        out[f"infer/noise{i}"] = np_(d)
    out["infer/n_noise"] = np.array(len(tap.draws))
    for name, t in dict(o=o_i, attn=attn_i, y_mask=y_mask_i, z=z_i, z_p=z_p_i, m_p=m_p_i, logs_p=logs_p_i).items():
        out["infer/" + name] = np_(t)

    # ---- voice conversion
    with torch.no_grad(), NoiseTap() as tap:
        o_vc, y_mask_vc, (z_vc, z_p_vc, z_hat_vc) = net.voice_conversion(spec, y_len, torch.tensor([0, 2]), torch.tensor([1, 0]))
    out["vc/noise0"] = np_(tap.draws[0])
    for name, t in dict(o=o_vc, z=z_vc, z_p=z_p_vc, z_hat=z_hat_vc).items():
        out["vc/" + name] = np_(t)

    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "model_tiny.npz"), **out)

    # =================== per-op fixtures ===================
    torch.manual_seed(4321)
    # (1) rational-quadratic spline, forward + inverse, with gradients
    n = 512
    xin = torch.linspace(-8, 8, n).reshape(2, 1, n // 2).clone()
    xin[0, 0, 5] = 5.0; xin[0, 0, 6] = -5.0
    uw, uh = torch.randn(2, 1, n // 2, 10), torch.randn(2, 1, n // 2, 10)
    ud = torch.randn(2, 1, n // 2, 9)
    for inverse in (False, True):
        a = [t.clone().requires_grad_(True) for t in (xin, uw, uh, ud)]
        y, lad = transforms.piecewise_rational_quadratic_transform(a[0], a[1], a[2], a[3], inverse=inverse, tails="linear", tail_bound=5.0)
        (y * torch.cos(y)).sum().add((lad * 0.7).sum()).backward()
        tag = "spline_inv/" if inverse else "spline_fwd/"
        ops.update({tag + "x": np_(xin), tag + "uw": np_(uw), tag + "uh": np_(uh), tag + "ud": np_(ud), tag + "y": np_(y), tag + "lad": np_(lad),
                    tag + "gx": np_(a[0].grad), tag + "guw": np_(a[1].grad), tag + "guh": np_(a[2].grad), tag + "gud": np_(a[3].grad)})

    # (2) relative-position attention (one MultiHeadAttention) for several lengths incl. T <= window+1
    for T in (3, 5, 9, 50):
        mha = attentions.MultiHeadAttention(16, 16, 2, p_dropout=0.0, window_size=4).eval()
        xx = torch.randn(2, 16, T)
        lens = torch.tensor([T, max(1, T - 2)])
        xm = commons.sequence_mask(lens, T).unsqueeze(1).float()
        am = xm.unsqueeze(2) * xm.unsqueeze(-1)
        yy = mha(xx, xx, am)
        tag = f"mha{T}/"
        ops.update({tag + "x": np_(xx), tag + "lens": np_(lens), tag + "y": np_(yy), tag + "p_attn": np_(mha.attn)})
        for k, v in mha.state_dict().items():
            ops[tag + "sd/" + k] = np_(v)

    # (3) commons helpers
    dur = torch.tensor([[[2., 0., 3., 1.]], [[1., 1., 0., 0.]]])
    xm = torch.tensor([[[1., 1., 1., 1.]], [[1., 1., 0., 0.]]])
    ylen = torch.clamp_min(dur.sum([1, 2]), 1).long()
    ym = commons.sequence_mask(ylen, None).unsqueeze(1).float()
    amask = xm.unsqueeze(2) * ym.unsqueeze(-1)
    ops["genpath/dur"] = np_(dur); ops["genpath/mask"] = np_(amask); ops["genpath/path"] = np_(commons.generate_path(dur, amask))
    xs = torch.randn(3, 4, 20); ids = torch.tensor([0, 7, 15])
    ops["slice/x"] = np_(xs); ops["slice/ids"] = np_(ids); ops["slice/y"] = np_(commons.slice_segments(xs, ids, 5))

    # (4) losses on random feature maps
    fr = [[torch.randn(2, 4, 9), torch.randn(2, 3, 5, 2)], [torch.randn(2, 1, 7)]]
    fg = [[torch.randn(2, 4, 9), torch.randn(2, 3, 5, 2)], [torch.randn(2, 1, 7)]]
    dr, dg = [torch.randn(2, 11), torch.randn(2, 6)], [torch.randn(2, 11), torch.randn(2, 6)]
    for i, (a, b) in enumerate(zip(sum(fr, []), sum(fg, []))):
        ops[f"loss/fr{i}"] = np_(a); ops[f"loss/fg{i}"] = np_(b)
    for i in range(2):
        ops[f"loss/dr{i}"] = np_(dr[i]); ops[f"loss/dg{i}"] = np_(dg[i])
    ops["loss/feature"] = np_(losses.feature_loss(fr, fg))
    ops["loss/disc"] = np_(losses.discriminator_loss(dr, dg)[0])
    ops["loss/gen"] = np_(losses.generator_loss(dg)[0])
    zp, lq, mp, lp = (torch.randn(2, 6, 9) for _ in range(4))
    zm = commons.sequence_mask(torch.tensor([9, 5]), 9).unsqueeze(1).float()
    ops.update({"loss/kl_zp": np_(zp), "loss/kl_lq": np_(lq), "loss/kl_mp": np_(mp), "loss/kl_lp": np_(lp), "loss/kl_mask": np_(zm),
                "loss/kl": np_(losses.kl_loss(zp, lq, mp, lp, zm))})

    # (5) linear spectrogram of the reference (mel_processing.spectrogram_torch); librosa is not
    #     installed, so stub modules are registered first (the mel filterbank itself is third-party
    #     librosa==0.9.2 arithmetic: NOT produced here -> mel-basis parity stays unpinned).
    lib = types.ModuleType("librosa"); lib_util = types.ModuleType("librosa.util"); lib_f = types.ModuleType("librosa.filters")
    lib_util.normalize = lib_util.pad_center = lib_util.tiny = None
    lib_f.mel = None
    lib.util, lib.filters = lib_util, lib_f
    sys.modules.update({"librosa": lib, "librosa.util": lib_util, "librosa.filters": lib_f})
    import mel_processing
    wav = torch.rand(2, 2048) * 1.6 - 0.8
    ops["stft/wav"] = np_(wav)
    ops["stft/spec_1024_256"] = np_(mel_processing.spectrogram_torch(wav, 1024, 22050, 256, 1024))
    ops["stft/spec_64_16"] = np_(mel_processing.spectrogram_torch(wav[:, :256], 64, 22050, 16, 64))

    # (6) discriminators: too large to store (46.7 M parameters, fixed architecture); store the output
    #     of a seeded instance on a small input plus parameter checksums so a re-implementation that
    #     reproduces torch's seeded construction order can be compared.
    torch.manual_seed(99)
    d = models.MultiPeriodDiscriminator(False).eval()
    yw = torch.rand(1, 1, 2048) * 2 - 1
    yh = torch.rand(1, 1, 2048) * 2 - 1
    with torch.no_grad():
        y_d_rs, y_d_gs, fmap_rs, fmap_gs = d(yw, yh)
    ops["disc/y"] = np_(yw); ops["disc/y_hat"] = np_(yh)
    for i in range(6):
        ops[f"disc/logit_r{i}"] = np_(y_d_rs[i]); ops[f"disc/logit_g{i}"] = np_(y_d_gs[i])
        ops[f"disc/fmap_r{i}_absmean"] = np.array([float(f.abs().mean()) for f in fmap_rs[i]], np.float64)
    ops["disc/param_checksum"] = np.array([float(p.double().sum()) for p in d.parameters()], np.float64)
    ops["disc/param_names"] = np.frombuffer(json.dumps([k for k, _ in d.named_parameters()]).encode(), dtype=np.uint8)

    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ops.npz"), **ops)
    for f in ("model_tiny.npz", "ops.npz"):
        print(f, os.path.getsize(os.path.join(ROOT, "tests", "golden", f)), "bytes")
    print("G params:", sum(p.numel() for p in net.parameters()), "keys:", len(net.state_dict()))


if __name__ == "__main__":
    main()
