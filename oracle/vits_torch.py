"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product package).

A functional, plain-PyTorch fp32 restatement of the reference's VITS graph, written against a
flat `sd` dict of tensors keyed like the reference's state_dict.  It is pinned by
tests/golden/model_tiny.npz and tests/golden/ops.npz, which were produced by running the reference
itself (tools/gen_golden_model.py), and serves as
  * the checker for the product's modules and HIP kernels (tests/, smoke()), and
  * bench.py's `cpu_baseline` leg (kind "port"): this graph + the C alignment DP on host cores.

Every function cites the reference lines it follows.  Noise is passed in explicitly (`noise` is a
list consumed in the order the reference draws: models.py:240 randn_like, :67 randn,
commons.py:65 rand for the train graph; models.py:90 randn, :520 randn_like for infer).
"""
import math

import numpy as np
import torch
from torch.nn import functional as F

from . import mas as omas

LRELU = 0.1
LOG_2PI = math.log(2 * math.pi)


# ----------------------------------------------------------------------------- helpers
def _w(sd, p):
    """Weight of conv `p`: plain, or g*v/||v|| over all dims but 0 (legacy weight_norm, dim=0)."""
    if p + ".weight" in sd:
        return sd[p + ".weight"]
    v, g = sd[p + ".weight_v"], sd[p + ".weight_g"]
    return v * (g / torch.linalg.vector_norm(v, 2, dim=tuple(range(1, v.dim())), keepdim=True))


def conv(sd, p, x, padding=0, dilation=1, groups=1, stride=1):
    return F.conv1d(x, _w(sd, p), sd.get(p + ".bias"), stride, padding, dilation, groups)


def sequence_mask(length, max_length=None):                  # commons.py:124-128
    if max_length is None:
        max_length = int(length.max())
    return torch.arange(max_length, device=length.device)[None, :] < length[:, None]


def layer_norm(sd, p, x, eps=1e-5):                          # modules.py:20-32
    return F.layer_norm(x.transpose(1, -1), (x.size(1),), sd[p + ".gamma"], sd[p + ".beta"], eps).transpose(1, -1)


# ----------------------------------------------------------------------------- attention
def rel_embeddings(emb, length, window):                     # attentions.py:199-212
    pad_length = max(length - (window + 1), 0)
    start = max((window + 1) - length, 0)
    if pad_length > 0:
        emb = F.pad(emb, (0, 0, pad_length, pad_length))
    return emb[:, start:start + 2 * length - 1]


def rel_to_abs(x):                                           # attentions.py:214-229
    b, h, l, _ = x.size()
    x = F.pad(x, (0, 1))
    x = F.pad(x.reshape(b, h, l * 2 * l), (0, l - 1))
    return x.view(b, h, l + 1, 2 * l - 1)[:, :, :l, l - 1:]


def abs_to_rel(x):                                           # attentions.py:231-243
    b, h, l, _ = x.size()
    x = F.pad(x, (0, l - 1))
    x = F.pad(x.reshape(b, h, l * l + l * (l - 1)), (l, 0))
    return x.view(b, h, l, 2 * l)[:, :, :, 1:]


def mha(sd, p, x, attn_mask, n_heads, window=4):             # attentions.py:141-182
    q, k, v = conv(sd, p + ".conv_q", x), conv(sd, p + ".conv_k", x), conv(sd, p + ".conv_v", x)
    b, d, t = q.shape
    dk = d // n_heads
    q = q.view(b, n_heads, dk, t).transpose(2, 3)
    k = k.view(b, n_heads, dk, t).transpose(2, 3)
    v = v.view(b, n_heads, dk, t).transpose(2, 3)
    scores = torch.matmul(q / math.sqrt(dk), k.transpose(-2, -1))
    ek = rel_embeddings(sd[p + ".emb_rel_k"], t, window)
    scores = scores + rel_to_abs(torch.matmul(q / math.sqrt(dk), ek.unsqueeze(0).transpose(-2, -1)))
    scores = scores.masked_fill(attn_mask == 0, -1e4)
    p_attn = F.softmax(scores, dim=-1)
    out = torch.matmul(p_attn, v)
    ev = rel_embeddings(sd[p + ".emb_rel_v"], t, window)
    out = out + torch.matmul(abs_to_rel(p_attn), ev.unsqueeze(0))
    out = out.transpose(2, 3).contiguous().view(b, d, t)
    return conv(sd, p + ".conv_o", out), p_attn


def text_encoder(sd, cfg, x, x_lengths):                     # models.py:167-176, attentions.py:35-47,277-303
    H = cfg["hidden_channels"]
    h = sd["enc_p.emb.weight"][x] * math.sqrt(H)
    h = h.transpose(1, -1)
    x_mask = sequence_mask(x_lengths, h.size(2)).unsqueeze(1).to(h.dtype)
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    h = h * x_mask                                           # (Encoder.forward multiplies again: idempotent)
    k = cfg["kernel_size"]
    for i in range(cfg["n_layers"]):
        e = "enc_p.encoder."
        y, _ = mha(sd, f"{e}attn_layers.{i}", h, attn_mask, cfg["n_heads"])
        h = layer_norm(sd, f"{e}norm_layers_1.{i}", h + y)
        y = conv(sd, f"{e}ffn_layers.{i}.conv_1", F.pad(h * x_mask, ((k - 1) // 2, k // 2)))
        y = conv(sd, f"{e}ffn_layers.{i}.conv_2", F.pad(torch.relu(y) * x_mask, ((k - 1) // 2, k // 2))) * x_mask
        h = layer_norm(sd, f"{e}norm_layers_2.{i}", h + y)
    h = h * x_mask
    stats = conv(sd, "enc_p.proj", h) * x_mask
    m, logs = torch.split(stats, cfg["inter_channels"], dim=1)
    return h, m, logs, x_mask


# ----------------------------------------------------------------------------- WN / flows
def wn(sd, p, x, x_mask, g, hidden, n_layers, kernel=5):     # modules.py:148-176, commons.py:103-110
    out = torch.zeros_like(x)
    if g is not None:
        g = conv(sd, p + ".cond_layer", g)
    for i in range(n_layers):
        x_in = conv(sd, f"{p}.in_layers.{i}", x, padding=(kernel - 1) // 2)
        if g is not None:
            x_in = x_in + g[:, i * 2 * hidden:(i + 1) * 2 * hidden, :]
        acts = torch.tanh(x_in[:, :hidden]) * torch.sigmoid(x_in[:, hidden:])
        rs = conv(sd, f"{p}.res_skip_layers.{i}", acts)
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * x_mask
            out = out + rs[:, hidden:]
        else:
            out = out + rs
    return out * x_mask


def posterior_encoder(sd, cfg, y, y_lengths, g, eps):        # models.py:234-241
    y_mask = sequence_mask(y_lengths, y.size(2)).unsqueeze(1).to(y.dtype)
    h = conv(sd, "enc_q.pre", y) * y_mask
    h = wn(sd, "enc_q.enc", h, y_mask, g, cfg["hidden_channels"], 16)
    stats = conv(sd, "enc_q.proj", h) * y_mask
    m, logs = torch.split(stats, cfg["inter_channels"], dim=1)
    z = (m + eps * torch.exp(logs)) * y_mask
    return z, m, logs, y_mask


def coupling_block(sd, cfg, x, x_mask, g, reverse=False):    # models.py:202-209, modules.py:324-343,270-277
    half = cfg["inter_channels"] // 2
    order = range(4) if not reverse else reversed(range(4))
    for i in order:
        p = f"flow.flows.{2 * i}"
        if reverse:
            x = torch.flip(x, [1])                            # Flip comes first when running backwards
        x0, x1 = torch.split(x, [half, half], 1)
        h = conv(sd, p + ".pre", x0) * x_mask
        h = wn(sd, p + ".enc", h, x_mask, g, cfg["hidden_channels"], 4)
        m = conv(sd, p + ".post", h) * x_mask                 # mean_only=True: logs = 0
        x1 = (m + x1 * x_mask) if not reverse else (x1 - m) * x_mask
        x = torch.cat([x0, x1], 1)
        if not reverse:
            x = torch.flip(x, [1])
    return x


# ----------------------------------------------------------------------------- spline / duration predictor
def rq_spline(inputs, uw, uh, ud, inverse, tail_bound=5.0, min_w=1e-3, min_h=1e-3, min_d=1e-3):
    """transforms.py:55-193, evaluated on every element (the reference gathers the in-interval
    elements first; the arithmetic per element is the same) and the linear tails selected last."""
    inside = (inputs >= -tail_bound) & (inputs <= tail_bound)
    x = torch.where(inside, inputs, torch.zeros_like(inputs))
    const = math.log(math.exp(1 - min_d) - 1)
    ud = F.pad(ud, (1, 1), value=const)
    nb = uw.shape[-1]

    def knots(u, mn):
        p = mn + (1 - mn * nb) * F.softmax(u, dim=-1)
        c = F.pad(torch.cumsum(p, dim=-1), (1, 0), value=0.0)
        c = 2 * tail_bound * c - tail_bound
        c = torch.cat([torch.full_like(c[..., :1], -tail_bound), c[..., 1:-1], torch.full_like(c[..., :1], tail_bound)], -1)
        return c, c[..., 1:] - c[..., :-1]

    cw, w = knots(uw, min_w)
    ch, h = knots(uh, min_h)
    d = min_d + F.softplus(ud)
    locs = ch if inverse else cw
    locs = torch.cat([locs[..., :-1], locs[..., -1:] + 1e-6], -1)      # searchsorted eps, transforms.py:47-52
    idx = (torch.sum(x[..., None] >= locs, dim=-1) - 1)[..., None]
    tk = lambda t: t.gather(-1, idx)[..., 0]
    icw, iw, ich, ih = tk(cw), tk(w), tk(ch), tk(h)
    idl, idv, idv1 = tk(h / w), tk(d), tk(d[..., 1:])
    if inverse:
        a = (x - ich) * (idv + idv1 - 2 * idl) + ih * (idl - idv)
        b = ih * idv - (x - ich) * (idv + idv1 - 2 * idl)
        c = -idl * (x - ich)
        root = (2 * c) / (-b - torch.sqrt(b.pow(2) - 4 * a * c))
        out = root * iw + icw
        tt = root * (1 - root)
        den = idl + (idv + idv1 - 2 * idl) * tt
        num = idl.pow(2) * (idv1 * root.pow(2) + 2 * idl * tt + idv * (1 - root).pow(2))
        lad = -(torch.log(num) - 2 * torch.log(den))
    else:
        th = (x - icw) / iw
        tt = th * (1 - th)
        den = idl + (idv + idv1 - 2 * idl) * tt
        out = ich + ih * (idl * th.pow(2) + idv * tt) / den
        num = idl.pow(2) * (idv1 * th.pow(2) + 2 * idl * tt + idv * (1 - th).pow(2))
        lad = torch.log(num) - 2 * torch.log(den)
    return torch.where(inside, out, inputs), torch.where(inside, lad, torch.zeros_like(lad))


def dds_conv(sd, p, x, x_mask, g=None, n_layers=3, k=3):     # modules.py:95-108
    if g is not None:
        x = x + g
    C = x.size(1)
    for i in range(n_layers):
        dil = k ** i
        y = conv(sd, f"{p}.convs_sep.{i}", x * x_mask, padding=(k * dil - dil) // 2, dilation=dil, groups=C)
        y = F.gelu(layer_norm(sd, f"{p}.norms_1.{i}", y))
        y = conv(sd, f"{p}.convs_1x1.{i}", y)
        y = F.gelu(layer_norm(sd, f"{p}.norms_2.{i}", y))
        x = x + y
    return x * x_mask


def conv_flow(sd, p, x, x_mask, g, reverse, filter_channels, bins=10):   # modules.py:364-390
    x0, x1 = torch.split(x, [1, 1], 1)
    h = conv(sd, p + ".pre", x0)
    h = dds_conv(sd, p + ".convs", h, x_mask, g=g)
    h = conv(sd, p + ".proj", h) * x_mask
    b, c, t = x0.shape
    h = h.reshape(b, c, -1, t).permute(0, 1, 3, 2)
    uw, uh, ud = h[..., :bins] / math.sqrt(filter_channels), h[..., bins:2 * bins] / math.sqrt(filter_channels), h[..., 2 * bins:]
    x1, lad = rq_spline(x1, uw, uh, ud, reverse)
    x = torch.cat([x0, x1], 1) * x_mask
    return (x, torch.sum(lad * x_mask, [1, 2])) if not reverse else x


def affine(sd, p, x, x_mask, reverse=False):                 # modules.py:287-295
    m, logs = sd[p + ".m"], sd[p + ".logs"]
    if not reverse:
        return (m + torch.exp(logs) * x) * x_mask, torch.sum(logs * x_mask, [1, 2])
    return (x - m) * torch.exp(-logs) * x_mask


def sdp(sd, cfg, x, x_mask, w=None, g=None, reverse=False, noise_scale=1.0, eps=None):   # models.py:50-95
    C = cfg["hidden_channels"]
    x = conv(sd, "dp.pre", x.detach())
    if g is not None:
        x = x + conv(sd, "dp.cond", g.detach())
    x = dds_conv(sd, "dp.convs", x, x_mask)
    x = conv(sd, "dp.proj", x) * x_mask
    if not reverse:
        h_w = conv(sd, "dp.post_pre", w)
        h_w = dds_conv(sd, "dp.post_convs", h_w, x_mask)
        h_w = conv(sd, "dp.post_proj", h_w) * x_mask
        e_q = eps * x_mask
        z_q, ld_q = affine(sd, "dp.post_flows.0", e_q, x_mask)
        for i in range(4):
            z_q, ld = conv_flow(sd, f"dp.post_flows.{2 * i + 1}", z_q, x_mask, x + h_w, False, C)
            ld_q = ld_q + ld
            z_q = torch.flip(z_q, [1])
        z_u, z1 = torch.split(z_q, [1, 1], 1)
        u = torch.sigmoid(z_u) * x_mask
        z0 = (w - u) * x_mask
        ld_q = ld_q + torch.sum((F.logsigmoid(z_u) + F.logsigmoid(-z_u)) * x_mask, [1, 2])
        logq = torch.sum(-0.5 * (LOG_2PI + e_q ** 2) * x_mask, [1, 2]) - ld_q
        z0 = torch.log(torch.clamp_min(z0, 1e-5)) * x_mask                        # modules.Log
        ld_tot = torch.sum(-z0, [1, 2])
        z = torch.cat([z0, z1], 1)
        z, ld = affine(sd, "dp.flows.0", z, x_mask)
        ld_tot = ld_tot + ld
        for i in range(4):
            z, ld = conv_flow(sd, f"dp.flows.{2 * i + 1}", z, x_mask, x, False, C)
            ld_tot = ld_tot + ld
            z = torch.flip(z, [1])
        nll = torch.sum(0.5 * (LOG_2PI + z ** 2) * x_mask, [1, 2]) - ld_tot
        return nll + logq
    # reverse: flows reversed, minus the "useless vflow" = the first ConvFlow (models.py:87-89)
    z = eps * noise_scale
    for i in (3, 2, 1):
        z = torch.flip(z, [1])
        z = conv_flow(sd, f"dp.flows.{2 * i + 1}", z, x_mask, x, True, C)
    z = torch.flip(z, [1])                                     # Flip that followed the dropped ConvFlow
    z = affine(sd, "dp.flows.0", z, x_mask, reverse=True)
    return torch.split(z, [1, 1], 1)[0]


# ----------------------------------------------------------------------------- decoder
def generator(sd, cfg, x, g):                                # models.py:270-289, modules.py:211-223
    x = conv(sd, "dec.conv_pre", x, padding=3)
    if g is not None:
        x = x + conv(sd, "dec.cond", g)
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU)
        x = F.conv_transpose1d(x, _w(sd, f"dec.ups.{i}"), sd[f"dec.ups.{i}.bias"], u, (k - u) // 2)
        xs = 0
        for j, (rk, rd) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            p = f"dec.resblocks.{i * nk + j}"
            r = x
            if cfg["resblock"] == "1":
                for l, d in enumerate(rd):
                    t = conv(sd, f"{p}.convs1.{l}", F.leaky_relu(r, LRELU), padding=(rk * d - d) // 2, dilation=d)
                    t = conv(sd, f"{p}.convs2.{l}", F.leaky_relu(t, LRELU), padding=(rk - 1) // 2)
                    r = t + r
            else:
                for l, d in enumerate(rd):
                    r = conv(sd, f"{p}.convs.{l}", F.leaky_relu(r, LRELU), padding=(rk * d - d) // 2, dilation=d) + r
            xs = xs + r
        x = xs / nk
    x = F.leaky_relu(x)                                        # default slope 0.01, models.py:285
    return torch.tanh(conv(sd, "dec.conv_post", x, padding=3))


# ----------------------------------------------------------------------------- synthesizer
def neg_cent(z_p, m_p, logs_p):                              # models.py:470-477
    s = torch.exp(-2 * logs_p)
    return (torch.sum(-0.5 * LOG_2PI - logs_p, [1], keepdim=True) + torch.matmul(-0.5 * (z_p ** 2).transpose(1, 2), s)
            + torch.matmul(z_p.transpose(1, 2), m_p * s) + torch.sum(-0.5 * (m_p ** 2) * s, [1], keepdim=True))


def maximum_path(nc, mask):                                  # monotonic_align/__init__.py:6-19
    t_ys = mask.sum(1)[:, 0].cpu().numpy().astype(np.int32)
    t_xs = mask.sum(2)[:, 0].cpu().numpy().astype(np.int32)
    return torch.from_numpy(omas.mas_port(nc.detach().cpu().numpy(), t_ys, t_xs)).to(nc.dtype)


def slice_segments(x, ids, n):                               # commons.py:48-57
    return torch.stack([x[i, :, int(ids[i]):int(ids[i]) + n] for i in range(x.size(0))])


def synthesizer_forward(sd, cfg, seg, x, x_lengths, y, y_lengths, sid, noise):   # models.py:459-497
    h, m_p, logs_p, x_mask = text_encoder(sd, cfg, x, x_lengths)
    g = sd["emb_g.weight"][sid].unsqueeze(-1)
    z, m_q, logs_q, y_mask = posterior_encoder(sd, cfg, y, y_lengths, g, noise[0])
    z_p = coupling_block(sd, cfg, z, y_mask, g)
    with torch.no_grad():
        nc = neg_cent(z_p, m_p, logs_p)
        attn_mask = x_mask.unsqueeze(2) * y_mask.unsqueeze(-1)
        attn = maximum_path(nc, attn_mask.squeeze(1)).unsqueeze(1)
    w = attn.sum(2)
    l_length = sdp(sd, cfg, h, x_mask, w, g, eps=noise[1]) / torch.sum(x_mask)
    m_p = torch.matmul(attn.squeeze(1), m_p.transpose(1, 2)).transpose(1, 2)
    logs_p = torch.matmul(attn.squeeze(1), logs_p.transpose(1, 2)).transpose(1, 2)
    ids = (noise[2] * (y_lengths - seg + 1)).to(torch.long)   # commons.py:60-67
    o = generator(sd, cfg, slice_segments(z, ids, seg), g)
    return o, l_length, attn, ids, x_mask, y_mask, (z, z_p, m_p, logs_p, m_q, logs_q), nc


def generate_path(duration, mask):                           # commons.py:131-146
    b, _, t_y, t_x = mask.shape
    cum = torch.cumsum(duration, -1).view(b * t_x)
    path = sequence_mask(cum, t_y).to(mask.dtype).view(b, t_x, t_y)
    path = path - F.pad(path, (0, 0, 1, 0))[:, :-1]
    return path.unsqueeze(1).transpose(2, 3) * mask


def synthesizer_infer(sd, cfg, x, x_lengths, sid, noise, noise_scale=1.0, length_scale=1.0, noise_scale_w=1.0, max_len=None):
    h, m_p, logs_p, x_mask = text_encoder(sd, cfg, x, x_lengths)           # models.py:499-523
    g = sd["emb_g.weight"][sid].unsqueeze(-1)
    logw = sdp(sd, cfg, h, x_mask, g=g, reverse=True, noise_scale=noise_scale_w, eps=noise[0])
    w_ceil = torch.ceil(torch.exp(logw) * x_mask * length_scale)
    y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
    y_mask = sequence_mask(y_lengths).unsqueeze(1).to(x_mask.dtype)
    attn = generate_path(w_ceil, x_mask.unsqueeze(2) * y_mask.unsqueeze(-1))
    m_p = torch.matmul(attn.squeeze(1), m_p.transpose(1, 2)).transpose(1, 2)
    logs_p = torch.matmul(attn.squeeze(1), logs_p.transpose(1, 2)).transpose(1, 2)
    z_p = m_p + noise[1] * torch.exp(logs_p) * noise_scale
    z = coupling_block(sd, cfg, z_p, y_mask, g, reverse=True)
    o = generator(sd, cfg, (z * y_mask)[:, :, :max_len], g)
    return o, attn, y_mask, (z, z_p, m_p, logs_p)


# ----------------------------------------------------------------------------- discriminators + losses + mel
def disc_s(sd, p, x):                                        # models.py:338-361
    fmap = []
    for i, (pad, stride, groups) in enumerate([(7, 1, 1), (20, 4, 4), (20, 4, 16), (20, 4, 64), (20, 4, 256), (2, 1, 1)]):
        x = F.leaky_relu(conv(sd, f"{p}.convs.{i}", x, padding=pad, groups=groups, stride=stride), LRELU)
        fmap.append(x)
    x = conv(sd, f"{p}.conv_post", x, padding=1)
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def disc_p(sd, p, x, period):                                # models.py:314-335
    fmap = []
    b, c, t = x.shape
    if t % period != 0:
        n_pad = period - (t % period)
        x = F.pad(x, (0, n_pad), "reflect")
        t = t + n_pad
    x = x.view(b, c, t // period, period)
    for i, stride in enumerate([3, 3, 3, 3, 1]):
        x = F.leaky_relu(F.conv2d(x, _w(sd, f"{p}.convs.{i}"), sd[f"{p}.convs.{i}.bias"], (stride, 1), (2, 0)), LRELU)
        fmap.append(x)
    x = F.conv2d(x, _w(sd, f"{p}.conv_post"), sd[f"{p}.conv_post.bias"], 1, (1, 0))
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def mpd(sd, y, y_hat):                                       # models.py:372-386
    outs = ([], [], [], [])
    for i, period in enumerate([None, 2, 3, 5, 7, 11]):
        p = f"discriminators.{i}"
        f = (lambda t: disc_s(sd, p, t)) if period is None else (lambda t: disc_p(sd, p, t, period))
        r, fr = f(y)
        gq, fg = f(y_hat)
        outs[0].append(r); outs[1].append(gq); outs[2].append(fr); outs[3].append(fg)
    return outs


def feature_loss(fmap_r, fmap_g):                            # losses.py:7-15
    return 2 * sum(torch.mean(torch.abs(rl.detach() - gl)) for dr, dg in zip(fmap_r, fmap_g) for rl, gl in zip(dr, dg))


def discriminator_loss(dr, dg):                              # losses.py:18-32
    return sum(torch.mean((1 - r) ** 2) + torch.mean(g ** 2) for r, g in zip(dr, dg))


def generator_loss(dg):                                      # losses.py:35-43
    return sum(torch.mean((1 - g) ** 2) for g in dg)


def kl_loss(z_p, logs_q, m_p, logs_p, z_mask):               # losses.py:46-61
    kl = logs_p - logs_q - 0.5 + 0.5 * ((z_p - m_p) ** 2) * torch.exp(-2.0 * logs_p)
    return torch.sum(kl * z_mask) / torch.sum(z_mask)


def spectrogram(y, n_fft, hop, win):                         # mel_processing.py:51-70
    pad = int((n_fft - hop) / 2)
    y = F.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.stft(y, n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win, dtype=y.dtype, device=y.device),
                      center=False, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    return torch.sqrt(torch.view_as_real(spec).pow(2).sum(-1) + 1e-6)


def mel_basis_slaney(sr, n_fft, n_mels, fmin, fmax):
    """librosa==0.9.2 `filters.mel` defaults restated (Slaney scale, slaney norm); third-party
    arithmetic, no reference fixture: PARITY UNPINNED."""
    fmax = fmax or sr / 2.0

    def hz2mel(f):
        f = np.asarray(f, np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) / (np.log(6.4) / 27.0), f / (200.0 / 3))

    def mel2hz(m):
        m = np.asarray(m, np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), (200.0 / 3) * m)

    freqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - freqs[None, :]
    w = np.maximum(0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    w *= (2.0 / (mel_f[2:] - mel_f[:-2]))[:, None]
    return torch.from_numpy(w.astype(np.float32))


def spec_to_mel(spec, basis):                                # mel_processing.py:73-82
    return torch.log(torch.clamp(torch.matmul(basis, spec), min=1e-5))


def train_losses(sd_g, sd_d, cfg, hp, batch, noise):
    """One iteration's loss graph (finetune_speaker_v2.py:180-226) -> (loss_disc, loss_gen_all, parts)."""
    x, x_lengths, spec, spec_lengths, y, y_lengths, sid = batch
    seg = hp["segment_size"] // hp["hop_length"]
    o, l_length, attn, ids, x_mask, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), _ = synthesizer_forward(
        sd_g, cfg, seg, x, x_lengths, spec, spec_lengths, sid, noise)
    basis = mel_basis_slaney(hp["sampling_rate"], hp["filter_length"], hp["n_mel_channels"], hp["mel_fmin"], hp["mel_fmax"])
    y_mel = slice_segments(spec_to_mel(spec, basis), ids, seg)
    y_hat_mel = spec_to_mel(spectrogram(o.squeeze(1), hp["filter_length"], hp["hop_length"], hp["win_length"]), basis)
    y_seg = slice_segments(y, ids * hp["hop_length"], hp["segment_size"])
    dr, dg, _, _ = mpd(sd_d, y_seg, o.detach())
    loss_disc = discriminator_loss(dr, dg)
    return loss_disc, (o, y_seg, y_mel, y_hat_mel, l_length, z_p, logs_q, m_p, logs_p, z_mask)


def generator_losses(sd_d, hp, o, y_seg, y_mel, y_hat_mel, l_length, z_p, logs_q, m_p, logs_p, z_mask):
    dr, dg, fr, fg = mpd(sd_d, y_seg, o)
    parts = dict(loss_dur=torch.sum(l_length), loss_mel=F.l1_loss(y_mel, y_hat_mel) * hp["c_mel"],
                 loss_kl=kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * hp["c_kl"], loss_fm=feature_loss(fr, fg),
                 loss_gen=generator_loss(dg))
    return sum(parts.values()), parts
