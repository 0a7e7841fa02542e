"""Operator layer of the hot path: every function here is what the nn.Modules call.

Each op is either
  * HIP  — a hand-written gfx950 kernel reached through the C ABI of libvitsmi.so (include/vitsmi.h),
           wrapped in a torch.autograd.Function where it needs a backward; or
  * ROCm — a composition of PyTorch-ROCm device ops (element-wise glue, rocBLAS for two small matmuls, the fused
           AdamW).  DESIGN.md §4 lists which is which; `BACKENDS` below is the same table in code, and bench.py
           prints it.  `conv1d` / `conv_transpose1d` here are CPU-side utilities for generic module calls (state_dict and
           shape helpers, the host-logic tests): they RAISE on GPU tensors — no library convolution is on the hot path.
There is no CPU implementation of the HIP ops: they raise on non-GPU tensors.
"""
import math

import os

import torch
from torch.nn import functional as F

from . import _lib
from . import monotonic_align as _ma

BACKENDS = {
    "maximum_path": "hip",                # csrc/mas.hip (vits_mas_f32)
    "conv1d (channels-last, fused epilogues: generator, text encoder, STFT/mel, discriminators)": "hip",   # conv1d_cl.hip, conv1d_flat.hip
    "conv1d weight+bias gradient": "hip",  # conv1d_cl_wgrad.hip
    "conv_transpose1d": "hip",            # 1x1 product on conv1d_cl + convt_fold.hip
    "weight_norm (+autograd), operand layouts": "hip",   # weight_prep.hip, one launch per network
    "layer_norm+gelu, depthwise conv": "hip",             # rowops.hip
    "relative-position attention": "hip",  # attn_softmax.hip + batched products on conv1d_cl
    "rational_quadratic_spline (+autograd)": "hip",       # rq_spline.hip
    "feature-matching / KL / duration sums, bias + conditioning column sums": "hip",   # reduce.hip
    "element-wise glue, adversarial loss arithmetic, embedding lookups, mel matmul": "rocm",   # PyTorch-ROCm device ops
    "adamw + gradient L2 norm (flat buffers, multi-tensor)": "hip",   # adamw.hip (optim.FlatAdamW)
    "neg_cent (alignment scores), slice_segments (+autograd), generate_path": "hip",   # align.hip
}


# ------------------------------------------------------------------ alignment (HIP)
def maximum_path(neg_cent, mask):
    return _ma.maximum_path(neg_cent, mask)


# ------------------------------------------------------------------ convolutions
def neg_cent(z_p, m_p, logs_p):
    """models.py:470-477 as ONE launch (csrc/align.hip, vits_neg_cent): z_p [b, c, t_t], m_p / logs_p [b, c, t_s] in the
    reference's layout — here transposed views of channels-last tensors (the prior statistics may be the two halves of one
    projection output), fp32 or bf16 — -> fp32 [b, t_t, t_s].  No gradient (the reference builds it under no_grad)."""
    _lib.require_cuda(z_p, m_p, logs_p)
    dt = {torch.float32: 0, torch.bfloat16: 2}
    rows = lambda t: t.transpose(1, 2) if t.stride(1) == 1 else t.transpose(1, 2).contiguous()     # -> [b][t][c], unit channel stride
    z, m, l = rows(z_p.detach()), rows(m_p.detach()), rows(logs_p.detach())
    if l.dtype != m.dtype:
        l = l.to(m.dtype)
    b, t_t, c = z.shape
    t_s = m.size(1)
    for t in (z, m, l):
        if t.stride(0) != t.size(1) * t.stride(1):
            raise ValueError("neg_cent: items must lie back to back")
    if m.stride(1) != l.stride(1):
        m, l = m.contiguous(), l.contiguous()
    out = torch.empty(b, t_t, t_s, dtype=torch.float32, device=z.device)
    e0 = _lib.timer.start("vits_neg_cent")
    rc = _lib.lib().vits_neg_cent(dt[z.dtype], z.data_ptr(), z.stride(1), dt[m.dtype], m.data_ptr(), l.data_ptr(), m.stride(1),
                                  out.data_ptr(), b, t_t, t_s, c, _lib.stream_ptr())
    _lib.timer.stop("vits_neg_cent", e0, (4.0 * b * t_t * t_s * c, z.element_size() * b * t_t * c + 2.0 * m.element_size() * b * t_s * c + 4.0 * b * t_t * t_s))
    _lib.check(rc, "vits_neg_cent")
    return out


class _SliceSegments(torch.autograd.Function):
    """commons.slice_segments on the GPU (csrc/align.hip): one launch forward, one launch backward (which writes the whole
    gradient, zeros included)."""

    @staticmethod
    def forward(ctx, x, ids, seg, mul):
        # x [b, d, t]: either contiguous (time innermost) or a transposed view of a channels-last [b, t, d] tensor
        rows = x.stride(1) == 1 and x.transpose(1, 2).is_contiguous() and x.size(1) > 1
        if not rows and not x.is_contiguous():
            x = x.contiguous()
        b, d, t = x.shape
        ids = ids.to(torch.int64).contiguous()
        y = torch.empty((b, seg, d) if rows else (b, d, seg), dtype=x.dtype, device=x.device)
        rc = _lib.lib().vits_slice_segments(x.element_size(), 0 if rows else 1, x.data_ptr(), ids.data_ptr(), mul, y.data_ptr(), b, d, t, seg, 0,
                                            _lib.stream_ptr())
        _lib.check(rc, "vits_slice_segments")
        ctx.save_for_backward(ids)
        ctx.geom = (rows, b, d, t, seg, mul)
        return y.transpose(1, 2) if rows else y

    @staticmethod
    def backward(ctx, dy):
        rows, b, d, t, seg, mul = ctx.geom
        ids, = ctx.saved_tensors
        dy = (dy.transpose(1, 2) if rows else dy).contiguous()
        dx = torch.empty((b, t, d) if rows else (b, d, t), dtype=dy.dtype, device=dy.device)
        rc = _lib.lib().vits_slice_segments(dy.element_size(), 0 if rows else 1, dy.data_ptr(), ids.data_ptr(), mul, dx.data_ptr(), b, d, t, seg, 1,
                                            _lib.stream_ptr())
        _lib.check(rc, "vits_slice_segments")
        return (dx.transpose(1, 2) if rows else dx), None, None, None


def slice_segments(x, ids_str, segment_size, ids_scale=1):
    _lib.require_cuda(x, ids_str)
    if x.element_size() not in (2, 4):
        raise ValueError("slice_segments: 2- or 4-byte elements")
    return _SliceSegments.apply(x, ids_str, int(segment_size), int(ids_scale))


def generate_path(duration, mask):
    """commons.generate_path on the GPU (csrc/align.hip): duration [b, 1, t_x], mask [b, 1, t_y, t_x] -> path of mask's shape."""
    _lib.require_cuda(duration, mask)
    b, _, t_y, t_x = mask.shape
    d = duration.detach().reshape(b, t_x).float().contiguous()
    m = mask.detach().reshape(b, t_y, t_x).float().contiguous()
    path = torch.empty_like(m)
    _lib.check(_lib.lib().vits_generate_path(d.data_ptr(), m.data_ptr(), path.data_ptr(), b, t_y, t_x, _lib.stream_ptr()), "vits_generate_path")
    return path.view(b, 1, t_y, t_x).to(mask.dtype)


def weight_norm(v, g):
    """w = g * v / ||v||_2, norm over every dim but 0 (legacy torch.nn.utils.weight_norm, dim=0;
    reference models.py:254, modules.py:128,135,145,191-206).  For ConvTranspose1d weights
    [c_in, c_out, k] dim 0 is the INPUT channel."""
    norm = torch.linalg.vector_norm(v.float(), 2, dim=tuple(range(1, v.dim())), keepdim=True)
    return (v.float() * (g.float() / norm)).to(v.dtype)


def _no_library_conv(x):
    if x.is_cuda:
        raise RuntimeError("library convolution reached on a GPU tensor: every convolution of the hot path runs through "
                           "libvitsmi.so (wn_cl.conv_cl / decoder_cl / disc_cl); generic nn.Module calls are CPU-side utilities only")


def conv1d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """Generic module call (state_dict / shape utilities on the CPU).  Raises on GPU tensors: nothing on the hot path may
    fall back to MIOpen (its bf16 backward-data solver for c_in = 1 faults on gfx950 / ROCm 7.2)."""
    _no_library_conv(x)
    return F.conv1d(x, weight.to(x.dtype), None if bias is None else bias.to(x.dtype), stride, padding, dilation, groups)


def conv_transpose1d(x, weight, bias=None, stride=1, padding=0):
    _no_library_conv(x)
    return F.conv_transpose1d(x, weight.to(x.dtype), None if bias is None else bias.to(x.dtype), stride, padding)


def leaky_relu(x, slope):
    return F.leaky_relu(x, slope)


def wn_gate(x_in, g_l, n_channels):
    """tanh((a+b)[:, :C]) * sigmoid((a+b)[:, C:]) (reference commons.py:103-110)."""
    if g_l is not None:
        x_in = x_in + g_l
    return torch.tanh(x_in[:, :n_channels]) * torch.sigmoid(x_in[:, n_channels:])


# ------------------------------------------------------------------ normalisation
def layer_norm_c(x, gamma, beta, eps):
    """LayerNorm over the CHANNEL dim of [b, c, t] (reference modules.py:20-32)."""
    xt = x.transpose(1, -1)
    xt = F.layer_norm(xt.float(), (x.size(1),), gamma.float(), beta.float(), eps).to(x.dtype)
    return xt.transpose(1, -1)


# ------------------------------------------------------------------ relative-position attention
def rel_attention(q, k, v, emb_rel_k, emb_rel_v, mask, n_heads, window_size, p_dropout=0.0, training=False):
    """Windowed relative-position self-attention (reference attentions.py:150-182).

    q, k, v: [b, d, t];  emb_rel_*: [1, 2w+1, d/h] (shared by the heads);  mask: [b, 1, t, t] or None.
    scores[i,j] = q_i.k_j/sqrt(dk) + (|j-i| <= w) q_i.E_k[j-i+w]/sqrt(dk);  masked_fill(mask==0, -1e4);
    out_i = sum_j p_ij v_j + sum_{|r|<=w} p_{i,i+r} E_v[r+w].
    The reference materialises the band with pad/reshape skews (attentions.py:199-243); here the
    band is gathered/scattered directly through a [t, t] relative-index map.
    Returns (out [b, d, t], p_attn [b, h, t, t]).
    """
    b, d, t = q.shape
    dk = d // n_heads
    qh = q.view(b, n_heads, dk, t).transpose(2, 3) / math.sqrt(dk)       # [b,h,t,dk]
    kh = k.view(b, n_heads, dk, t).transpose(2, 3)
    vh = v.view(b, n_heads, dk, t).transpose(2, 3)
    scores = torch.matmul(qh, kh.transpose(-2, -1))                       # [b,h,t,t]
    pos = torch.arange(t, device=q.device)
    rel = pos[None, :] - pos[:, None] + window_size                       # [t,t] index into 2w+1
    band = (rel >= 0) & (rel <= 2 * window_size)
    relc = rel.clamp(0, 2 * window_size)
    rel_logits = torch.matmul(qh, emb_rel_k.to(q.dtype).unsqueeze(0).transpose(-2, -1))   # [b,h,t,2w+1]
    scores = scores + torch.where(band, rel_logits.gather(-1, relc.expand(b, n_heads, t, t)), torch.zeros((), dtype=q.dtype, device=q.device))
    if mask is not None:
        scores = scores.masked_fill(mask == 0, -1e4)
    p_attn = F.softmax(scores, dim=-1)
    p_drop = F.dropout(p_attn, p_dropout, training) if p_dropout > 0 else p_attn
    out = torch.matmul(p_drop, vh)
    # relative values: weights[b,h,i,r] = p[i, i+r-w]
    w_rel = torch.zeros(b, n_heads, t, 2 * window_size + 1, dtype=q.dtype, device=q.device)
    w_rel = w_rel.scatter_add(-1, relc.expand(b, n_heads, t, t), p_drop * band.to(q.dtype))
    out = out + torch.matmul(w_rel, emb_rel_v.to(q.dtype).unsqueeze(0))
    out = out.transpose(2, 3).contiguous().view(b, d, t)
    return out, p_attn


# ------------------------------------------------------------------ spline
def rq_spline(inputs, uw, uh, ud, inverse, tail_bound, min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3):
    from .transforms import rq_spline_torch
    return rq_spline_torch(inputs.float(), uw.float(), uh.float(), ud.float(), inverse, tail_bound,
                           min_bin_width, min_bin_height, min_derivative)


# ------------------------------------------------------------------ STFT
_dft_cache = {}


def _dft_operand(n_fft, hop, window):
    """Windowed real-DFT basis as a tap-major convolution operand [n_fft/hop][2*Fp][hop] (fp32):
    rows [0, F) = w[n] cos(2 pi c n / n_fft), rows [Fp, Fp+F) = -w[n] sin(...), F = n_fft/2+1, Fp = F rounded up to 8."""
    key = (n_fft, hop, window.data_ptr(), str(window.device))
    op = _dft_cache.get(key)
    if op is None:
        from . import weight_arena
        F_, taps = n_fft // 2 + 1, n_fft // hop
        Fp = (F_ + 7) // 8 * 8
        n = torch.arange(n_fft, dtype=torch.float64, device=window.device)
        c = torch.arange(F_, dtype=torch.float64, device=window.device)
        ang = 2 * math.pi * c[:, None] * n[None, :] / n_fft
        w64 = window.double()[None, :]
        basis = torch.zeros(2 * Fp, n_fft, dtype=torch.float64, device=window.device)
        basis[:F_] = torch.cos(ang) * w64
        basis[Fp:Fp + F_] = -torch.sin(ang) * w64
        op = basis.view(2 * Fp, taps, hop).permute(1, 0, 2).contiguous().float()        # [taps][2Fp][hop]
        weight_arena.register_constant(op)
        _dft_cache[key] = op
    return op


def _stft_ri(y, n_fft, hop, win, window, prepadded=False):
    """The framed windowed real DFT of the reflect-padded signal as an exact-fp32 matrix-core product: the padded signal is
    viewed as rows of `hop` samples, a frame is n_fft/hop consecutive rows, so the STFT is vits_conv1d_cl with k = n_fft/hop
    taps over a windowed cos/sin basis (fp32 MFMA = fmaf chain).  -> (ri [b, frames, 2*Fp] fp32: re | im, F, Fp), or None when
    hop does not divide n_fft / the window is not n_fft long / the signal is on the host (the caller then uses torch.stft)."""
    if not (y.is_cuda and n_fft % hop == 0 and win == n_fft and hop % 4 == 0):
        return None
    from . import wn_cl
    pad = int((n_fft - hop) / 2)
    yp = y if prepadded else F.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    b, tp = yp.shape
    taps = n_fft // hop
    frames = (tp - n_fft) // hop + 1
    rows = frames + taps - 1
    need = rows * hop
    yp = yp.float()
    if tp < need:
        yp = F.pad(yp, (0, need - tp))
    x = yp[:, :need].contiguous().reshape(b, rows, hop)            # (a no-op unless the padded signal is longer than the frames cover)
    op = _dft_operand(n_fft, hop, window.float())
    ri = wn_cl.conv_cl(x, op, dtype=torch.float32)                     # [b, frames, 2*Fp]
    return ri, n_fft // 2 + 1, op.size(1) // 2


def stft_magnitude(y, n_fft, hop, win, window, prepadded=False):
    """reflect-pad (n_fft-hop)/2, framed windowed real DFT, sqrt(re^2+im^2+1e-6)
    (reference mel_processing.py:63-69).  y [b, t] -> [b, n_fft/2+1, frames].  On the GPU the DFT is _stft_ri's matrix-core
    product; otherwise torch.stft (rocFFT / host)."""
    got = _stft_ri(y, n_fft, hop, win, window, prepadded)
    if got is None:
        pad = int((n_fft - hop) / 2)
        yp = y if prepadded else F.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
        spec = torch.stft(yp, n_fft, hop_length=hop, win_length=win, window=window, center=False,
                          normalized=False, onesided=True, return_complex=True)
        return torch.sqrt(spec.real.pow(2) + spec.imag.pow(2) + 1e-6)
    ri, F_, Fp = got
    mag = torch.sqrt(ri[..., :F_].pow(2) + ri[..., Fp:Fp + F_].pow(2) + 1e-6)
    return mag.transpose(1, 2)


class _StftMel(torch.autograd.Function):
    """log(clamp(basis @ sqrt(re^2 + im^2 + 1e-6), clip)) of a DFT product ri [b, frames, 2*Fp] -> [b, M, frames]
    (csrc/stft_mel.hip: one launch per direction; reference mel_processing.py:63-69, 73-82)."""

    @staticmethod
    def forward(ctx, ri, basis, F_, Fp, clip):
        _lib.require_cuda(ri, basis)
        assert ri.dtype == torch.float32 and ri.is_contiguous() and basis.dtype == torch.float32 and basis.is_contiguous()
        b, frames, ld = ri.shape
        M = basis.size(0)
        assert basis.size(1) == F_ and ld >= 2 * Fp
        mel = torch.empty((b, M, frames), device=ri.device, dtype=torch.float32)
        lin = torch.empty_like(mel)
        rc = _lib.lib().vits_stft_mel_fwd(ri.data_ptr(), basis.data_ptr(), mel.data_ptr(), lin.data_ptr(), b * frames, frames, F_, Fp, ld, M,
                                          float(clip), _lib.stream_ptr())
        _lib.check(rc, "vits_stft_mel_fwd")
        ctx.save_for_backward(ri, basis, lin)
        ctx.geom = (F_, Fp, float(clip))
        return mel

    @staticmethod
    def backward(ctx, dmel):
        ri, basis, lin = ctx.saved_tensors
        F_, Fp, clip = ctx.geom
        b, frames, ld = ri.shape
        dmel = dmel.float().contiguous()
        dri = torch.empty_like(ri)
        rc = _lib.lib().vits_stft_mel_bwd(ri.data_ptr(), basis.data_ptr(), lin.data_ptr(), dmel.data_ptr(), dri.data_ptr(), b * frames, frames,
                                          F_, Fp, ld, basis.size(0), clip, _lib.stream_ptr())
        _lib.check(rc, "vits_stft_mel_bwd")
        return dri, None, None, None, None


def stft_mel(y, n_fft, hop, win, window, basis, clip=1e-5):
    """mel_spectrogram of the reference (mel_processing.py:85-112) on the GPU: DFT product + ONE fused magnitude / filter-bank /
    log-clamp launch (and one for its backward).  None when the DFT product does not apply (the caller composes torch ops)."""
    got = _stft_ri(y, n_fft, hop, win, window)
    if got is None:
        return None
    ri, F_, Fp = got
    return _StftMel.apply(ri.contiguous(), basis.float().contiguous(), F_, Fp, clip)


# ================================================================================================
# Channels-last HIP convolution (csrc/conv1d_cl.hip).  Raw launcher: tensors are [b, t, c]
# contiguous, weights are tap-major [k, c_out, c_in] in the activation dtype.
# ================================================================================================
CONV_MASK_IN, CONV_MASK_OUT, CONV_TANH, CONV_ACCUM, CONV_RES_AFTER, CONV_GATE, CONV_GATE_BWD, CONV_OUT_LRELU = 1, 2, 4, 8, 16, 32, 64, 128
CONV_FLAT = 256
CONV_BIG_TILES = 512
CONV_RES_SKIP = 1024
_DT = {torch.float32: 0, torch.bfloat16: 2}


def _rows(t, name):
    """A [b, t, c] tensor whose rows are dense in c and whose batches follow each other: returns its
    row pitch.  Covers contiguous tensors and channel slices x[..., a:b] of contiguous tensors."""
    assert t.dim() == 3 and t.stride(2) == 1 and (t.size(0) == 1 or t.stride(0) == t.size(1) * t.stride(1)), \
        f"{name}: unsupported strides {t.stride()} for shape {tuple(t.shape)}"
    return t.stride(1)


def _conv_desc(x, w, bias=None, bias_b=None, res=None, mg_src=None, out=None, lengths=None, dil=1, pad=0, stride=1,
               in_slope=1.0, mg_slope=1.0, out_scale=1.0, flags=0, gate_h=0, out2=None, out_slope=None, in_div=1, t_out=None, groups=1):
    """-> (vits_conv_desc, y, (FLOP, algorithmic bytes), shape tag) of one vits_conv1d_cl launch; y is allocated unless `out` is given."""
    _lib.require_cuda(x, w)
    assert x.dtype == w.dtype
    b, t, c_in = x.shape
    ldw, wbs = 0, 0
    if w.dim() == 4:                       # per-item operand [b][k=1][c_out][c_in]: batched product Y[b] = X[b] . W[b]^T
        assert w.size(0) == b and w.size(1) == 1 and w.stride(3) == 1
        ldw, wbs = w.stride(2), w.stride(0)
        k, c_out, c_in_w = 1, w.size(2), w.size(3)
    else:
        assert w.dim() == 3 and w.is_contiguous()
        k, c_out, c_in_w = w.shape
    assert c_in_w == c_in, (tuple(x.shape), tuple(w.shape))
    if in_div > 1:                         # data gradient of a stride-`in_div` convolution: the caller states the length
        assert stride == 1 and t_out is not None
    else:
        t_out = (t + 2 * pad - dil * (k - 1) - 1) // stride + 1
    y_cols = gate_h if (flags & (CONV_GATE | CONV_RES_SKIP)) else (2 * gate_h if (flags & CONV_GATE_BWD) else c_out)
    if flags & CONV_RES_SKIP:               # res half -> y (+ res), skip half (+)= into out2; ACCUM is about out2
        assert c_out == 2 * gate_h and out2 is not None and tuple(out2.shape) == (b, t_out, gate_h) and res is not None and lengths is not None
    if out is None:
        assert not (flags & CONV_ACCUM) or (flags & CONV_RES_SKIP)
        out = torch.empty((b, t_out, y_cols), device=x.device, dtype=x.dtype)
    assert out.dtype == x.dtype and tuple(out.shape) == (b, t_out, y_cols)
    ldy = _rows(out, "out")
    for tns, nm in ((res, "res"), (mg_src, "mg_src")):
        assert tns is None or (tns.dtype == x.dtype and tns.shape[:2] == out.shape[:2] and _rows(tns, nm) == ldy), nm
    for tns in (bias, bias_b):
        assert tns is None or (tns.dtype == torch.float32 and tns.is_contiguous())
    assert lengths is None or lengths.dtype == torch.int32
    p = lambda v: None if v is None else v.data_ptr()
    if out_slope is not None:
        flags |= CONV_OUT_LRELU
    d = _lib.ConvDesc(dtype=_DT[x.dtype], b=b, t=t, c_in=c_in, c_out=c_out, k=k, dil=dil, pad=pad, stride=stride, flags=int(flags),
                      ldx=_rows(x, "x"), ldy=ldy, ldy2=0 if out2 is None else _rows(out2, "out2"), gate_h=gate_h,
                      ldw=ldw, in_div=in_div, t_out_override=(t_out if in_div > 1 else 0), groups=groups, w_batch_stride=wbs,
                      in_slope=float(in_slope), mg_slope=float(mg_slope), out_scale=float(out_scale), out_slope=float(out_slope or 0.0),
                      x=x.data_ptr(), w=w.data_ptr(), bias=p(bias), bias_b=p(bias_b), res=p(res), mg_src=p(mg_src),
                      y=out.data_ptr(), y2=p(out2), lengths=p(lengths))
    es = x.element_size()                                      # units = (FLOP, algorithmic bytes: x + y + w read/written once)
    units = (2.0 * b * t_out * c_out * (c_in // groups) * k,
             es * (b * t * c_in + b * t_out * out.size(2) + k * c_out * c_in + (0 if res is None else res.numel())))
    return d, out, units, f"b{b} t{t} ci{c_in} co{c_out} k{k} d{dil} s{stride}/{in_div} f{flags} {str(x.dtype)[6:]}"


def conv1d_cl_raw(x, w, *args, **kw):
    """Launch vits_conv1d_cl.  x [b,t,c_in], w [k,c_out,c_in] (tap-major) in the same dtype; see
    include/vitsmi.h for the fused prologue/epilogue (keywords: _conv_desc).  Returns y (allocated unless `out` is given)."""
    import ctypes
    d, out, units, shape = _conv_desc(x, w, *args, **kw)
    e0 = _lib.timer.start("vits_conv1d_cl")
    rc = _lib.lib().vits_conv1d_cl(ctypes.addressof(d), _lib.stream_ptr())
    _lib.timer.stop("vits_conv1d_cl", e0, units, shape=shape)
    _lib.check(rc, "vits_conv1d_cl")
    return out


MULTI_LAUNCH = os.environ.get("VITS_NO_MULTI", "0") == "0"        # (measurement switch: off = always one launch per call)


def conv1d_cl_multi(calls):
    """Several independent vits_conv1d_cl launches — calls[i] = (x, w, keywords of conv1d_cl_raw) — as ONE launch where the
    library can place them side by side (vits_conv1d_cl_multi: the same layer of the period discriminators), else one after
    the other.  Returns the list of outputs; bitwise the results of the separate calls either way."""
    import ctypes
    built = [_conv_desc(x, w, **kw) for x, w, kw in calls]
    n = len(built)
    L, s = _lib.lib(), _lib.stream_ptr()
    if MULTI_LAUNCH and 2 <= n <= 8:
        arr = (_lib.ConvDesc * n)(*[b[0] for b in built])
        e0 = _lib.timer.start("vits_conv1d_cl")
        rc = L.vits_conv1d_cl_multi(ctypes.addressof(arr), n, s)
        if rc != _lib.E_UNSUPPORTED:
            _lib.timer.stop("vits_conv1d_cl", e0, tuple(sum(u) for u in zip(*[b[2] for b in built])), shape=f"multi x{n}: {built[0][3]}")
            _lib.check(rc, "vits_conv1d_cl_multi")
            return [b[1] for b in built]
    for d, out, units, shape in built:
        e0 = _lib.timer.start("vits_conv1d_cl")
        rc = L.vits_conv1d_cl(ctypes.addressof(d), s)
        _lib.timer.stop("vits_conv1d_cl", e0, units, shape=shape)
        _lib.check(rc, "vits_conv1d_cl")
    return [b[1] for b in built]


class WnPacked:
    """The operands of a WaveNet stack in the fragment order vits_wn_layer_fwd / _bwd read (vits_wn_pack, csrc/wn_layer.hip):
    one persistent buffer per (stack, dtype), re-filled by ONE launch per forward (the weights change with every optimizer
    step); the backward of the same step reads the data-gradient operands packed by that launch."""

    def __init__(self, H, k, L, dtype, device):
        lib = _lib.lib()
        dt = _DT[dtype]
        self.H, self.k, self.L, self.dtype = H, k, L, dtype
        self.es = torch.empty((), dtype=dtype).element_size()
        size = lambda mode, rows, kel, taps: lib.vits_wn_pack_bytes(mode, dt, H, rows, kel, taps)
        self.off, total = [], 0
        for i in range(L):
            c_rs = H if i == L - 1 else 2 * H
            sizes = (size(0, 2 * H, H, k), size(1, c_rs, H, 1), size(2, H, c_rs, 1), size(2, H, 2 * H, k))
            offs = []
            for n in sizes:
                offs.append(total)
                total += n
            self.off.append(offs)
        self.buf = torch.empty(total, dtype=torch.uint8, device=device)
        self.with_bwd = False

    def ptr(self, layer, which):
        return self.buf.data_ptr() + self.off[layer][which]

    def fill(self, fwd_ops, bwd_ops):
        """fwd_ops[i] = (w_in [k][2H][H], w_rs [1][2H|H][H]); bwd_ops[i] = (w_rs_t [1][H][2H|H], w_in_t [k][H][2H]) or None."""
        import ctypes
        H, k, es = self.H, self.k, self.es
        segs = []
        for i, (w_in, w_rs) in enumerate(fwd_ops):
            c_rs = w_rs.size(1)
            assert w_in.is_contiguous() and w_rs.is_contiguous() and tuple(w_in.shape) == (k, 2 * H, H) and w_rs.size(2) == H
            segs.append((w_in.data_ptr(), self.ptr(i, 0), 0, H, 2 * H, H * es, k, H * es // 32))
            segs.append((w_rs.data_ptr(), self.ptr(i, 1), 1, H, c_rs, H * es, 1, H * es // 32))
            if bwd_ops is not None:
                w_rs_t, w_in_t = bwd_ops[i]
                assert w_rs_t.is_contiguous() and w_in_t.is_contiguous() and tuple(w_in_t.shape) == (k, H, 2 * H) and tuple(w_rs_t.shape) == (1, H, c_rs)
                segs.append((w_rs_t.data_ptr(), self.ptr(i, 2), 2, H, H, c_rs * es, 1, (c_rs * es + 63) // 64))
                segs.append((w_in_t.data_ptr(), self.ptr(i, 3), 2, H, H, 2 * H * es, k, 2 * H * es // 64))
        arr = (_lib.WnPackSeg * len(segs))()
        for a, (src, dst, mode, h, rows, rowbytes, taps, spt) in zip(arr, segs):
            a.src, a.dst, a.mode, a.h, a.rows, a.rowbytes, a.taps, a.spt = src, dst, mode, h, rows, rowbytes, taps, spt
        _lib.check(_lib.lib().vits_wn_pack(ctypes.addressof(arr), len(segs), _lib.stream_ptr()), "vits_wn_pack")
        self.with_bwd = bwd_ops is not None


def wn_fusable(H, k, dil_max=1, es=2):
    """Shapes vits_wn_layer_fwd / _bwd take (else the caller composes the layer from convolution launches): the gate
    interleave's granularity, 4 waves x 3 column tiles, and a (k-1)*dil halo that fits the staged row tile (64 rows in bf16, 32
    in fp32) and the forward's one-batch staging of it."""
    rows = 64 if es == 2 else 32
    halo = (k - 1) * dil_max
    return H % 16 == 0 and H <= 192 and k % 2 == 1 and halo < rows and (rows + halo) * (H * es // 16) <= 2560


def wn_layer_fwd(x, packed, layer, b_in, cond, b_rs, lengths, dil, skip, accumulate, last, pre=None, acts=None):
    """One WaveNet layer in one launch (csrc/wn_layer.hip, vits_wn_layer_fwd; reference modules.py:157-176): x [b,t,H] with rows
    >= lengths zero, `packed` the stack's WnPacked operands (filled this forward), b_in / b_rs fp32, cond fp32 [b][2H] or None.
    Writes pre [b,t,2H] / acts [b,t,H] when given, accumulates the skip half into `skip`, returns h_out (None for the last
    layer)."""
    _lib.require_cuda(x, skip)
    b, t, H = x.shape
    k = packed.k
    assert wn_fusable(H, k) and packed.H == H and packed.dtype == x.dtype
    for tns in (b_in, b_rs, cond):
        assert tns is None or (tns.dtype == torch.float32 and tns.is_contiguous())
    assert cond is None or tuple(cond.shape) == (b, 2 * H)
    assert lengths is None or lengths.dtype == torch.int32
    h_out = None if last else torch.empty((b, t, H), device=x.device, dtype=x.dtype)
    for tns, c in ((pre, 2 * H), (acts, H), (skip, H)):
        assert tns is None or (tns.dtype == x.dtype and tuple(tns.shape) == (b, t, c))
    p = lambda v: None if v is None else v.data_ptr()
    d = _lib.WnLayerDesc(dtype=_DT[x.dtype], b=b, t=t, h=H, k=k, dil=dil, last=int(bool(last)), accumulate=int(bool(accumulate)),
                         ldx=_rows(x, "x"), ldh=H, ldskip=_rows(skip, "skip"), ldacts=0 if acts is None else _rows(acts, "acts"),
                         ldpre=0 if pre is None else _rows(pre, "pre"),
                         x=x.data_ptr(), w_in=packed.ptr(layer, 0), b_in=p(b_in), cond=p(cond), w_rs=packed.ptr(layer, 1), b_rs=p(b_rs),
                         pre=p(pre), acts=p(acts), h_out=p(h_out), skip=skip.data_ptr(), lengths=p(lengths))
    import ctypes
    e0 = _lib.timer.start("vits_wn_layer_fwd")
    rc = _lib.lib().vits_wn_layer_fwd(ctypes.addressof(d), _lib.stream_ptr())
    if e0 is not None:
        es = x.element_size()
        c_rs = H if last else 2 * H
        _lib.timer.stop("vits_wn_layer_fwd", e0, (2.0 * b * t * H * (2 * H * k + c_rs),
                                                   es * (b * t * H * (2 + (0 if last else 1) + (1 if accumulate else 0) + (2 if pre is not None else 0)
                                                                      + (1 if acts is not None else 0)) + (2 * k + c_rs // H) * H * H)),
                        shape=f"b{b} t{t} H{H} k{k} d{dil} last{int(bool(last))} {str(x.dtype)[6:]}")
    _lib.check(rc, "vits_wn_layer_fwd")
    return h_out


def wn_layer_bwd(d_h, d_o, pre, packed, layer, lengths, dil, last, d_pre, d_h_out):
    """Data-gradient half of one WaveNet layer's backward in one launch (csrc/wn_layer.hip, vits_wn_layer_bwd):
    d_pre = gate'(pre) * ([d_h | d_o] . W_rs) (masked) into `d_pre` [b,t,2H]; d_h_out = (d_h + conv^T(d_pre; W_in)) * mask.
    d_h / d_o / d_h_out [b,t,H] may be column slices of wider tensors; `packed` holds the data-gradient operands packed by this
    step's forward (WnPacked.fill)."""
    _lib.require_cuda(d_o, pre, d_pre, d_h_out)
    b, t, H = d_o.shape
    k = packed.k
    dt = d_o.dtype
    assert wn_fusable(H, k) and packed.H == H and packed.dtype == dt and packed.with_bwd
    assert all(x.dtype == dt for x in (pre, d_pre, d_h_out)) and (d_h is None or d_h.dtype == dt)
    assert tuple(pre.shape) == (b, t, 2 * H) and tuple(d_pre.shape) == (b, t, 2 * H) and tuple(d_h_out.shape) == (b, t, H)
    assert last or tuple(d_h.shape) == (b, t, H)
    assert lengths is None or lengths.dtype == torch.int32
    p = lambda v: None if v is None else v.data_ptr()
    d = _lib.WnLayerBwdDesc(dtype=_DT[dt], b=b, t=t, h=H, k=k, dil=dil, last=int(bool(last)),
                            ld_dh=0 if d_h is None else _rows(d_h, "d_h"), ld_do=_rows(d_o, "d_o"), ldpre=_rows(pre, "pre"),
                            lddpre=_rows(d_pre, "d_pre"), ldout=_rows(d_h_out, "d_h_out"),
                            d_h=None if last else p(d_h), d_o=d_o.data_ptr(), pre=pre.data_ptr(), w_rs_t=packed.ptr(layer, 2),
                            w_in_t=packed.ptr(layer, 3), d_pre=d_pre.data_ptr(), d_h_out=d_h_out.data_ptr(), lengths=p(lengths))
    import ctypes
    e0 = _lib.timer.start("vits_wn_layer_bwd")
    rc = _lib.lib().vits_wn_layer_bwd(ctypes.addressof(d), _lib.stream_ptr())
    if e0 is not None:
        es = d_o.element_size()
        c_rs = H if last else 2 * H
        _lib.timer.stop("vits_wn_layer_bwd", e0, (2.0 * b * t * H * (2 * H * k + c_rs),
                                                   es * (b * t * H * (6 + (0 if last else 1)) + (2 * k + c_rs // H) * H * H)),
                        shape=f"b{b} t{t} H{H} k{k} d{dil} last{int(bool(last))} {str(dt)[6:]}")
    _lib.check(rc, "vits_wn_layer_bwd")
    return True


def colsum(x, per_item=False):
    """float32 column sums of a contiguous channels-last [b, t, c] tensor: per item ([b, c]) or over the batch ([c])."""
    _lib.require_cuda(x)
    assert x.is_contiguous() and x.dim() == 3
    b, t, c = x.shape
    n_seg, rows = (b, t) if per_item else (1, b * t)
    out = torch.empty((b, c) if per_item else (c,), device=x.device, dtype=torch.float32)
    L = _lib.lib()
    nbytes = L.vits_colsum_workspace(n_seg, rows, c)
    ws = workspace(nbytes, x.device)
    rc = L.vits_colsum(_DT[x.dtype], x.data_ptr(), n_seg, rows, c, out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "vits_colsum")
    return out


def lrelu_mask_bwd(dy, y=None, slope=1.0, lengths=None):
    """dy * (y > 0 ? 1 : slope) with rows >= lengths zeroed, one launch (vits_lrelu_mask_bwd); dy [b,t,c] contiguous."""
    _lib.require_cuda(dy)
    assert dy.is_contiguous() and dy.dim() == 3 and (y is None or (y.is_contiguous() and y.shape == dy.shape and y.dtype == dy.dtype))
    out = torch.empty_like(dy)
    b, t, c = dy.shape
    rc = _lib.lib().vits_lrelu_mask_bwd(_DT[dy.dtype], dy.data_ptr(), None if y is None else y.data_ptr(), float(slope),
                                        None if lengths is None else lengths.data_ptr(), b, t, c, out.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "vits_lrelu_mask_bwd")
    return out


_workspace = {}


def workspace(nbytes, device):
    """Grow-only device scratch for the kernels that need one: one buffer per (device, stream), reused in stream order
    (independent branches of the step run on their own streams and must not share slabs)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream) if torch.device(device).type == "cuda" else (device, 0)
    buf = _workspace.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspace[key] = buf
    return buf


_side_streams = {}
_open_branches = []          # SideBranch objects between __enter__ and __exit__ (innermost last)


class SideBranch:
    """Runs a block of launches on a second HIP stream, forked from the current one and joined later with join():

        br = SideBranch(device, inputs...)
        with br:
            out = small_latency_bound_subgraph(...)
        ... independent work on the main stream ...
        br.join(out)

    PyTorch runs the backward of every op on the stream its forward ran on, so the backward of the block overlaps the
    backward of the independent work as well; inside a captured hipGraph the two become parallel branches (one fork and one
    join edge each way, not one per kernel).  Used for the stochastic duration predictor: ~800 launches on [16, 201, 192]
    tensors that keep a few CUs busy, next to the decoder / discriminators."""

    def __init__(self, device, *inputs, lane=0):
        """lane: which of the device's side streams to use (branches on different lanes also overlap each other)."""
        device = torch.device(device)
        self.main = torch.cuda.current_stream(device)
        self.side = _side_streams.get((device.index, lane))
        if self.side is None:
            self.side = _side_streams[(device.index, lane)] = torch.cuda.Stream(device)
        self.inputs = [t for t in inputs if torch.is_tensor(t)]
        self._ctx = None

    def __enter__(self):
        # A branch opened inside another one (from its side stream, or twice on one lane) is refused: the nested fork/join
        # pattern is what crashed hipStreamEndCapture when the step was captured — an inner branch joins into
        # the OUTER side stream, so the capture's origin stream never sees that lane rejoin.  Open branches one after the
        # other from the main stream instead (they still overlap each other on different lanes).
        if _open_branches:
            raise RuntimeError("SideBranch: opened inside another open branch (nested branches are not supported)")
        if any(self.main.cuda_stream == s.cuda_stream for s in _side_streams.values()):
            raise RuntimeError("SideBranch: opened from a side stream (nested branches are not supported)")
        _open_branches.append(self)
        self.side.wait_stream(self.main)
        for t in self.inputs:
            t.record_stream(self.side)
        self._ctx = torch.cuda.stream(self.side)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self._ctx.__exit__(*exc)
        if _open_branches and _open_branches[-1] is self:
            _open_branches.pop()
        return False

    def join(self, *outputs):
        """The CURRENT stream (the one the branch was forked from, or another branch that consumes its results) waits for the
        branch.  Returns the outputs wrapped so that, in the backward pass, the gradient that re-enters the branch is first
        copied into memory that belongs to the side stream (see _HandOff)."""
        cur = torch.cuda.current_stream(self.side.device)
        cur.wait_stream(self.side)
        res = []
        for t in outputs:
            if torch.is_tensor(t):
                t.record_stream(cur)
                if t.requires_grad:
                    t = _HandOff.apply(t, self.side)
            res.append(t)
        return res[0] if len(res) == 1 else tuple(res)


class _HandOff(torch.autograd.Function):
    """Identity whose backward moves the incoming gradient (allocated on the main stream) into a block owned by the side
    stream before the branch's backward nodes read it: those nodes may still be running long after the main stream has freed
    and re-used the original block."""

    @staticmethod
    def forward(ctx, x, side):
        ctx.side = side
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        main, side = torch.cuda.current_stream(g.device), ctx.side
        side.wait_stream(main)
        with torch.cuda.stream(side):
            g2 = g.clone()
        g.record_stream(side)
        return g2, None


class DeferredReductions:
    """Collects the second stages of a group of weight-gradient launches (vits_conv1d_cl_wgrad_deferred) and runs them as ONE
    launch (vits_wgrad_reduce_pending).  Each deferred call gets its own slice of a per-stream slab buffer shared by all
    collectors (bump allocation; the space is recycled when no collector has anything pending).  The buffer starts small and
    grows to what one backward needs (grow-only, re-allocated only while nothing is pending, i.e. during the warm-up steps
    before a graph capture); when it is full mid-backward the collector flushes early or the call falls back to the immediate
    form.  A collector belongs to the stream of its first add(): flushing from another stream is an error (the slabs would
    be reduced without a dependency on the launches that wrote them).  Used by the fused layer nodes and by the weight
    arena, whose weight gradients are only consumed after their backward returns."""
    INITIAL = 32 << 20
    LIMIT = 1 << 30
    _state = {}                  # (device, stream) -> [buffer, bump offset, pending entries of all collectors, high-water mark, bytes since the last reset]

    def __init__(self, device):
        self.device, self.pending, self.stream = device, [], None
        self.batch = []                # weight-gradient launches handed in by add_wgrad(): run as one batched launch at the flush

    def _key(self):
        return (self.device, torch.cuda.current_stream(self.device).cuda_stream)

    def _st(self):
        key = self._key()
        if self.stream is not None and key[1] != self.stream:
            raise RuntimeError("DeferredReductions: used from a different stream than the one its pending launches ran on")
        st = DeferredReductions._state.get(key)
        if st is None:
            # a new stream (e.g. the capture stream after warm-up on a side stream) starts at the largest need seen so far on
            # this device: inside a capture the buffer cannot be replaced any more
            seen = max([v[3] for k, v in DeferredReductions._state.items() if k[0] == key[0]], default=0)
            size = max(self.INITIAL, min(self.LIMIT, (seen * 5 // 4 + 255) & ~255))
            st = DeferredReductions._state[key] = [torch.empty(size, dtype=torch.uint8, device=self.device), 0, 0, seen, 0]
        return st

    def alloc(self, nbytes):
        nbytes = (nbytes + 255) & ~255
        st = self._st()
        early = False
        # right-size an idle buffer before a backward starts filling it: the high-water mark of the previous backward (on any
        # stream of this device) says what one backward needs — without this a buffer could stay too small for good when every
        # overflow happened while something was pending (no growth possible then), and the captured step would take the
        # unsplit fall-backs while an eager step on a fresh stream (right-sized from the start) split: same math, other rounding
        if st[2] == 0 and st[1]:
            # nothing pending on this stream: whatever was handed out before has been reduced, or its launch turned out unsplit
            # and never used it (an allocation without add()) — such slabs used to stay allocated until the next flush WITH
            # pending entries, and on a stream where none came (direct calls in tests) the buffer filled up for good
            st[1], st[4] = 0, 0
        if st[1] == 0 and st[2] == 0 and not torch.cuda.is_current_stream_capturing():
            seen = max(v[3] for k, v in DeferredReductions._state.items() if k[0] == self.device)
            want = min(self.LIMIT, (seen * 5 // 4 + 255) & ~255)
            if st[0].numel() < want:
                st[0] = torch.empty(want, dtype=torch.uint8, device=self.device)
        if st[1] + nbytes > st[0].numel():
            run = st[4]
            self.flush()                                  # early flush: the buffer is too small for this backward
            st[4], early = run, True
        if st[1] + nbytes > st[0].numel() and st[2] == 0 and not torch.cuda.is_current_stream_capturing():
            want = max(2 * st[0].numel(), st[3] * 5 // 4, nbytes)
            if want <= self.LIMIT:
                st[0] = torch.empty(want, dtype=torch.uint8, device=self.device)      # nothing pending: safe to replace
                st[1] = 0
        st[4] += nbytes
        st[3] = max(st[3], st[4])                         # bytes one backward would need without early flushes
        if st[1] + nbytes > st[0].numel():
            if os.environ.get("VITS_DEBUG_DEFER"):
                print(f"[defer] no room: stream {self._key()[1]} want {nbytes >> 10} KiB at {st[1] >> 10} of {st[0].numel() >> 10} KiB, pending {st[2]}, "
                      f"high-water {st[3] >> 10} KiB, capturing {torch.cuda.is_current_stream_capturing()}", flush=True)
            return None                                   # caller falls back to the immediate form
        view = st[0][st[1]:st[1] + nbytes]
        st[1] += nbytes
        return view

    def add(self, pend):
        st = self._st()
        if self.stream is None:
            self.stream = self._key()[1]
        self.pending.append(pend)
        st[2] += 1

    def add_wgrad(self, entry):
        """Defers a whole weight-gradient LAUNCH (an entry of conv1d_cl_wgrad_batch: x, dy, k, out, ...; the tensors are kept
        alive here) to the flush, where all of them run as one batched launch — the single convolutions of the text encoder,
        the duration predictor and the flows' projections are ~100 launches of one or two workgroup rounds each otherwise.
        Returns False if the entry is not eligible for the batched kernel (the caller then launches it itself)."""
        if not wgrad_batch_eligible(entry):
            return False
        self._st()                                     # (stream check)
        if self.stream is None:
            self.stream = self._key()[1]
        self.batch.append(entry)
        return True

    def flush(self):
        if self.batch:
            batch, self.batch = self.batch, []
            if not conv1d_cl_wgrad_batch(batch, defer=self):
                raise RuntimeError("DeferredReductions: a deferred weight-gradient batch was refused")      # (eligibility was tested at add_wgrad)
        if not self.pending:
            if not self.batch:
                self.stream = None
            return
        import ctypes
        st = self._st()
        arr = (_lib.WgradPending * len(self.pending))(*self.pending)
        rc = _lib.lib().vits_wgrad_reduce_pending(ctypes.addressof(arr), len(self.pending), _lib.stream_ptr())
        _lib.check(rc, "vits_wgrad_reduce_pending")
        st[2] -= len(self.pending)
        self.pending, self.stream = [], None
        if st[2] <= 0:
            st[1], st[2], st[4] = 0, 0, 0


_counters = {}
# Measured: correct (bitwise equal to the two-launch result, tests/test_conv_gpu.py) but 3x SLOWER for the whole step: the
# agent-scope release/acquire fences the hand-off needs write back / invalidate a whole XCD's L2 on gfx950, once per
# workgroup.  Left as an opt-in experiment; the default sums the slabs in a second launch.
FUSED_WGRAD_REDUCE = False


def tile_counters(device):
    """Zero-initialised, self-re-arming per-tile counters of vits_conv1d_cl_wgrad's fused slab reduction: one buffer per
    (device, stream), like the scratch."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _counters.get(key)
    if buf is None:
        buf = _counters[key] = torch.zeros(1 << 16, dtype=torch.int32, device=device)
    return buf


def conv1d_cl_wgrad_raw(x, dy, k, lengths=None, dil=1, pad=0, stride=1, in_slope=1.0, flags=0, out=None, dbias=None, groups=1, defer=None):
    """dW [k, c_out, c_in] float32 of conv1d_cl_raw(x, w, ...) given dy [b, t_out, c_out]; optionally the bias
    gradient (column sums of dy) into `dbias` float32 [c_out] in the same launch."""
    _lib.require_cuda(x, dy)
    assert x.dtype == dy.dtype
    b, t, c_in = x.shape
    c_out = dy.shape[2]
    t_out = (t + 2 * pad - dil * (k - 1) - 1) // stride + 1
    assert tuple(dy.shape[:2]) == (b, t_out), (tuple(x.shape), tuple(dy.shape))
    c_in_w = c_in // groups                   # grouped: compact dw [k][c_out][c_in / groups]
    if out is None:
        assert not (flags & CONV_ACCUM)
        out = torch.empty((k, c_out, c_in_w), device=x.device, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (k, c_out, c_in_w)
    L = _lib.lib()
    ws_bytes = L.vits_conv1d_cl_wgrad_workspace(b, t_out, c_in_w, c_out, k)
    ws = defer.alloc(ws_bytes) if defer is not None else None          # deferred second stage: a slice of its own
    if ws is None:
        defer, ws = None, workspace(ws_bytes, x.device)
    d = _lib.WgradDesc(dtype=_DT[x.dtype], b=b, t=t, c_in=c_in, c_out=c_out, k=k, dil=dil, pad=pad, stride=stride, flags=int(flags),
                       ldx=_rows(x, "x"), lddy=_rows(dy, "dy"), in_slope=float(in_slope), groups=groups,
                       x=x.data_ptr(), dy=dy.data_ptr(), dw=out.data_ptr(), workspace=ws.data_ptr(), workspace_bytes=ws.numel(),
                       lengths=None if lengths is None else lengths.data_ptr(),
                       dbias=None if dbias is None else dbias.data_ptr())
    if FUSED_WGRAD_REDUCE:
        cnt = tile_counters(x.device)
        d.counters, d.counters_len = cnt.data_ptr(), cnt.numel()
    assert dbias is None or (dbias.dtype == torch.float32 and dbias.is_contiguous() and dbias.numel() == c_out)
    import ctypes
    e0 = _lib.timer.start("vits_conv1d_cl_wgrad")
    if defer is not None:
        pend = _lib.WgradPending()
        rc = L.vits_conv1d_cl_wgrad_deferred(ctypes.addressof(d), _lib.stream_ptr(), ctypes.addressof(pend))
        if rc == 0 and pend.splits > 0:
            defer.add(pend)
    else:
        rc = L.vits_conv1d_cl_wgrad(ctypes.addressof(d), _lib.stream_ptr())
    if e0 is not None:
        es = x.element_size()
        _lib.timer.stop("vits_conv1d_cl_wgrad", e0, (2.0 * b * t_out * c_out * c_in_w * k,
                                                      es * (b * t * c_in + b * t_out * c_out) + 4.0 * k * c_out * c_in_w),
                        shape=f"b{b} t{t} ci{c_in} co{c_out} k{k} d{dil} s{stride} {str(x.dtype)[6:]}")
    _lib.check(rc, "vits_conv1d_cl_wgrad")
    return out


def _wgrad_batch_descs(entries):
    descs = (_lib.WgradDesc * len(entries))()
    for d, e in zip(descs, entries):
        x, dy, out = e["x"], e["dy"], e["out"]
        _lib.require_cuda(x, dy, out)
        b, t, c_in = x.shape
        c_out, k = dy.shape[2], e["k"]
        assert x.dtype == dy.dtype and tuple(dy.shape[:2]) == (b, t)
        assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (k, c_out, c_in)
        db, lengths = e.get("dbias"), e.get("lengths")
        assert db is None or (db.dtype == torch.float32 and db.is_contiguous() and db.numel() == c_out)
        d.dtype, d.b, d.t, d.c_in, d.c_out, d.k = _DT[x.dtype], b, t, c_in, c_out, k
        d.dil, d.pad, d.stride, d.flags = e.get("dil", 1), e.get("pad", 0), 1, int(e.get("flags", 0))
        d.ldx, d.lddy, d.in_slope, d.groups = _rows(x, "x"), _rows(dy, "dy"), float(e.get("in_slope", 1.0)), 1
        d.x, d.dy, d.dw = x.data_ptr(), dy.data_ptr(), out.data_ptr()
        d.lengths = None if lengths is None else lengths.data_ptr()
        d.dbias = None if db is None else db.data_ptr()
    return descs


def wgrad_batch_eligible(entry):
    """Whether conv1d_cl_wgrad_batch takes this entry (host-side test, nothing is launched)."""
    import ctypes
    x, dy = entry["x"], entry["dy"]
    if x.dim() != 3 or dy.dim() != 3 or x.stride(2) != 1 or dy.stride(2) != 1 or tuple(dy.shape[:2]) != tuple(x.shape[:2]) or x.dtype != dy.dtype:
        return False
    d, s = _wgrad_batch_descs([entry]), (ctypes.c_int * 1)()
    return _lib.lib().vits_conv1d_cl_wgrad_batch_plan(ctypes.addressof(d), 1, ctypes.addressof(s)) == 0


def conv1d_cl_wgrad_batch(entries, defer=None):
    """The weight (+ bias) gradients of a group of stride-1 "same" convolutions in one launch per taps-per-group class
    (vits_conv1d_cl_wgrad_batch, csrc/conv1d_wgrad_batch.hip).  entries: dicts with x [b,t,c_in], dy [b,t,c_out], k, out
    (fp32 [k,c_out,c_in], written), and optionally dbias (fp32 [c_out]), lengths, dil, pad, flags, in_slope.  With enough tiles in the
    group nothing but `out` is written; a small group splits the reduction into slabs, which need `defer` (a
    DeferredReductions collector: the caller flushes it).  Returns False when an entry is not eligible (nothing was launched)."""
    import ctypes
    L = _lib.lib()
    n = len(entries)
    descs = _wgrad_batch_descs(entries)
    # slabs only for entries whose reduction the launcher will split (long reductions over few tiles): ONE allocation for the
    # whole batch, carved up here (several allocations before the launch could be recycled by an early flush in between)
    pend = None
    if defer is not None:
        splits = (ctypes.c_int * n)()
        rc = L.vits_conv1d_cl_wgrad_batch_plan(ctypes.addressof(descs), n, ctypes.addressof(splits))
        if rc == _lib.E_UNSUPPORTED:
            return False
        _lib.check(rc, "vits_conv1d_cl_wgrad_batch_plan")
        need = [(((S * (d.k * d.c_out * d.c_in + (d.c_out if d.dbias else 0)) * 4) + 255) & ~255) if S > 1 else 0 for d, S in zip(descs, splits)]
        ws = defer.alloc(sum(need)) if sum(need) else None
        if ws is not None:
            off = 0
            for d, nb in zip(descs, need):
                if nb:
                    d.workspace, d.workspace_bytes = ws.data_ptr() + off, nb
                    off += nb
            pend = (_lib.WgradPending * n)()
    e0 = _lib.timer.start("vits_conv1d_cl_wgrad")
    rc = L.vits_conv1d_cl_wgrad_batch(ctypes.addressof(descs), n, _lib.stream_ptr(), None if pend is None else ctypes.addressof(pend))
    if rc == _lib.E_UNSUPPORTED:
        return False
    if e0 is not None:
        es = entries[0]["x"].element_size()
        fl = sum(2.0 * d.b * d.t * d.c_out * d.c_in * d.k for d in descs)
        by = sum(es * d.b * d.t * (d.c_in + d.c_out) + 4.0 * d.k * d.c_out * d.c_in for d in descs)
        _lib.timer.stop("vits_conv1d_cl_wgrad", e0, (fl, by), shape=f"batch of {n}: first b{descs[0].b} t{descs[0].t} ci{descs[0].c_in} co{descs[0].c_out} k{descs[0].k}")
    _lib.check(rc, "vits_conv1d_cl_wgrad_batch")
    if pend is not None:
        for i in range(n):
            if pend[i].splits > 0:
                defer.add(pend[i])
    return True
