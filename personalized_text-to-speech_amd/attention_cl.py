"""Windowed relative-position self-attention (reference attentions.MultiHeadAttention.attention,
attentions.py:150-182) on the HIP kernels, forward and backward, one autograd node.

Every contraction is a launch of the MFMA convolution kernel with k = 1: shared operands for the
relative embeddings (Q E_k^T, P_band E_v and their gradients), per-item operands (vits_conv_desc
w_batch_stride / ldw) for Q K^T, P V and their gradients; the softmax with its relative-logit skew, mask,
dropout and band extraction is one row kernel each way (csrc/attn_softmax.hip).  Heads are channel
slices of the channels-last tensors (row pitch = channels), never copied.
"""
import torch
import torch.nn.functional as F

from . import _lib
from . import kernels as K

BAND = 16


def relsoftmax(s, r, keep, lengths, window, scale, want_pd):
    b, t, ld = s.shape
    p = torch.empty_like(s)
    pd = torch.empty_like(s) if want_pd else None
    pband = torch.empty(b, t, BAND, device=s.device, dtype=s.dtype)
    ptr = lambda x: None if x is None else x.data_ptr()
    rc = _lib.lib().vits_relsoftmax(K._DT[s.dtype], s.data_ptr(), ptr(r), ptr(keep), ptr(lengths), p.data_ptr(), ptr(pd),
                                    pband.data_ptr(), b, t, ld, window, float(scale), _lib.stream_ptr())
    _lib.check(rc, "vits_relsoftmax")
    return p, (pd if want_pd else p), pband


def relsoftmax_bwd(p, dpd, dpband, keep, lengths, window, scale):
    b, t, ld = p.shape
    ds = torch.empty_like(p)
    dsband = torch.empty(b, t, BAND, device=p.device, dtype=p.dtype)
    ptr = lambda x: None if x is None else x.data_ptr()
    rc = _lib.lib().vits_relsoftmax_bwd(K._DT[p.dtype], p.data_ptr(), dpd.data_ptr(), ptr(dpband), ptr(keep), ptr(lengths),
                                        ds.data_ptr(), dsband.data_ptr(), b, t, ld, window, float(scale), _lib.stream_ptr())
    _lib.check(rc, "vits_relsoftmax_bwd")
    return ds, dsband


def _t_pad(x, t8):
    """[b, t, c] -> [b, c, t8]: transposed, time zero-padded to the vector width."""
    return F.pad(x.transpose(1, 2), (0, t8 - x.size(1))).contiguous()


def _rel_operand(emb, dtype):
    """[1, 2w+1, dk] parameter -> [16][dk] (rows >= 2w+1 zero) in the compute dtype."""
    e = emb.detach()[0].to(dtype)
    return F.pad(e, (0, 0, 0, BAND - e.size(0))).contiguous()


class AttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dtype, q, k, v, emb_k, emb_v, lengths, n_heads, window, p_drop, training):
        C_ = K.conv1d_cl_raw
        q, k, v = (x.detach().to(dtype).contiguous() for x in (q, k, v))
        b, t, c = q.shape
        dk = c // n_heads
        t8 = (t + 7) // 8 * 8
        scale = dk ** -0.5
        ek, ev = _rel_operand(emb_k, dtype), _rel_operand(emb_v, dtype)
        ev_t = ev.t().contiguous()                                           # [dk][16]
        v_t = _t_pad(v, t8)                                                  # [b, c, t8]
        out = torch.empty_like(q)
        saved, probs = [], []
        for h in range(n_heads):
            sl = slice(h * dk, (h + 1) * dk)
            qh, kh = q[..., sl], k[..., sl]
            s = torch.empty(b, t, t8, device=q.device, dtype=dtype)
            C_(qh, kh.unsqueeze(1), out=s[..., :t])                           # raw scores Q K^T
            r = C_(qh, ek.unsqueeze(0))                                       # raw relative-key logits [b, t, 16]
            keep = None
            if training and p_drop > 0:
                keep = ((torch.rand(b, t, t8, device=q.device) >= p_drop).to(dtype) / (1.0 - p_drop))
            p, pd, pband = relsoftmax(s, r, keep, lengths, window, scale, keep is not None)
            oh = out[..., sl]
            C_(pd, v_t[:, sl, :].unsqueeze(1), out=oh)                        # P V
            C_(pband, ev_t.unsqueeze(0), res=oh, out=oh)                      # + P_band E_v
            saved += [p, pd, pband] + ([keep] if keep is not None else [])
            probs.append(pd[..., :t])
        ctx.cfg = (dtype, n_heads, window, scale, dk, t8, training and p_drop > 0, q.dtype)
        ctx.lengths = lengths
        ctx.save_for_backward(q, k, v, ek, ev, *saved)
        ctx.mark_non_differentiable(*[])
        return out, torch.stack(probs, 1)                                     # [b, t, c], [b, h, t, t]

    @staticmethod
    def backward(ctx, d_out, _d_probs):
        C_, WG = K.conv1d_cl_raw, K.conv1d_cl_wgrad_raw
        dtype, n_heads, window, scale, dk, t8, dropped, _ = ctx.cfg
        q, k, v, ek, ev, *saved = ctx.saved_tensors
        lengths = ctx.lengths
        b, t, c = q.shape
        d_out = d_out.to(dtype).contiguous()
        ek_t = ek.t().contiguous()                                            # [dk][16]
        q_t, k_t, do_t = _t_pad(q, t8), _t_pad(k, t8), _t_pad(d_out, t8)      # [b, c, t8]
        dq, dk_, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        d_ek = torch.empty(1, BAND, dk, dtype=torch.float32, device=q.device)
        d_ev = torch.empty(1, BAND, dk, dtype=torch.float32, device=q.device)
        per = 4 if dropped else 3
        for h in range(n_heads):
            sl = slice(h * dk, (h + 1) * dk)
            p, pd, pband = saved[per * h: per * h + 3]
            keep = saved[per * h + 3] if dropped else None
            doh = d_out[..., sl]
            dpband = C_(doh, ev.unsqueeze(0))                                 # dO E_v^T           [b, t, 16]
            dpd = torch.empty(b, t, t8, device=q.device, dtype=dtype)
            C_(doh, v[..., sl].unsqueeze(1), out=dpd[..., :t])                # dO V^T             [b, t, t]
            ds, dsband = relsoftmax_bwd(p, dpd, dpband, keep, lengths, window, scale)
            C_(_t_pad(pd[..., :t], t8), do_t[:, sl, :].unsqueeze(1), out=dv[..., sl])      # dV = P^T dO
            dqh = dq[..., sl]
            C_(ds, k_t[:, sl, :].unsqueeze(1), out=dqh)                       # dQ = dS K
            C_(dsband, ek_t.unsqueeze(0), res=dqh, out=dqh)                   #    + dS_band E_k
            C_(_t_pad(ds[..., :t], t8), q_t[:, sl, :].unsqueeze(1), out=dk_[..., sl])      # dK = dS^T Q
            acc = K.CONV_ACCUM if h > 0 else 0
            WG(q[..., sl], dsband, 1, flags=acc, out=d_ek)                    # dE_k = dS_band^T Q
            WG(doh, pband, 1, flags=acc, out=d_ev)                            # dE_v = P_band^T dO
        nrel = 2 * window + 1
        return None, dq, dk_, dv, d_ek[:, :nrel], d_ev[:, :nrel], None, None, None, None, None


def rel_attention_cl(q, k, v, emb_k, emb_v, lengths, n_heads, window, p_drop, training, dtype):
    """q, k, v [b, t, c] channels-last -> (out [b, t, c], p_attn [b, h, t, t])."""
    return AttnFn.apply(dtype, q, k, v, emb_k, emb_v, lengths, n_heads, window, p_drop, training)
