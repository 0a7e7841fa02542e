"""Autograd wrappers of the row-wise HIP kernels (csrc/rowops.hip, csrc/rq_spline.hip):
LayerNorm(+GELU)(+residual) over channels, depth-wise dilated convolution, rational-quadratic spline.
All tensors channels-last; parameters stay float32."""
import torch

from . import _lib
from . import kernels as K


def _ws(rows, c, k, device):
    n = _lib.lib().vits_rowops_workspace(rows, c, k)
    return K.workspace(n, device)


class LnActFn(torch.autograd.Function):
    """y = [res +] act(LayerNorm_c(x) * gamma + beta); x [..., c] contiguous; act 0 = identity, 1 = GELU."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, eps, act):
        _lib.require_cuda(x)
        xd = x.detach().contiguous()
        c = xd.size(-1)
        rows = xd.numel() // c
        y = torch.empty_like(xd)
        rd = None if res is None else res.detach().to(xd.dtype).contiguous()
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        rc = _lib.lib().vits_ln_act_cl(K._DT[xd.dtype], xd.data_ptr(), g32.data_ptr(), b32.data_ptr(),
                                       None if rd is None else rd.data_ptr(), y.data_ptr(), rows, c, float(eps), int(act), _lib.stream_ptr())
        _lib.check(rc, "vits_ln_act_cl")
        ctx.save_for_backward(xd, g32, b32)
        ctx.cfg = (rows, c, float(eps), int(act), res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xd, g32, b32 = ctx.saved_tensors
        rows, c, eps, act, has_res = ctx.cfg
        dy = dy.to(xd.dtype).contiguous()
        dx = torch.empty_like(xd)
        dg = torch.empty(c, dtype=torch.float32, device=xd.device)
        db = torch.empty_like(dg)
        ws = _ws(rows, c, 1, xd.device)
        rc = _lib.lib().vits_ln_act_cl_bwd(K._DT[xd.dtype], xd.data_ptr(), g32.data_ptr(), b32.data_ptr(), dy.data_ptr(), dx.data_ptr(),
                                           dg.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), rows, c, eps, act, 0, _lib.stream_ptr())
        _lib.check(rc, "vits_ln_act_cl_bwd")
        return dx, dg, db, (dy if has_res else None), None, None


def ln_act(x, gamma, beta, res=None, eps=1e-5, act=0):
    return LnActFn.apply(x, gamma, beta, res, eps, act)


class DwConvFn(torch.autograd.Function):
    """Depth-wise conv over time of x [b,t,c] (rows >= lengths[b] read as zero), weight [c,1,k] (torch layout)."""

    @staticmethod
    def forward(ctx, x, weight, bias, lengths, dil):
        _lib.require_cuda(x)
        xd = x.detach().contiguous()
        b, t, c = xd.shape
        k = weight.size(-1)
        w32 = weight.detach().float().reshape(c, k).contiguous()
        b32 = None if bias is None else bias.detach().float().contiguous()
        y = torch.empty_like(xd)
        rc = _lib.lib().vits_dwconv_cl(K._DT[xd.dtype], xd.data_ptr(), w32.data_ptr(), None if b32 is None else b32.data_ptr(),
                                       None if lengths is None else lengths.data_ptr(), y.data_ptr(), b, t, c, k, int(dil), _lib.stream_ptr())
        _lib.check(rc, "vits_dwconv_cl")
        ctx.save_for_backward(xd, w32)
        ctx.lengths, ctx.cfg = lengths, (b, t, c, k, int(dil), bias is not None, tuple(weight.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        xd, w32 = ctx.saved_tensors
        b, t, c, k, dil, has_bias, wshape = ctx.cfg
        dy = dy.to(xd.dtype).contiguous()
        dx = torch.empty_like(xd)
        dw = torch.empty(c, k, dtype=torch.float32, device=xd.device)
        db = torch.empty(c, dtype=torch.float32, device=xd.device)
        ws = _ws(b * t, c, k, xd.device)
        rc = _lib.lib().vits_dwconv_cl_bwd(K._DT[xd.dtype], xd.data_ptr(), w32.data_ptr(), None if ctx.lengths is None else ctx.lengths.data_ptr(),
                                           dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(),
                                           b, t, c, k, dil, 0, _lib.stream_ptr())
        _lib.check(rc, "vits_dwconv_cl_bwd")
        return dx, dw.view(wshape), (db if has_bias else None), None, None


def dwconv(x, weight, bias, lengths, dil):
    return DwConvFn.apply(x, weight, bias, lengths, dil)


class SplineFn(torch.autograd.Function):
    """(y, logabsdet) = rq_spline(x [n] fp32, h [n, ldh >= 29] fp32|bf16)."""

    @staticmethod
    def forward(ctx, x, h, hscale, inverse, tail_bound):
        _lib.require_cuda(x, h)
        xd = x.detach().float().contiguous()
        hd = h.detach().contiguous()
        n, ldh = hd.shape
        assert xd.numel() == n
        y = torch.empty_like(xd)
        lad = torch.empty_like(xd)
        rc = _lib.lib().vits_rq_spline(K._DT[hd.dtype], xd.data_ptr(), hd.data_ptr(), ldh, float(hscale), int(bool(inverse)), float(tail_bound),
                                       y.data_ptr(), lad.data_ptr(), n, _lib.stream_ptr())
        _lib.check(rc, "vits_rq_spline")
        ctx.save_for_backward(xd, hd)
        ctx.cfg = (float(hscale), int(bool(inverse)), float(tail_bound), x.dtype)
        return y, lad

    @staticmethod
    def backward(ctx, gy, gl):
        xd, hd = ctx.saved_tensors
        hscale, inverse, tb, xdtype = ctx.cfg
        n, ldh = hd.shape
        gy = gy.float().contiguous()
        gl = gl.float().contiguous()
        gx = torch.empty_like(xd)
        gh = torch.empty_like(hd)
        rc = _lib.lib().vits_rq_spline_bwd(K._DT[hd.dtype], xd.data_ptr(), hd.data_ptr(), ldh, hscale, inverse, tb,
                                           gy.data_ptr(), gl.data_ptr(), gx.data_ptr(), gh.data_ptr(), n, _lib.stream_ptr())
        _lib.check(rc, "vits_rq_spline_bwd")
        return gx.to(xdtype), gh, None, None, None


def rq_spline(x, h, hscale, inverse, tail_bound):
    return SplineFn.apply(x, h, hscale, inverse, tail_bound)


class FlowFrontFn(torch.autograd.Function):
    """h = x[..., c0, None] * w + bias (+ g): the Conv1d(1, C, 1) in front of a flow's DDSConv stack with that stack's `x + g`
    folded in (csrc/flow_edge.hip).  x [b, t, xs] float32 (xs = 1 | 2), w [C, 1, 1] / bias [C] parameters, g [b, t, C] or None."""

    @staticmethod
    def forward(ctx, x, c0, w, bias, g, dtype):
        _lib.require_cuda(x, w)
        xd = x.detach().float().contiguous()
        b, t, xs = xd.shape
        C = w.numel()
        wd, bd = w.detach().float().reshape(C).contiguous(), (None if bias is None else bias.detach().float().contiguous())
        gd = None if g is None else g.detach().to(dtype).contiguous()
        h = torch.empty(b, t, C, dtype=dtype, device=xd.device)
        rc = _lib.lib().vits_flow_front(K._DT[dtype], xd.data_ptr(), xs, c0, wd.data_ptr(), None if bd is None else bd.data_ptr(),
                                        None if gd is None else gd.data_ptr(), h.data_ptr(), b * t, C, _lib.stream_ptr())
        _lib.check(rc, "vits_flow_front")
        ctx.save_for_backward(xd, wd)
        ctx.cfg = (c0, w.shape, bias is not None, None if g is None else g.dtype, x.dtype)
        return h

    @staticmethod
    def backward(ctx, dh):
        xd, wd = ctx.saved_tensors
        c0, wshape, has_bias, g_dtype, x_dtype = ctx.cfg
        b, t, xs = xd.shape
        C = wd.numel()
        dh = dh.contiguous()
        dx = torch.empty_like(xd)
        dw = torch.empty(C, dtype=torch.float32, device=xd.device)
        db = torch.empty(C, dtype=torch.float32, device=xd.device)
        L = _lib.lib()
        nbytes = L.vits_flow_front_workspace(b * t, C)
        ws = K.workspace(nbytes, xd.device)
        rc = L.vits_flow_front_bwd(K._DT[dh.dtype], xd.data_ptr(), xs, c0, wd.data_ptr(), dh.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                   db.data_ptr(), ws.data_ptr(), nbytes, b * t, C, _lib.stream_ptr())
        _lib.check(rc, "vits_flow_front_bwd")
        dg = None
        if g_dtype is not None and ctx.needs_input_grad[4]:
            dg = dh if dh.dtype == g_dtype else dh.to(g_dtype)
        return dx.to(x_dtype), None, dw.view(wshape), (db if has_bias else None), dg, None


def flow_front(x, c0, w, bias, g, dtype):
    return FlowFrontFn.apply(x, int(c0), w, bias, g, dtype)


class FlowTailFn(torch.autograd.Function):
    """(out [b, t, 2], logdet [b]) of one ConvFlow layer given its spline parameters: channel c1 of x is transformed, the other
    passes through, both times the mask; logdet = sum_t logabsdet * mask (modules.py:373-390) — csrc/rq_spline.hip FLOW mode."""

    @staticmethod
    def forward(ctx, x, h, mask, hscale, inverse, tail_bound, c1):
        _lib.require_cuda(x, h, mask)
        xd = x.detach().float().contiguous()
        b, t, two = xd.shape
        assert two == 2
        hd = h.detach().reshape(b * t, -1).contiguous()
        md = mask.detach().float().reshape(b * t).contiguous()
        out = torch.empty_like(xd)
        ladm = torch.empty(b, t, 1, dtype=torch.float32, device=xd.device)
        L = _lib.lib()
        rc = L.vits_flow_spline(K._DT[hd.dtype], xd.data_ptr(), hd.data_ptr(), hd.size(1), float(hscale), int(bool(inverse)), float(tail_bound),
                                md.data_ptr(), int(c1), out.data_ptr(), ladm.data_ptr(), b * t, _lib.stream_ptr())
        _lib.check(rc, "vits_flow_spline")
        from . import reduce
        logdet = reduce.sum12(ladm)                                   # [b] (two launches, fixed order)
        ctx.save_for_backward(xd, hd, md)
        ctx.cfg = (float(hscale), int(bool(inverse)), float(tail_bound), int(c1), x.dtype, h.shape)
        return out.to(x.dtype), logdet

    @staticmethod
    def backward(ctx, dout, dlogdet):
        xd, hd, md = ctx.saved_tensors
        hscale, inverse, tb, c1, x_dtype, hshape = ctx.cfg
        b, t, _ = xd.shape
        dout = torch.zeros_like(xd) if dout is None else dout.float().contiguous()
        dl = torch.zeros(b, dtype=torch.float32, device=xd.device) if dlogdet is None else dlogdet.float().contiguous()
        dx = torch.empty_like(xd)
        gh = torch.empty_like(hd)
        rc = _lib.lib().vits_flow_spline_bwd(K._DT[hd.dtype], xd.data_ptr(), hd.data_ptr(), hd.size(1), hscale, inverse, tb, md.data_ptr(), c1,
                                             dout.data_ptr(), dl.data_ptr(), t, dx.data_ptr(), gh.data_ptr(), b * t, _lib.stream_ptr())
        _lib.check(rc, "vits_flow_spline_bwd")
        return dx.to(x_dtype), gh.view(hshape), None, None, None, None, None


def flow_tail(x, h, mask, hscale, inverse, tail_bound, c1):
    return FlowTailFn.apply(x, h, mask, hscale, inverse, tail_bound, c1)


class CouplingTailFn(torch.autograd.Function):
    """flip([x0, stats + x1 * mask]) of a mean-only coupling layer and the Flip that follows it, one launch each way
    (csrc/flow_edge.hip).  x [b, t, C], stats [b, t, C - half] (same dtype), lengths int32 [b]."""

    @staticmethod
    def forward(ctx, x, stats, lengths, half, flip):
        _lib.require_cuda(x, stats)
        xd, sd = x.detach().contiguous(), stats.detach().to(x.dtype).contiguous()
        b, t, C = xd.shape
        y = torch.empty_like(xd)
        rc = _lib.lib().vits_coupling_tail(K._DT[xd.dtype], xd.data_ptr(), sd.data_ptr(), None if lengths is None else lengths.data_ptr(),
                                           y.data_ptr(), b, t, C, int(half), int(bool(flip)), _lib.stream_ptr())
        _lib.check(rc, "vits_coupling_tail")
        ctx.lengths, ctx.cfg = lengths, (int(half), int(bool(flip)), stats.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        half, flip, sdtype = ctx.cfg
        dy = dy.contiguous()
        b, t, C = dy.shape
        dx = torch.empty_like(dy)
        ds = torch.empty(b, t, C - half, dtype=dy.dtype, device=dy.device)
        rc = _lib.lib().vits_coupling_tail_bwd(K._DT[dy.dtype], dy.data_ptr(), None if ctx.lengths is None else ctx.lengths.data_ptr(),
                                               dx.data_ptr(), ds.data_ptr(), b, t, C, half, flip, _lib.stream_ptr())
        _lib.check(rc, "vits_coupling_tail_bwd")
        return dx, (ds if ds.dtype == sdtype else ds.to(sdtype)), None, None, None


def coupling_tail(x, stats, lengths, half, flip):
    return CouplingTailFn.apply(x, stats, lengths, half, flip)


class FlowAffineFn(torch.autograd.Function):
    """modules.ElementwiseAffine on a channels-last float32 state [b, t, C] (csrc/flow_edge.hip): forward -> (y, logdet [b]),
    inverse -> (y, None).  m / logs: the module's [C, 1] parameters; swap: indexed as if the channels were flipped."""

    @staticmethod
    def forward(ctx, x, m, logs, lengths, swap, inverse):
        _lib.require_cuda(x, m, logs, lengths)
        xd = x.detach().float().contiguous()
        b, t, C = xd.shape
        md, ld = m.detach().float().reshape(C).contiguous(), logs.detach().float().reshape(C).contiguous()
        y = torch.empty_like(xd)
        logdet = None if inverse else torch.empty(b, dtype=torch.float32, device=xd.device)
        rc = _lib.lib().vits_flow_affine(xd.data_ptr(), md.data_ptr(), ld.data_ptr(), lengths.data_ptr(), y.data_ptr(),
                                         None if logdet is None else logdet.data_ptr(), b, t, C, int(bool(swap)), int(bool(inverse)), _lib.stream_ptr())
        _lib.check(rc, "vits_flow_affine")
        ctx.save_for_backward(xd, ld, lengths)
        ctx.cfg = (int(bool(swap)), bool(inverse), m.shape, x.dtype)
        if inverse:
            ctx.mark_non_differentiable(y)
            return y.to(x.dtype), None
        return y.to(x.dtype), logdet

    @staticmethod
    def backward(ctx, dy, dlogdet):
        xd, ld, lengths = ctx.saved_tensors
        swap, inverse, pshape, x_dtype = ctx.cfg
        assert not inverse, "the inverse direction is inference-only"
        b, t, C = xd.shape
        dyc = None if dy is None else dy.float().contiguous()
        dlc = None if dlogdet is None else dlogdet.float().contiguous()
        dx = torch.empty_like(xd)
        dm = torch.empty(C, dtype=torch.float32, device=xd.device)
        dlogs = torch.empty(C, dtype=torch.float32, device=xd.device)
        rc = _lib.lib().vits_flow_affine_bwd(xd.data_ptr(), ld.data_ptr(), lengths.data_ptr(), None if dyc is None else dyc.data_ptr(),
                                             None if dlc is None else dlc.data_ptr(), dx.data_ptr(), dm.data_ptr(), dlogs.data_ptr(), b, t, C, swap,
                                             _lib.stream_ptr())
        _lib.check(rc, "vits_flow_affine_bwd")
        return dx.to(x_dtype), dm.view(pshape), dlogs.view(pshape), None, None, None


def flow_affine(x, m, logs, lengths, swap=False, inverse=False):
    return FlowAffineFn.apply(x, m, logs, lengths, swap, inverse)


class DequantLogFn(torch.autograd.Function):
    """The variational-dequantisation + modules.Log step of StochasticDurationPredictor.forward (reference models.py:71-80):
    z_q [b, t, 2] = [z_u, z1], w [b, t, 1] -> (z [b, t, 2] = [log(clamp((w - sigmoid(z_u) m) m)) m, z1],
    s1 [b] = sum (logsigmoid(z_u) + logsigmoid(-z_u)) m, s2 [b] = sum -z[..., 0])."""

    @staticmethod
    def forward(ctx, zq, w, lengths):
        _lib.require_cuda(zq, w, lengths)
        zd, wd = zq.detach().float().contiguous(), w.detach().float().contiguous()
        b, t, _ = zd.shape
        out = torch.empty_like(zd)
        s1 = torch.empty(b, dtype=torch.float32, device=zd.device)
        s2 = torch.empty(b, dtype=torch.float32, device=zd.device)
        rc = _lib.lib().vits_flow_dequant_log(zd.data_ptr(), wd.data_ptr(), lengths.data_ptr(), out.data_ptr(), s1.data_ptr(), s2.data_ptr(), b, t,
                                              _lib.stream_ptr())
        _lib.check(rc, "vits_flow_dequant_log")
        ctx.save_for_backward(zd, wd, lengths)
        ctx.zdtype = zq.dtype
        return out, s1, s2

    @staticmethod
    def backward(ctx, dout, ds1, ds2):
        zd, wd, lengths = ctx.saved_tensors
        b, t, _ = zd.shape
        c = lambda g: None if g is None else g.float().contiguous()
        dout, ds1, ds2 = c(dout), c(ds1), c(ds2)
        dz = torch.empty_like(zd)
        p = lambda g: None if g is None else g.data_ptr()
        rc = _lib.lib().vits_flow_dequant_log_bwd(zd.data_ptr(), wd.data_ptr(), lengths.data_ptr(), p(dout), p(ds1), p(ds2), dz.data_ptr(), b, t,
                                                  _lib.stream_ptr())
        _lib.check(rc, "vits_flow_dequant_log_bwd")
        return dz.to(ctx.zdtype), None, None


def dequant_log(zq, w, lengths):
    return DequantLogFn.apply(zq, w, lengths)
