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
