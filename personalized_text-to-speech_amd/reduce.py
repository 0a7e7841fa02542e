"""Loss-side reductions on this library's deterministic two-stage kernels (csrc/reduce.hip) instead of torch.sum /
torch.mean: those zero a semaphore with a device memset before every multi-block launch, and memset nodes are what a
replayed hipGraph cannot rely on.  Also fuses the feature-matching loss (reference losses.py:7-15).

  sum12(x)            torch.sum(x, [1, 2]) for a float32 [b, ., .] tensor            (models.py:71-102, modules.py)
  sum_all(x)          torch.sum(x) for a float32 tensor                                 (kl_loss, losses.py:46-61)
  feature_l1(hs)      sum_l 2 * mean |real_l - generated_l| over channels-last feature maps whose first half of the
                      batch is real and second half generated; gradient only to the generated half (losses.py:11)
"""
import torch

from . import _lib
from . import kernels as K

_DT = {torch.float32: 0, torch.bfloat16: 2}


def _ws(n_seg, device):
    L = _lib.lib()
    nbytes = L.vits_reduce_workspace(n_seg)
    return K.workspace(nbytes, device), nbytes


class _SegSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_seg):
        xd = x.detach()
        if xd.dtype != torch.float32 or not xd.is_contiguous():
            xd = xd.float().contiguous()
        _lib.require_cuda(xd)
        out = torch.empty(n_seg, device=x.device, dtype=torch.float32)
        ws, nbytes = _ws(n_seg, x.device)
        rc = _lib.lib().vits_segsum_f32(xd.data_ptr(), n_seg, xd.numel() // n_seg, out.data_ptr(), ws.data_ptr(), nbytes, _lib.stream_ptr())
        _lib.check(rc, "vits_segsum_f32")
        ctx.shape, ctx.dtype, ctx.n_seg = x.shape, x.dtype, n_seg
        return out

    @staticmethod
    def backward(ctx, g):
        shape = ctx.shape
        gx = g.to(ctx.dtype).view([ctx.n_seg] + [1] * (len(shape) - 1)).expand(shape) if ctx.n_seg > 1 else g.to(ctx.dtype).expand(shape)
        return gx, None


def sum12(x):
    """torch.sum(x, [1, 2]) -> [b]"""
    assert x.dim() == 3
    return _SegSum.apply(x, x.size(0))


def sum_all(x):
    """torch.sum(x) -> 0-d"""
    return _SegSum.apply(x, 1).view(())


class _FeatureL1(torch.autograd.Function):
    """All feature maps in one pass (vits_feature_l1[_bwd]): two launches forward, one backward."""

    @staticmethod
    def _items(hs, dens, dhs=None):
        items = (_lib.FeatItem * len(hs))()
        for i, h in enumerate(hs):
            items[i].h, items[i].dh = h.data_ptr(), (None if dhs is None else dhs[i].data_ptr())
            items[i].n, items[i].scale = h.numel() // 2, 2.0 / dens[i]
        return items

    @staticmethod
    def forward(ctx, dens, *hs):
        import ctypes
        L = _lib.lib()
        hd = [h.detach() for h in hs]
        assert 0 < len(hd) <= 48 and all(h.is_contiguous() and h.size(0) % 2 == 0 and h.dtype == hd[0].dtype and h.dtype in _DT for h in hd)
        _lib.require_cuda(*hd)
        dev = hd[0].device
        out = torch.empty(1, device=dev, dtype=torch.float32)
        nbytes = L.vits_feature_l1_workspace(len(hd))
        ws = K.workspace(nbytes, dev)
        items = _FeatureL1._items(hd, dens)
        _lib.check(L.vits_feature_l1(_DT[hd[0].dtype], ctypes.addressof(items), len(hd), out.data_ptr(), ws.data_ptr(), nbytes, _lib.stream_ptr()),
                   "vits_feature_l1")
        ctx.save_for_backward(*hd)
        ctx.dens = dens
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        import ctypes
        hd = ctx.saved_tensors
        gf = g.detach().float().contiguous().view(1)
        dhs = [torch.empty_like(h) for h in hd]
        items = _FeatureL1._items(hd, ctx.dens, dhs)
        _lib.check(_lib.lib().vits_feature_l1_bwd(_DT[hd[0].dtype], ctypes.addressof(items), len(hd), gf.data_ptr(), _lib.stream_ptr()),
                   "vits_feature_l1_bwd")
        return (None, *[d if ctx.needs_input_grad[i + 1] else None for i, d in enumerate(dhs)])


def feature_l1(hs, dens=None):
    """dens[i]: number of elements of ONE half of feature map i (differs from hs[i].numel() / 2 when hs[i] carries
    zero padding channels)."""
    dens = tuple(int(d) for d in (dens or [h.numel() // 2 for h in hs]))
    return _FeatureL1.apply(dens, *hs)


class _Lsgan(torch.autograd.Function):
    """losses.discriminator_loss / generator_loss over the logit tensors of all discriminators: two launches forward
    (partials, fixed-order final), one backward (csrc/reduce.hip, vits_lsgan_loss[_bwd])."""

    @staticmethod
    def forward(ctx, mode, *y8s):
        import ctypes
        L = _lib.lib()
        ys = [y.detach() for y in y8s]
        assert all(y.is_contiguous() and y.dim() == 3 and y.size(2) == 8 and y.dtype == ys[0].dtype for y in ys)
        _lib.require_cuda(*ys)
        items = (_lib.LsganItem * len(ys))()
        for it, y in zip(items, ys):
            it.y8, it.dy8, it.J, it.R = y.data_ptr(), None, y.size(0), y.size(1)
        out = torch.empty(1 + 2 * len(ys), device=ys[0].device, dtype=torch.float32)
        total = torch.empty((), device=ys[0].device, dtype=torch.float32)
        nbytes = L.vits_lsgan_workspace(len(ys))
        ws = K.workspace(nbytes, ys[0].device)
        _lib.check(L.vits_lsgan_loss(_DT[ys[0].dtype], ctypes.addressof(items), len(ys), mode, out.data_ptr(), total.data_ptr(), ws.data_ptr(),
                                     nbytes, _lib.stream_ptr()), "vits_lsgan_loss")
        ctx.save_for_backward(*ys)
        ctx.mode = mode
        ctx.mark_non_differentiable(out)                     # only the total (returned separately) carries a gradient
        return total, out

    @staticmethod
    def backward(ctx, g, _):
        import ctypes
        ys = ctx.saved_tensors
        dys = [torch.empty_like(y) for y in ys]
        items = (_lib.LsganItem * len(ys))()
        for it, y, dy in zip(items, ys, dys):
            it.y8, it.dy8, it.J, it.R = y.data_ptr(), dy.data_ptr(), y.size(0), y.size(1)
        gf = g.detach().float().contiguous().view(1)
        _lib.check(_lib.lib().vits_lsgan_loss_bwd(_DT[ys[0].dtype], ctypes.addressof(items), len(ys), ctx.mode, gf.data_ptr(), _lib.stream_ptr()),
                   "vits_lsgan_loss_bwd")
        return (None, *dys)


def lsgan(y8s, mode):
    """-> (total loss with gradient, per-term values [1 + 2 n]: [total, real_0, generated_0, real_1, ...] detached)"""
    return _Lsgan.apply(int(mode), *y8s)


class LogitLists(list):
    """The per-discriminator logit lists MultiPeriodDiscriminator returns (reference layout [b, t'] views), carrying the
    contiguous [J][R][8] tensors they are views of (`y8`: real items first, generated second) for the fused GAN losses."""
    y8 = None


class FmapLists(list):
    """The list of per-discriminator feature-map lists MultiPeriodDiscriminator returns (reference layout views), carrying
    the channels-last tensors they are views of (`cl`: real items first, generated items second) for the fused loss."""
    cl = None
