"""Drop-in for the reference's ``monotonic_align`` package (monotonic_align/__init__.py:6-19).

Same signature and return convention as the reference wrapper — ``maximum_path(neg_cent, mask)``
returns the hard 0/1 alignment with ``neg_cent``'s device and dtype — but the DP runs on the GPU
through ``vits_mas_f32`` (include/vitsmi.h): no device->host copy, no stream synchronisation.
"""
import torch

from . import _lib

__all__ = ["maximum_path", "maximum_path_lengths"]


def maximum_path_lengths(neg_cent, t_ys, t_xs, out_dtype=None, status=None):
    """neg_cent [b,t_t,t_s] float32 CUDA; t_ys/t_xs int32 CUDA [b].  Returns path [b,t_t,t_s]."""
    _lib.require_cuda(neg_cent, t_ys, t_xs)
    if neg_cent.dim() != 3:
        raise ValueError("neg_cent must be [b, t_t, t_s]")
    out_dtype = out_dtype or neg_cent.dtype
    nc = neg_cent.detach()
    if nc.dtype != torch.float32:          # reference: .astype(np.float32), __init__.py:13
        nc = nc.float()
    nc = nc.contiguous()
    b, t_t, t_s = nc.shape
    kernel_dtype = {torch.float32: 0, torch.int32: 1}.get(out_dtype)
    path = torch.empty((b, t_t, t_s), device=nc.device, dtype=out_dtype if kernel_dtype is not None else torch.float32)
    t_ys = t_ys.to(torch.int32).contiguous()
    t_xs = t_xs.to(torch.int32).contiguous()
    e0 = _lib.timer.start("vits_mas_f32")
    rc = _lib.lib().vits_mas_f32(nc.data_ptr(), path.data_ptr(), kernel_dtype if kernel_dtype is not None else 0,
                                 t_ys.data_ptr(), t_xs.data_ptr(), b, t_t, t_s,
                                 status.data_ptr() if status is not None else None, _lib.stream_ptr())
    _lib.timer.stop("vits_mas_f32", e0, (b * t_t * t_s, 8.0 * b * t_t * t_s))       # units = (DP cells, bytes: 4 read + 4 written per cell)
    _lib.check(rc, "vits_mas_f32")
    return path if kernel_dtype is not None else path.to(out_dtype)


def maximum_path(neg_cent, mask):
    """neg_cent: [b, t_t, t_s]; mask: [b, t_t, t_s] (reference monotonic_align/__init__.py:6-19)."""
    t_ys = mask.sum(1)[:, 0].to(torch.int32)     # __init__.py:16
    t_xs = mask.sum(2)[:, 0].to(torch.int32)     # __init__.py:17
    return maximum_path_lengths(neg_cent, t_ys, t_xs, out_dtype=neg_cent.dtype)
