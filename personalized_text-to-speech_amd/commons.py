"""Mirror of the reference's commons.py for the functions on the hot path.
Same names, arguments and results; the per-item Python loops and host syncs are gone."""
import math

import torch
from torch.nn import functional as F

from .rng import noise


def init_weights(m, mean=0.0, std=0.01):
    # reference commons.py:8-11
    if m.__class__.__name__.find("Conv") != -1:
        m.weight.data.normal_(mean, std)


def get_padding(kernel_size, dilation=1):
    # reference commons.py:14-15
    return int((kernel_size * dilation - dilation) / 2)


def intersperse(lst, item):
    # reference commons.py:24-27
    result = [item] * (len(lst) * 2 + 1)
    result[1::2] = lst
    return result


def sequence_mask(length, max_length=None):
    # reference commons.py:124-128
    if max_length is None:
        max_length = length.max()
    x = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return x.unsqueeze(0) < length.unsqueeze(1)


def slice_segments(x, ids_str, segment_size=4, ids_scale=1):
    """x [b, d, t] -> [b, d, segment_size] starting at ids_str[b] * ids_scale (reference commons.py:48-57, there a Python loop
    over the batch with a host sync per item).  GPU tensors: one HIP launch (kernels.slice_segments, differentiable);
    host tensors (data pipeline, unit tests): one gather."""
    if x.is_cuda:
        from . import kernels
        return kernels.slice_segments(x, ids_str, segment_size, ids_scale)
    ids_str = ids_str * ids_scale
    idx = ids_str.view(-1, 1, 1) + torch.arange(segment_size, device=x.device).view(1, 1, -1)
    idx = idx.expand(-1, x.size(1), -1)
    return torch.gather(x, 2, idx)


def rand_slice_segments(x, x_lengths=None, segment_size=4):
    # reference commons.py:60-67: ids = (rand(b) * (len - seg + 1)).long()
    b, d, t = x.size()
    if x_lengths is None:
        x_lengths = t
    ids_str_max = x_lengths - segment_size + 1
    ids_str = (noise.rand(b, device=x.device, dtype=torch.float32) * ids_str_max).to(dtype=torch.long)
    return slice_segments(x, ids_str, segment_size), ids_str


def generate_path(duration, mask):
    """duration [b,1,t_x], mask [b,1,t_y,t_x] -> hard monotonic path (reference commons.py:131-146)."""
    if duration.is_cuda:
        from . import kernels
        return kernels.generate_path(duration, mask)
    b, _, t_y, t_x = mask.shape
    cum = torch.cumsum(duration, -1).view(b, t_x, 1)
    frame = torch.arange(t_y, dtype=duration.dtype, device=duration.device).view(1, 1, t_y)
    below = (frame < cum).to(mask.dtype)                      # sequence_mask(cum_duration, t_y)
    path = below - F.pad(below, (0, 0, 1, 0))[:, :-1]          # minus the previous token's mask
    return path.unsqueeze(1).transpose(2, 3) * mask


def fused_add_tanh_sigmoid_multiply(input_a, input_b, n_channels):
    """Same signature as the reference's TorchScript op (commons.py:103-110); the WN stack itself
    calls the fused kernel path in kernels.py."""
    n = int(n_channels[0]) if not isinstance(n_channels, int) else n_channels
    in_act = input_a + input_b
    return torch.tanh(in_act[:, :n, :]) * torch.sigmoid(in_act[:, n:, :])


def grad_norm_l2(parameters):
    """What reference commons.clip_grad_value_(params, None) returns (commons.py:149-164): the
    global L2 norm of the gradients — computed with one fused multi-tensor reduction and returned
    as a 0-d tensor (the reference does one .item() host sync per parameter tensor)."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.zeros(())
    if grads[0].is_cuda:
        from .optim import grad_norm_l2 as hip_norm             # csrc/adamw.hip in norm-only mode (vits_adamw + vits_gradnorm_final)
        return hip_norm(grads)
    norms = torch._foreach_norm(grads, 2)                       # host tensors (data-pipeline / unit-test use only)
    return torch.linalg.vector_norm(torch.stack(norms), 2)


def clip_grad_value_(parameters, clip_value, norm_type=2):
    # reference commons.py:149-164 (kept for call-surface compatibility)
    parameters = [p for p in ([parameters] if isinstance(parameters, torch.Tensor) else parameters) if p.grad is not None]
    if norm_type != 2:
        raise NotImplementedError("only the L2 norm the reference uses")
    total = grad_norm_l2(parameters)
    if clip_value is not None:
        for p in parameters:
            p.grad.data.clamp_(min=-float(clip_value), max=float(clip_value))
    return total


def kl_divergence(m_p, logs_p, m_q, logs_q):
    # reference commons.py:30-34
    kl = (logs_q - logs_p) - 0.5
    kl = kl + 0.5 * (torch.exp(2.0 * logs_p) + ((m_p - m_q) ** 2)) * torch.exp(-2.0 * logs_q)
    return kl


LOG_2PI = math.log(2 * math.pi)


def sum12(x):
    """torch.sum(x, [1, 2]) — on the GPU through this library's deterministic two-stage reduction (reduce.py: torch's
    multi-block path depends on a device memset, which a replayed hipGraph cannot rely on)."""
    if x.is_cuda:
        from . import reduce
        return reduce.sum12(x)
    return torch.sum(x, [1, 2])
