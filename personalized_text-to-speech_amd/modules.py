"""Building blocks of the VITS generator — mirror of the reference's modules.py (same class names,
constructor arguments, parameter names and forward semantics), with the arithmetic routed through
kernels.py.  File:line citations point at the reference.
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

from . import commons
from . import kernels as K
from .commons import get_padding
from .transforms import piecewise_rational_quadratic_transform

LRELU_SLOPE = 0.1


# ---------------------------------------------------------------------------------------------
# Convolution modules.  The reference wraps nn.Conv1d / nn.ConvTranspose1d in the legacy
# torch.nn.utils.weight_norm, which yields the parameters `bias`, `weight_g` [c0,1,1] and
# `weight_v` (in that registration order).  These classes own exactly those parameters, initialise
# them by drawing from torch's generator the way nn.Conv1d.reset_parameters does (so a seeded
# construction consumes the same random stream), and hand them to the kernels.
# ---------------------------------------------------------------------------------------------
class Conv1d(nn.Module):
    """Plain conv (parameters `weight`, `bias`), forward through kernels.conv1d."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        proto = nn.Conv1d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups
        self.weight = nn.Parameter(proto.weight.data)
        self.bias = nn.Parameter(proto.bias.data) if bias else None

    def forward(self, x):
        if x.is_cuda and x.size(-1) == 1 and self.kernel_size == 1 and self.groups == 1 and self.in_channels % 4 == 0:
            # one-frame conditioning vectors (speaker embedding -> per-item bias): an exact-fp32 product on this library's kernel
            # (deterministic; the library convolution picked different algorithms from call to call)
            from . import wn_cl
            y = wn_cl.conv_cl(x.float().reshape(x.size(0), 1, x.size(1)), wn_cl.weight_of(self), self.bias, dtype=torch.float32)
            return y.reshape(x.size(0), -1, 1)
        return K.conv1d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)


def remove_weight_norm(module):
    """torch.nn.utils.remove_weight_norm for the WN* modules below (reference models.py:291-296, modules.py:178-184,225-229,
    254-256): g * v / ||v|| is folded into a plain `weight` parameter; afterwards state_dict holds `weight` instead of
    `weight_g` / `weight_v`, exactly like the reference's modules after the call, and the kernels take the plain weight."""
    if "weight_g" not in module._parameters:
        raise ValueError(f"weight_norm of 'weight' not found in {module}")
    w = K.weight_norm(module.weight_v, module.weight_g).detach().clone()
    del module._parameters["weight_g"], module._parameters["weight_v"]
    module._parameters["weight"] = nn.Parameter(w)          # (the class attribute `weight` is a property that returns it)
    return module


class WNConv1d(nn.Module):
    """weight_norm(Conv1d): parameters `bias`, `weight_g`, `weight_v`."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1):
        super().__init__()
        proto = nn.Conv1d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups
        self.bias = nn.Parameter(proto.bias.data)
        v = proto.weight.data
        self.weight_g = nn.Parameter(torch.linalg.vector_norm(v, 2, dim=(1, 2), keepdim=True))
        self.weight_v = nn.Parameter(v)

    @property
    def weight(self):
        plain = self._parameters.get("weight")              # after remove_weight_norm()
        if plain is not None:
            return plain
        from . import weight_arena
        h = weight_arena.handle_for(self, "torch")          # inside a scope: prepared by the multi-tensor kernel
        return h if h is not None else K.weight_norm(self.weight_v, self.weight_g)

    def forward(self, x):
        return K.conv1d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)


class WNConvTranspose1d(nn.Module):
    """weight_norm(ConvTranspose1d): weight_v is [c_in, c_out, k], the norm is per INPUT channel."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        proto = nn.ConvTranspose1d(in_channels, out_channels, kernel_size, stride, padding)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding = stride, padding
        self.bias = nn.Parameter(proto.bias.data)
        v = proto.weight.data
        self.weight_g = nn.Parameter(torch.linalg.vector_norm(v, 2, dim=(1, 2), keepdim=True))
        self.weight_v = nn.Parameter(v)

    @property
    def weight(self):
        plain = self._parameters.get("weight")              # after remove_weight_norm()
        return plain if plain is not None else K.weight_norm(self.weight_v, self.weight_g)

    def forward(self, x):
        return K.conv_transpose1d(x, self.weight, self.bias, self.stride, self.padding)


def _burn_init_weights(module_list):
    """The reference calls `.apply(init_weights)` on weight-normed convs (models.py:265,
    modules.py:199,209,240).  With legacy weight_norm that only overwrites the derived `weight`
    attribute, which the next forward recomputes from weight_g/weight_v — the parameters keep
    their default init, but the call does consume numel(weight) normal draws per conv.  Draw the
    same amount so a seeded construction stays aligned with the reference's random stream."""
    for m in module_list:
        torch.empty_like(m.weight_v).normal_(0.0, 0.01)


class LayerNorm(nn.Module):
    # modules.py:20-32 — normalises the channel dim of [b, c, t]
    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.channels = channels
        self.eps = eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))

    def forward(self, x):
        return K.layer_norm_c(x, self.gamma, self.beta, self.eps)


class DDSConv(nn.Module):
    """Dilated and depth-separable convolution (modules.py:70-108)."""

    def __init__(self, channels, kernel_size, n_layers, p_dropout=0.0):
        super().__init__()
        self.channels, self.kernel_size, self.n_layers, self.p_dropout = channels, kernel_size, n_layers, p_dropout
        self.drop = nn.Dropout(p_dropout)
        self.convs_sep = nn.ModuleList()
        self.convs_1x1 = nn.ModuleList()
        self.norms_1 = nn.ModuleList()
        self.norms_2 = nn.ModuleList()
        for i in range(n_layers):
            dilation = kernel_size ** i
            padding = (kernel_size * dilation - dilation) // 2
            self.convs_sep.append(Conv1d(channels, channels, kernel_size, groups=channels, dilation=dilation, padding=padding))
            self.convs_1x1.append(Conv1d(channels, channels, 1))
            self.norms_1.append(LayerNorm(channels))
            self.norms_2.append(LayerNorm(channels))

    def forward(self, x, x_mask, g=None):
        """Reference layout [b, c, t]; runs channels-last on the HIP row kernels + MFMA 1x1 convolution."""
        from . import wn_cl
        dtype = wn_cl.compute_dtype()
        y = self.forward_cl(x.transpose(1, 2).to(dtype), wn_cl.lengths_of(x_mask), x_mask.transpose(1, 2),
                            None if g is None else g.transpose(1, 2).to(dtype))
        return y.transpose(1, 2).to(x.dtype)

    def forward_cl(self, x, lengths, mask_cl, g=None, final_mask=True):
        """x [b, t, c] channels-last (compute dtype).  Per layer (modules.py:97-107): masked depth-wise
        dilated conv -> LayerNorm+GELU -> 1x1 conv (matrix cores) -> LayerNorm+GELU (+ residual when
        dropout is off) : four launches.  final_mask=False leaves the rows beyond the lengths unmasked at the output: every
        caller in this package feeds the result to a 1x1 convolution with mask_out=True, which is row-wise — conv(x * m) * m ==
        conv(x) * m, values and gradients — so the reference's trailing `x * x_mask` (modules.py:108) would be two wasted
        launches each way (inside the stack the depth-wise kernels read rows beyond the lengths as zero themselves)."""
        from . import rowops, wn_cl
        if g is not None:
            x = x + g
        drop = self.training and self.p_dropout > 0
        for i in range(self.n_layers):
            sep = self.convs_sep[i]
            y = rowops.dwconv(x, sep.weight, sep.bias, lengths, sep.dilation)
            y = rowops.ln_act(y, self.norms_1[i].gamma, self.norms_1[i].beta, None, self.norms_1[i].eps, 1)
            y = wn_cl.conv_cl(y, wn_cl.weight_of(self.convs_1x1[i]), wn_cl.bias_of(self.convs_1x1[i]), dtype=x.dtype)
            if drop:
                y = rowops.ln_act(y, self.norms_2[i].gamma, self.norms_2[i].beta, None, self.norms_2[i].eps, 1)
                x = x + self.drop(y)
            else:
                x = rowops.ln_act(y, self.norms_2[i].gamma, self.norms_2[i].beta, x, self.norms_2[i].eps, 1)
        return x * mask_cl.to(x.dtype) if final_mask else x


class WN(nn.Module):
    """Gated WaveNet stack (modules.py:111-184)."""

    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0):
        super().__init__()
        assert kernel_size % 2 == 1
        self.hidden_channels = hidden_channels
        self.kernel_size = (kernel_size,)
        self.dilation_rate, self.n_layers, self.gin_channels, self.p_dropout = dilation_rate, n_layers, gin_channels, p_dropout
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        self.drop = nn.Dropout(p_dropout)
        if gin_channels != 0:
            self.cond_layer = WNConv1d(gin_channels, 2 * hidden_channels * n_layers, 1)
        for i in range(n_layers):
            dilation = dilation_rate ** i
            padding = int((kernel_size * dilation - dilation) / 2)
            self.in_layers.append(WNConv1d(hidden_channels, 2 * hidden_channels, kernel_size, dilation=dilation, padding=padding))
            res_skip_channels = 2 * hidden_channels if i < n_layers - 1 else hidden_channels
            self.res_skip_layers.append(WNConv1d(hidden_channels, res_skip_channels, 1))

    def forward(self, x, x_mask, g=None, **kwargs):
        """x [b, H, t] (reference layout, zero beyond the mask) -> output * x_mask.  Runs as one
        autograd node over the channels-last HIP kernels (wn_cl.WNFn)."""
        from . import wn_cl
        out = wn_cl.wn_forward_cl(self, x.transpose(1, 2).contiguous(), wn_cl.lengths_of(x_mask), g)
        return out.transpose(1, 2).to(x.dtype)

    def remove_weight_norm(self):                              # modules.py:178-184
        if self.gin_channels != 0:
            remove_weight_norm(self.cond_layer)
        for l in list(self.in_layers) + list(self.res_skip_layers):
            remove_weight_norm(l)


class ResBlock1(nn.Module):
    # modules.py:187-229
    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.convs1 = nn.ModuleList([
            WNConv1d(channels, channels, kernel_size, 1, dilation=d, padding=get_padding(kernel_size, d)) for d in dilation])
        _burn_init_weights(self.convs1)
        self.convs2 = nn.ModuleList([
            WNConv1d(channels, channels, kernel_size, 1, dilation=1, padding=get_padding(kernel_size, 1)) for _ in dilation])
        _burn_init_weights(self.convs2)

    def forward(self, x, x_mask=None):
        for c1, c2 in zip(self.convs1, self.convs2):
            xt = K.leaky_relu(x, LRELU_SLOPE)
            if x_mask is not None:
                xt = xt * x_mask
            xt = c1(xt)
            xt = K.leaky_relu(xt, LRELU_SLOPE)
            if x_mask is not None:
                xt = xt * x_mask
            xt = c2(xt)
            x = xt + x
        if x_mask is not None:
            x = x * x_mask
        return x

    def remove_weight_norm(self):                              # modules.py:225-229
        for l in list(self.convs1) + list(self.convs2):
            remove_weight_norm(l)


class ResBlock2(nn.Module):
    # modules.py:232-256
    def __init__(self, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__()
        self.convs = nn.ModuleList([
            WNConv1d(channels, channels, kernel_size, 1, dilation=d, padding=get_padding(kernel_size, d)) for d in dilation])
        _burn_init_weights(self.convs)

    def forward(self, x, x_mask=None):
        for c in self.convs:
            xt = K.leaky_relu(x, LRELU_SLOPE)
            if x_mask is not None:
                xt = xt * x_mask
            xt = c(xt)
            x = xt + x
        if x_mask is not None:
            x = x * x_mask
        return x

    def remove_weight_norm(self):                              # modules.py:254-256
        for l in self.convs:
            remove_weight_norm(l)


class Log(nn.Module):
    # modules.py:259-267
    def forward(self, x, x_mask, reverse=False, **kwargs):
        if not reverse:
            y = torch.log(torch.clamp_min(x, 1e-5)) * x_mask
            return y, commons.sum12(-y)
        return torch.exp(x) * x_mask


class Flip(nn.Module):
    # modules.py:270-277
    def forward(self, x, *args, reverse=False, **kwargs):
        x = torch.flip(x, [1])
        if not reverse:
            return x, torch.zeros(x.size(0), dtype=x.dtype, device=x.device)
        return x


class ElementwiseAffine(nn.Module):
    # modules.py:280-295
    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.m = nn.Parameter(torch.zeros(channels, 1))
        self.logs = nn.Parameter(torch.zeros(channels, 1))

    def forward(self, x, x_mask, reverse=False, **kwargs):
        if not reverse:
            y = (self.m + torch.exp(self.logs) * x) * x_mask
            return y, commons.sum12(self.logs * x_mask)
        return (x - self.m) * torch.exp(-self.logs) * x_mask


class ResidualCouplingLayer(nn.Module):
    # modules.py:298-343
    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=0, gin_channels=0, mean_only=False):
        assert channels % 2 == 0, "channels should be divisible by 2"
        super().__init__()
        self.channels, self.hidden_channels, self.kernel_size = channels, hidden_channels, kernel_size
        self.dilation_rate, self.n_layers = dilation_rate, n_layers
        self.half_channels = channels // 2
        self.mean_only = mean_only
        self.pre = Conv1d(self.half_channels, hidden_channels, 1)
        self.enc = WN(hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=p_dropout, gin_channels=gin_channels)
        self.post = Conv1d(hidden_channels, self.half_channels * (2 - mean_only), 1)
        self.post.weight.data.zero_()
        self.post.bias.data.zero_()

    def forward(self, x, x_mask, g=None, reverse=False):
        y = self.forward_cl(x.transpose(1, 2).contiguous(), None, x_mask.transpose(1, 2), g, reverse)
        if not reverse:
            logdet = y[1] if y[1] is not None else torch.zeros(x.size(0), dtype=x.dtype, device=x.device)     # mean-only: log-determinant 0
            return y[0].transpose(1, 2), logdet
        return y.transpose(1, 2)

    def forward_cl(self, x, lengths, mask_cl, g=None, reverse=False, flip_after=False):
        """x [b, t, channels] channels-last; mask_cl [b, t, 1].  pre / WN / post run on the HIP kernels.  flip_after: also apply
        the modules.Flip that follows this layer in ResidualCouplingBlock (folded into the element-wise tail kernel)."""
        from . import rowops, wn_cl
        if lengths is None:
            lengths = mask_cl[:, :, 0].sum(-1).to(torch.int32)
        half = self.half_channels
        x0, x1 = x[..., :half], x[..., half:]
        h = wn_cl.conv_cl(x0, wn_cl.weight_of(self.pre), wn_cl.bias_of(self.pre), lengths, mask_out=True)
        h = wn_cl.wn_forward_cl(self.enc, h, lengths, g)
        stats = wn_cl.conv_cl(h, wn_cl.weight_of(self.post), wn_cl.bias_of(self.post), lengths, mask_out=True).to(x.dtype)
        if not self.mean_only:
            m, logs = stats[..., :half], stats[..., half:]
        else:
            m, logs = stats, None
        if not reverse:
            if logs is None:
                # mean-only (every VITS configuration): [x0, m + x1 * mask] and the flip in one launch, one more for the backward
                return rowops.coupling_tail(x, stats, lengths, half, flip_after), None
            x1 = m + (x1 * torch.exp(logs) if logs is not None else x1) * mask_cl
            logdet = commons.sum12(logs) if logs is not None else torch.zeros(x.size(0), dtype=x.dtype, device=x.device)
            out = torch.cat([x0, x1], -1)
            return (torch.flip(out, [2]) if flip_after else out), logdet
        x1 = (x1 - m) * (torch.exp(-logs) if logs is not None else 1.0) * mask_cl
        out = torch.cat([x0, x1], -1)
        return torch.flip(out, [2]) if flip_after else out


class ConvFlow(nn.Module):
    # modules.py:346-390
    def __init__(self, in_channels, filter_channels, kernel_size, n_layers, num_bins=10, tail_bound=5.0):
        super().__init__()
        self.in_channels, self.filter_channels, self.kernel_size, self.n_layers = in_channels, filter_channels, kernel_size, n_layers
        self.num_bins, self.tail_bound = num_bins, tail_bound
        self.half_channels = in_channels // 2
        self.pre = Conv1d(self.half_channels, filter_channels, 1)
        self.convs = DDSConv(filter_channels, kernel_size, n_layers, p_dropout=0.0)
        self.proj = Conv1d(filter_channels, self.half_channels * (num_bins * 3 - 1), 1)
        self.proj.weight.data.zero_()
        self.proj.bias.data.zero_()

    def forward(self, x, x_mask, g=None, reverse=False):
        from . import wn_cl
        dtype = wn_cl.compute_dtype()
        out = self.forward_cl(x.transpose(1, 2), wn_cl.lengths_of(x_mask), x_mask.transpose(1, 2),
                              None if g is None else g.transpose(1, 2).to(dtype), reverse)
        if not reverse:
            return out[0].transpose(1, 2), out[1]
        return out.transpose(1, 2)

    def forward_cl(self, x, lengths, mask_cl, g=None, reverse=False, swap=False):
        """x [b, t, 2] float32 channels-last; g [b, t, filter_channels] (compute dtype) or None.  swap: the roles of the two
        channels are exchanged (channel 1 conditions, channel 0 is transformed) — what a physical modules.Flip in front of this
        layer would achieve; the caller alternates it instead of flipping (and un-flipping) the state."""
        from . import rowops, wn_cl
        assert self.half_channels == 1, "the VITS duration flows transform one channel conditioned on the other"
        dtype = wn_cl.compute_dtype()
        c0, c1 = (1, 0) if swap else (0, 1)
        # Conv1d(1, C, 1) on the conditioning channel (+ the conditioning input of the DDSConv stack) as one row kernel
        h = rowops.flow_front(x, c0, self.pre.weight, self.pre.bias, g, dtype)
        h = self.convs.forward_cl(h, lengths, mask_cl, None, final_mask=False)       # (proj below is a masked 1x1 convolution)
        n_par = self.num_bins * 3 - 1
        pad = (-n_par) % 8                                                                   # 29 -> 32 output columns
        hp = wn_cl.conv_cl(h, wn_cl.weight_of(self.proj, pad_out=pad), wn_cl.bias_of(self.proj, pad), lengths, mask_out=True, dtype=dtype)
        out, logdet = rowops.flow_tail(x, hp, mask_cl, 1.0 / math.sqrt(self.filter_channels), reverse, self.tail_bound, c1)
        if not reverse:
            return out, logdet
        return out
