"""Piecewise rational-quadratic spline with linear tails — mirror of the reference's transforms.py
(piecewise_rational_quadratic_transform :12-43, unconstrained_rational_quadratic_spline :55-95,
rational_quadratic_spline :97-193).

Same call surface and results, restructured for the GPU: the reference gathers the in-interval
elements with boolean masks (a `nonzero` host sync per call, transforms.py:77-92); here every
element is evaluated with static shapes and the tails are selected at the end.  Consequence
(documented in DESIGN.md): a call whose inputs ALL lie outside [-tail_bound, tail_bound] returns
the identity instead of raising as the reference does (torch.min of an empty tensor, :105).
"""
import math

import torch
from torch.nn import functional as F

DEFAULT_MIN_BIN_WIDTH = 1e-3
DEFAULT_MIN_BIN_HEIGHT = 1e-3
DEFAULT_MIN_DERIVATIVE = 1e-3


def piecewise_rational_quadratic_transform(inputs, unnormalized_widths, unnormalized_heights,
                                           unnormalized_derivatives, inverse=False, tails=None, tail_bound=1.0,
                                           min_bin_width=DEFAULT_MIN_BIN_WIDTH, min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
                                           min_derivative=DEFAULT_MIN_DERIVATIVE):
    if tails != "linear":
        raise RuntimeError("only tails='linear' is on the VITS path (reference modules.py:384)")
    from . import kernels
    return kernels.rq_spline(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives,
                             inverse, float(tail_bound), min_bin_width, min_bin_height, min_derivative)


def _knots(unnormalized, lo, hi, min_size):
    # transforms.py:118-126 / :130-137: softmax -> floor -> cumsum -> affine -> pinned ends
    n = unnormalized.shape[-1]
    p = F.softmax(unnormalized, dim=-1)
    p = min_size + (1 - min_size * n) * p
    cum = torch.cumsum(p, dim=-1)
    cum = F.pad(cum, pad=(1, 0), mode="constant", value=0.0)
    cum = (hi - lo) * cum + lo
    cum = torch.cat([torch.full_like(cum[..., :1], lo), cum[..., 1:-1], torch.full_like(cum[..., :1], hi)], dim=-1)
    sizes = cum[..., 1:] - cum[..., :-1]
    return cum, sizes


def rq_spline_torch(inputs, uw, uh, ud, inverse, tail_bound, min_bin_width, min_bin_height, min_derivative):
    """Static-shape torch composition of the spline (autograd-differentiable)."""
    inside = (inputs >= -tail_bound) & (inputs <= tail_bound)            # transforms.py:65
    x = torch.where(inside, inputs, torch.zeros_like(inputs))            # any in-domain value for the tails
    const = math.log(math.exp(1 - min_derivative) - 1)                   # transforms.py:72-75
    edge = torch.full_like(ud[..., :1], const)
    ud = torch.cat([edge, ud, edge], dim=-1)

    cumwidths, widths = _knots(uw, -tail_bound, tail_bound, min_bin_width)
    cumheights, heights = _knots(uh, -tail_bound, tail_bound, min_bin_height)
    derivatives = min_derivative + F.softplus(ud)

    # searchsorted (transforms.py:47-52) adds eps to the LAST edge in place, and the mutated
    # tensor is what the later gathers read — only entry [-1] changes, which no gather touches
    # (bin_idx <= n-1 reads cum[..., :n]); the widths/heights were taken before the mutation.
    locs = cumheights if inverse else cumwidths
    locs_eps = torch.cat([locs[..., :-1], locs[..., -1:] + 1e-6], dim=-1)
    bin_idx = (torch.sum(x[..., None] >= locs_eps, dim=-1) - 1)[..., None]

    def take(t):
        return t.gather(-1, bin_idx)[..., 0]

    in_cw, in_w = take(cumwidths), take(widths)
    in_ch, in_h = take(cumheights), take(heights)
    delta = heights / widths
    in_delta = take(delta)
    in_d = take(derivatives)
    in_d1 = take(derivatives[..., 1:])

    if inverse:
        a = (x - in_ch) * (in_d + in_d1 - 2 * in_delta) + in_h * (in_delta - in_d)
        b = in_h * in_d - (x - in_ch) * (in_d + in_d1 - 2 * in_delta)
        c = -in_delta * (x - in_ch)
        disc = b.pow(2) - 4 * a * c
        root = (2 * c) / (-b - torch.sqrt(disc))
        out = root * in_w + in_cw
        t1mt = root * (1 - root)
        denom = in_delta + (in_d + in_d1 - 2 * in_delta) * t1mt
        dnum = in_delta.pow(2) * (in_d1 * root.pow(2) + 2 * in_delta * t1mt + in_d * (1 - root).pow(2))
        lad = -(torch.log(dnum) - 2 * torch.log(denom))
    else:
        theta = (x - in_cw) / in_w
        t1mt = theta * (1 - theta)
        num = in_h * (in_delta * theta.pow(2) + in_d * t1mt)
        denom = in_delta + (in_d + in_d1 - 2 * in_delta) * t1mt
        out = in_ch + num / denom
        dnum = in_delta.pow(2) * (in_d1 * theta.pow(2) + 2 * in_delta * t1mt + in_d * (1 - theta).pow(2))
        lad = torch.log(dnum) - 2 * torch.log(denom)
    out = torch.where(inside, out, inputs)                               # linear tails: identity
    lad = torch.where(inside, lad, torch.zeros_like(lad))
    return out, lad
