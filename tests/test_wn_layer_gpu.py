"""GPU: the one-launch WaveNet layer (csrc/wn_layer.hip, vits_wn_layer_fwd / _bwd; reference modules.py:157-176,
commons.py:103-110) against (i) the three-launch composition on vits_conv1d_cl it replaces and (ii) the oracle's WN on the CPU,
forward and every gradient, fp32 (exact-fp32 products: tight bars) and bf16."""
import importlib

import pytest
import torch

from model_util import rel_err

pytestmark = pytest.mark.gpu


def _wn(pkg, H, k, L, gin, dil_rate=1, seed=0):
    M = importlib.import_module("personalized_text-to-speech_amd.modules")
    torch.manual_seed(seed)
    wn = M.WN(H, k, dil_rate, L, gin_channels=gin)
    with torch.no_grad():
        for p in wn.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    return wn.cuda()


def _run(pkg, wn, x, lengths, g, fused, dtype):
    W = importlib.import_module("personalized_text-to-speech_amd.wn_cl")
    W.FUSED_LAYERS = fused
    try:
        wn.zero_grad()
        xa = x.clone().requires_grad_(True)
        ga = None if g is None else g.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            y = W.wn_forward_cl(wn, xa, lengths, ga)
        probe = torch.randn(y.shape, device=y.device, generator=torch.Generator(device="cuda").manual_seed(1))
        (y.float() * probe).sum().backward()
        grads = {n: p.grad.detach().clone() for n, p in wn.named_parameters()}
        return y.detach().float(), xa.grad.detach().float(), None if ga is None else ga.grad.detach().float(), grads
    finally:
        W.FUSED_LAYERS = False


CASES = [  # (b, t, H, k, L, gin, dilation_rate)
    (3, 150, 192, 5, 3, 256, 1),      # the step's shape family: H = 192, k = 5, speaker conditioning
    (2, 70, 16, 3, 2, 8, 1),          # the tiny fixture's family: one column tile, k = 3
    (2, 64, 96, 5, 2, 0, 2),          # no conditioning, dilation 1, 2; exactly one time tile
    (4, 129, 64, 3, 1, 0, 1),         # a single (= last) layer; one row into the third tile
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_layer_equals_composition(pkg, case, dtype):
    b, t, H, k, L, gin, dr = case
    wn = _wn(pkg, H, k, L, gin, dr)
    torch.manual_seed(3)
    lengths = torch.tensor([t, max(1, t - 37), max(1, t // 2), 5][:b], dtype=torch.int32, device="cuda")
    x = torch.randn(b, t, H, device="cuda") * (torch.arange(t, device="cuda")[None, :, None] < lengths[:, None, None])
    g = torch.randn(b, gin, 1, device="cuda") if gin else None
    ya, dxa, dga, ga = _run(pkg, wn, x, lengths, g, True, dtype)
    yb, dxb, dgb, gb = _run(pkg, wn, x, lengths, g, False, dtype)
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    assert rel_err(ya, yb) < tol
    assert float(ya[1, int(lengths[1]):].abs().max()) == 0.0 if int(lengths[1]) < t else True      # masked rows are exact zeros
    assert rel_err(dxa, dxb) < tol
    if g is not None:
        assert rel_err(dga, dgb) < tol
    for n in ga:
        assert rel_err(ga[n], gb[n]) < tol, n


def test_fused_layer_matches_oracle_fp32(pkg):
    """Against the oracle's WN (the restated reference graph) on the CPU, not only against this library's other kernels."""
    from oracle import vits_torch as O
    b, t, H, k, L, gin = 2, 90, 32, 5, 3, 16
    wn = _wn(pkg, H, k, L, gin, seed=4)
    sd = {"wn." + n: v.detach().cpu().clone() for n, v in wn.state_dict().items()}
    lengths = torch.tensor([90, 51], dtype=torch.int32, device="cuda")
    mask = (torch.arange(t, device="cuda")[None, None, :] < lengths[:, None, None]).float()
    torch.manual_seed(5)
    x = torch.randn(b, H, t, device="cuda") * mask
    g = torch.randn(b, gin, 1, device="cuda")
    W = importlib.import_module("personalized_text-to-speech_amd.wn_cl")
    W.FUSED_LAYERS = True
    try:
        y = W.wn_forward_cl(wn, x.transpose(1, 2).contiguous(), lengths, g).transpose(1, 2)
    finally:
        W.FUSED_LAYERS = False
    want = O.wn(sd, "wn", x.cpu(), mask.cpu(), g.cpu(), H, L, kernel=k)
    assert rel_err(y, want) < 1e-4
