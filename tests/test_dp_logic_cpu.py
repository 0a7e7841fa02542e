"""CPU: the channels-last stochastic duration predictor (DDSConv / ConvFlow / spline orchestration in
modules.py + models.py) with every HIP kernel emulated: forward (training nll+logq), its gradients and
the reverse (inference) path must equal the oracle."""
import importlib

import pytest
import torch

import cl_emul
from model_util import build_tiny, load_tiny, rel_err
from oracle import vits_torch as O


@pytest.fixture()
def emulated(pkg, monkeypatch):
    monkeypatch.setattr(pkg.kernels, "conv1d_cl_raw", cl_emul.conv1d_cl_raw)
    monkeypatch.setattr(pkg.kernels, "lrelu_mask_bwd", cl_emul.lrelu_mask_bwd)
    monkeypatch.setattr(pkg.kernels, "colsum", cl_emul.colsum)
    monkeypatch.setattr(pkg.kernels, "conv1d_cl_wgrad_raw", cl_emul.conv1d_cl_wgrad_raw)
    monkeypatch.setattr(pkg.kernels, "conv1d_cl_wgrad_batch", cl_emul.conv1d_cl_wgrad_batch)
    monkeypatch.setattr(pkg.kernels, "conv1d_cl_multi", cl_emul.conv1d_cl_multi)
    cl_emul.install_rowops(monkeypatch)
    return pkg


def test_sdp_forward_backward_reverse(emulated):
    pkg = emulated
    g_, cfg = load_tiny()
    net = build_tiny(pkg, g_, cfg)                      # eval mode: dropout off, as in the fixtures
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in net.state_dict().items()}
    torch.manual_seed(2)
    H = cfg["model"]["hidden_channels"]
    lens = torch.tensor([11, 7])
    xm = O.sequence_mask(lens, 11).unsqueeze(1).float()
    x = torch.randn(2, H, 11) * xm
    w = (torch.rand(2, 1, 11) * 4).round() * xm
    g = torch.randn(2, cfg["model"]["gin_channels"], 1)
    eps = torch.randn(2, 2, 11)
    want = O.sdp(sd, cfg["model"], x, xm, w, g, eps=eps)
    want.sum().backward()
    net.zero_grad()
    with pkg.rng.noise.replay([eps]):
        got = net.dp(x, xm, w, g=g)
    assert rel_err(got, want) < 2e-5
    got.sum().backward()
    for k, p in net.named_parameters():
        if k.startswith("dp.") and sd[k].grad is not None:
            assert rel_err(p.grad, sd[k].grad) < 1e-4, k
    with torch.no_grad(), pkg.rng.noise.replay([eps]):
        logw = net.dp(x, xm, g=g, reverse=True, noise_scale=0.8)
    assert rel_err(logw, O.sdp(sd, cfg["model"], x, xm, g=g, reverse=True, noise_scale=0.8, eps=eps)) < 2e-5
