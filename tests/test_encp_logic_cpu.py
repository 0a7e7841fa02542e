"""CPU: channels-last text encoder orchestration (attentions.Encoder on conv_cl / ln_act, kernels emulated)
against the oracle: outputs and every enc_p parameter gradient."""
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
    cl_emul.install_attention(monkeypatch)
    return pkg


def test_text_encoder(emulated):
    pkg = emulated
    g_, cfg = load_tiny()
    net = build_tiny(pkg, g_, cfg)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in net.state_dict().items()}
    x = torch.from_numpy(g_["in/x"]); xl = torch.from_numpy(g_["in/x_lengths"])
    h_o, m_o, logs_o, xm = O.text_encoder(sd, cfg["model"], x, xl)
    probe = torch.randn_like(h_o)
    ((h_o * probe).sum() + (m_o * logs_o).sum()).backward()
    net.zero_grad()
    h_p, m_p, logs_p, xm_p = net.enc_p(x, xl)
    ((h_p * probe).sum() + (m_p * logs_p).sum()).backward()
    for a, b in ((h_p, h_o), (m_p, m_o), (logs_p, logs_o)):
        assert rel_err(a, b) < 1e-5
    for k, p in net.named_parameters():
        if k.startswith("enc_p."):
            # conv_k.bias has an exactly-zero gradient (softmax is invariant to a per-query constant): compare absolutely there
            assert rel_err(p.grad, sd[k].grad) < 5e-5 or float((p.grad - sd[k].grad).abs().max()) < 1e-6, k
