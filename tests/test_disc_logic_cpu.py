"""CPU: the period discriminators on the channels-last kernels (kernels emulated): logits, feature maps
(reference layout) and gradients wrt the input waveform and the parameters equal the oracle."""
import pytest
import torch

import cl_emul
from model_util import rel_err
from oracle import vits_torch as O


@pytest.fixture()
def emulated(pkg, monkeypatch):
    monkeypatch.setattr(pkg.kernels, "conv1d_cl_raw", cl_emul.conv1d_cl_raw)
    monkeypatch.setattr(pkg.kernels, "lrelu_mask_bwd", cl_emul.lrelu_mask_bwd)
    monkeypatch.setattr(pkg.kernels, "colsum", cl_emul.colsum)
    monkeypatch.setattr(pkg.kernels, "conv1d_cl_wgrad_raw", cl_emul.conv1d_cl_wgrad_raw)
    monkeypatch.setattr(pkg.kernels, "conv1d_cl_wgrad_batch", cl_emul.conv1d_cl_wgrad_batch)
    monkeypatch.setattr(pkg.kernels, "conv1d_cl_multi", cl_emul.conv1d_cl_multi)
    cl_emul.install_arena_emulation(monkeypatch)
    cl_emul.install_disc(monkeypatch)
    return pkg


@pytest.mark.parametrize("period", [2, 3, 5])
def test_discriminator_p(emulated, period):
    pkg = emulated
    torch.manual_seed(period)
    d = pkg.models.DiscriminatorP(period)
    sd = {"d." + k: v.detach().clone().requires_grad_(True) for k, v in d.state_dict().items()}
    x = (torch.rand(2, 1, 500) * 2 - 1)
    xo, xp = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    lo, fo = O.disc_p(sd, "d", xo, period)
    lp, fp = d.forward_hip(xp)
    assert lp.shape == lo.shape and rel_err(lp, lo) < 1e-5
    for a, b in zip(fp, fo):
        assert a.shape == b.shape and rel_err(a, b) < 1e-5
    probe = [torch.randn_like(f) for f in fo]
    sum((f * q).sum() for f, q in zip(fo, probe)).backward()
    sum((f.float() * q).sum() for f, q in zip(fp, probe)).backward()
    assert rel_err(xp.grad, xo.grad) < 2e-5
    for k, p in d.named_parameters():
        assert rel_err(p.grad, sd["d." + k].grad) < 5e-5, k


def test_mpd_through_arena(emulated, monkeypatch):
    """MultiPeriodDiscriminator as one autograd node (disc_cl.DiscFn) with the weight-norm of all 37 layers coming from the
    arena: outputs and parameter gradients equal the oracle."""
    pkg = emulated
    torch.manual_seed(0)
    d = pkg.MultiPeriodDiscriminator(False)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in d.state_dict().items()}
    y, y_hat = torch.rand(1, 1, 600) * 2 - 1, torch.rand(1, 1, 600) * 2 - 1
    rs, gs, fr, fg = d(y, y_hat)
    ro, go, fro, fgo = O.mpd(sd, y, y_hat)
    for a, b in zip(rs + gs, ro + go):
        assert rel_err(a, b) < 1e-5
    (sum(g.pow(2).mean() for g in gs) + sum(f.abs().mean() for fm in fg for f in fm)).backward()
    (sum(g.pow(2).mean() for g in go) + sum(f.abs().mean() for fm in fgo for f in fm)).backward()
    for k, p in d.named_parameters():
        assert rel_err(p.grad, sd[k].grad) < 5e-5, k


@pytest.mark.parametrize("in_scope", [False, True])
def test_discriminator_s_channels_last(emulated, in_scope):
    """DiscriminatorS on the channels-last kernels (grouped layers as dense block-diagonal operands, compact weight
    gradients) against the oracle: logits, feature maps and every parameter gradient."""
    pkg = emulated
    import importlib
    WA = importlib.import_module("personalized_text-to-speech_amd.weight_arena")
    torch.manual_seed(3)
    mpd = pkg.MultiPeriodDiscriminator(False)
    d = mpd.discriminators[0]
    sd = {"d." + k: v.detach().clone().requires_grad_(True) for k, v in d.state_dict().items()}
    x = torch.rand(2, 1, 1200) * 2 - 1
    if in_scope:
        with WA.scope(mpd, type(mpd)._arena_specs):
            la, fa = d.forward_hip(x)
    else:
        la, fa = d.forward_hip(x)
    lo, fo = O.disc_s(sd, "d", x)
    assert rel_err(la, lo) < 1e-5
    for a, b in zip(fa, fo):
        assert a.shape == b.shape and rel_err(a, b) < 1e-5
    (la.pow(2).mean() + sum(f.abs().mean() for f in fa)).backward()
    (lo.pow(2).mean() + sum(f.abs().mean() for f in fo)).backward()
    for k, p in d.named_parameters():
        assert rel_err(p.grad, sd["d." + k].grad) < 5e-5, k


def test_mpd_generator_step_half_batch(emulated):
    """Frozen discriminator (the generator step): only the generated half of the batch runs backward (n_lo = b); the gradient
    wrt y_hat equals the oracle's, the real half contributes nothing (reference losses.py:11 detaches it)."""
    pkg = emulated
    torch.manual_seed(1)
    d = pkg.MultiPeriodDiscriminator(False)
    sd = {k: v.detach().clone() for k, v in d.state_dict().items()}
    for p in d.parameters():
        p.requires_grad_(False)
    y = torch.rand(2, 1, 700) * 2 - 1
    ya, yb = (torch.rand(2, 1, 700) * 2 - 1).requires_grad_(True), None
    yb = ya.detach().clone().requires_grad_(True)
    rs, gs, fr, fg = d(y, ya)
    (pkg.losses.generator_loss(gs)[0] + sum((a.detach() - b).abs().mean() for x, z in zip(fr, fg) for a, b in zip(x, z))).backward()
    ro, go, fro, fgo = O.mpd(sd, y, yb)
    (O.generator_loss(go) + sum((a.detach() - b).abs().mean() for x, z in zip(fro, fgo) for a, b in zip(x, z))).backward()
    assert rel_err(ya.grad, yb.grad) < 2e-5
