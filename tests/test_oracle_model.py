"""CPU: the torch restatement in oracle/vits_torch.py against the fixtures produced by running the
reference (tools/gen_golden_model.py).  Tolerances: fp32, relative to the tensor's max magnitude."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from model_util import load_tiny, noise_list, rel_err
from oracle import vits_torch as O

TOL = 2e-5


@pytest.fixture(scope="module")
def tiny():
    g, cfg = load_tiny()
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    return g, cfg, sd


@pytest.fixture(scope="module")
def ops():
    return np.load(os.path.join(ROOT, "tests", "golden", "ops.npz"))


def _in(g):
    t = lambda k: torch.from_numpy(g["in/" + k])
    return t("x"), t("x_lengths"), t("spec"), t("spec_lengths"), t("sid")


def test_forward(tiny):
    g, cfg, sd = tiny
    x, xl, spec, sl, sid = _in(g)
    o, l_length, attn, ids, xm, ym, (z, z_p, m_p, logs_p, m_q, logs_q), nc = O.synthesizer_forward(
        sd, cfg["model"], cfg["segment_size"], x, xl, spec, sl, sid, noise_list(g, "fwd"))
    assert rel_err(nc, g["fwd/neg_cent"]) < TOL
    assert np.array_equal(attn.numpy(), g["fwd/attn"]) and np.array_equal(ids.numpy(), g["fwd/ids_slice"])
    for name, t in dict(o=o, l_length=l_length, z=z, z_p=z_p, m_p=m_p, logs_p=logs_p, m_q=m_q, logs_q=logs_q).items():
        assert rel_err(t, g["fwd/" + name]) < TOL, name


def test_forward_gradients(tiny):
    g, cfg, sd = tiny
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    x, xl, spec, sl, sid = _in(g)
    o, l_length, attn, ids, xm, ym, (z, z_p, m_p, logs_p, m_q, logs_q), _ = O.synthesizer_forward(
        sd, cfg["model"], cfg["segment_size"], x, xl, spec, sl, sid, noise_list(g, "fwd"))
    probe = o.pow(2).mean() + l_length.sum() + O.kl_loss(z_p, logs_q, m_p, logs_p, ym)
    assert abs(float(probe) - float(g["fwd/probe"])) < 1e-4 * abs(float(g["fwd/probe"]))
    probe.backward()
    for k in [k for k in g.files if k.startswith("fwd/grad/")]:
        assert rel_err(sd[k[9:]].grad, g[k]) < 5e-5, k


def test_infer_and_vc(tiny):
    g, cfg, sd = tiny
    x, xl, spec, sl, sid = _in(g)
    with torch.no_grad():
        o, attn, ym, (z, z_p, m_p, logs_p) = O.synthesizer_infer(sd, cfg["model"], x, xl, sid, noise_list(g, "infer"),
                                                                 noise_scale=0.667, length_scale=1.1, noise_scale_w=0.8)
        assert np.array_equal(attn.numpy(), g["infer/attn"])
        for name, t in dict(o=o, y_mask=ym, z=z, z_p=z_p, m_p=m_p, logs_p=logs_p).items():
            assert rel_err(t, g["infer/" + name]) < TOL, name
        # voice conversion (models.py:525-533) composed from the same pieces
        m = cfg["model"]
        g_src, g_tgt = sd["emb_g.weight"][torch.tensor([0, 2])].unsqueeze(-1), sd["emb_g.weight"][torch.tensor([1, 0])].unsqueeze(-1)
        z, _, _, ym = O.posterior_encoder(sd, m, spec, sl, g_src, torch.from_numpy(g["vc/noise0"]))
        z_p = O.coupling_block(sd, m, z, ym, g_src)
        z_hat = O.coupling_block(sd, m, z_p, ym, g_tgt, reverse=True)
        o = O.generator(sd, m, z_hat * ym, g_tgt)
        for name, t in dict(o=o, z=z, z_p=z_p, z_hat=z_hat).items():
            assert rel_err(t, g["vc/" + name]) < TOL, name


@pytest.mark.parametrize("inverse", [False, True])
def test_spline(ops, inverse):
    tag = "spline_inv/" if inverse else "spline_fwd/"
    a = [torch.from_numpy(ops[tag + k]).requires_grad_(True) for k in ("x", "uw", "uh", "ud")]
    y, lad = O.rq_spline(a[0], a[1], a[2], a[3], inverse)
    assert rel_err(y, ops[tag + "y"]) < 1e-6 and rel_err(lad, ops[tag + "lad"]) < 2e-5
    (y * torch.cos(y)).sum().add((lad * 0.7).sum()).backward()
    for t, k in zip(a, ("gx", "guw", "guh", "gud")):
        assert rel_err(t.grad, ops[tag + k]) < 5e-5, k


def test_spline_roundtrip():
    torch.manual_seed(0)
    x = torch.rand(3, 1, 200) * 12 - 6
    uw, uh, ud = torch.randn(3, 1, 200, 10), torch.randn(3, 1, 200, 10), torch.randn(3, 1, 200, 9)
    y, lad = O.rq_spline(x, uw, uh, ud, False)
    x2, lad2 = O.rq_spline(y, uw, uh, ud, True)
    assert (x2 - x).abs().max() < 5e-4 and (lad + lad2).abs().max() < 5e-3      # fp32 quadratic root
    assert torch.equal(y[x.abs() > 5], x[x.abs() > 5])             # linear tails are the identity


@pytest.mark.parametrize("T", [3, 5, 9, 50])
def test_mha(ops, T):
    tag = f"mha{T}/"
    sd = {"a." + k[len(tag) + 3:]: torch.from_numpy(ops[k]) for k in ops.files if k.startswith(tag + "sd/")}
    x = torch.from_numpy(ops[tag + "x"])
    xm = O.sequence_mask(torch.from_numpy(ops[tag + "lens"]), T).unsqueeze(1).float()
    y, p = O.mha(sd, "a", x, xm.unsqueeze(2) * xm.unsqueeze(-1), 2)
    assert rel_err(y, ops[tag + "y"]) < TOL and rel_err(p, ops[tag + "p_attn"]) < TOL


def test_small_helpers(ops):
    assert np.array_equal(O.generate_path(torch.from_numpy(ops["genpath/dur"]), torch.from_numpy(ops["genpath/mask"])).numpy(), ops["genpath/path"])
    assert np.array_equal(O.slice_segments(torch.from_numpy(ops["slice/x"]), torch.from_numpy(ops["slice/ids"]), 5).numpy(), ops["slice/y"])
    t = lambda k: torch.from_numpy(ops["loss/" + k])
    fr, fg = [[t("fr0"), t("fr1")], [t("fr2")]], [[t("fg0"), t("fg1")], [t("fg2")]]
    assert abs(float(O.feature_loss(fr, fg)) - float(ops["loss/feature"])) < 1e-5
    assert abs(float(O.discriminator_loss([t("dr0"), t("dr1")], [t("dg0"), t("dg1")])) - float(ops["loss/disc"])) < 1e-5
    assert abs(float(O.generator_loss([t("dg0"), t("dg1")])) - float(ops["loss/gen"])) < 1e-5
    assert abs(float(O.kl_loss(t("kl_zp"), t("kl_lq"), t("kl_mp"), t("kl_lp"), t("kl_mask"))) - float(ops["loss/kl"])) < 1e-5


def test_spectrogram(ops):
    wav = torch.from_numpy(ops["stft/wav"])
    assert rel_err(O.spectrogram(wav, 1024, 256, 1024), ops["stft/spec_1024_256"]) < 1e-5
    assert rel_err(O.spectrogram(wav[:, :256], 64, 16, 64), ops["stft/spec_64_16"]) < 1e-5


def test_mel_basis_shape_and_norm():
    """librosa's filterbank is third-party: parity UNPINNED (no reference fixture); structure only."""
    b = O.mel_basis_slaney(22050, 1024, 80, 0.0, None)
    assert tuple(b.shape) == (80, 513) and b.dtype == torch.float32 and (b >= 0).all()
    assert (b.sum(1) > 0).all() and float(b[0, 0]) == 0.0


def test_step_losses_and_gradients_match_reference_step(pkg):
    """The oracle's restatement of one fine-tune iteration (train_losses + generator_losses around two AdamW updates,
    reference finetune_speaker_v2.py:174-232) against tests/golden/step_tiny.npz, produced by driving the reference's own
    modules, losses and optimizers (tools/gen_golden_step.py).  Pins the CPU baseline's step and the D-before-G ordering."""
    import json
    g = np.load(os.path.join(ROOT, "tests", "golden", "step_tiny.npz"))
    cfg = json.loads(bytes(g["config"]).decode())
    sd_g = {k[3:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("sd/")}
    for v in sd_g.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    torch.manual_seed(cfg["d_seed"])
    d = pkg.MultiPeriodDiscriminator(False)
    assert np.allclose([float(p.detach().double().sum()) for p in d.parameters()], g["d/param_checksum"], rtol=1e-6, atol=1e-6)
    sd_d = {k: v.detach().clone().requires_grad_(True) for k, v in d.state_dict().items()}
    t = lambda k: torch.from_numpy(g["in/" + k])
    batch = (t("x"), t("x_lengths"), t("spec"), t("spec_lengths"), t("y"), t("y_lengths"), t("sid"))
    noise = [torch.from_numpy(g[f"noise{i}"]) for i in range(int(g["n_noise"]))]
    hp = dict(cfg["data"]); hp.update(cfg["train"])
    kw = dict(betas=hp["betas"], eps=hp["eps"])
    opt_g = torch.optim.AdamW([v for v in sd_g.values() if v.requires_grad], hp["learning_rate"], **kw)
    opt_d = torch.optim.AdamW(list(sd_d.values()), hp["learning_rate"], **kw)
    loss_disc, rest = O.train_losses(sd_g, sd_d, cfg["model"], hp, batch, noise)
    opt_d.zero_grad(); loss_disc.backward()
    gn_d = torch.sqrt(sum(v.grad.double().pow(2).sum() for v in sd_d.values()))
    for k in [k for k in g.files if k.startswith("grad_d/")]:
        assert rel_err(sd_d[k[7:]].grad, g[k]) < 1e-4, k
    opt_d.step()
    loss_gen_all, parts = O.generator_losses(sd_d, hp, *rest)
    opt_g.zero_grad(); loss_gen_all.backward()
    gn_g = torch.sqrt(sum(v.grad.double().pow(2).sum() for v in sd_g.values() if v.grad is not None))
    got = dict(loss_disc=loss_disc, grad_norm_d=gn_d, grad_norm_g=gn_g, **parts)
    for k, v in got.items():
        ref = float(g["out/" + k])
        assert abs(float(v) - ref) <= 1e-4 * abs(ref), (k, float(v), ref)
    for k in [k for k in g.files if k.startswith("grad_g/")]:
        assert rel_err(sd_g[k[7:]].grad, g[k]) < 1e-4, k
    opt_g.step()
    for k in [k for k in g.files if k.startswith("new_g/")]:
        assert rel_err(sd_g[k[6:]], g[k]) < 1e-4, k
