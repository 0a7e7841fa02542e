"""GPU: the product modules (HIP alignment kernel + device ops) against the fixtures produced by
running the reference, within the 1e-3 relative tolerance BASELINE.json states for waveform/mel
tensors (fp32 mode; alignment indices bit-exact)."""
import numpy as np
import pytest
import torch

from model_util import build_tiny, inputs, load_tiny, noise_list, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def tiny(pkg):
    g, cfg = load_tiny()
    return g, cfg, build_tiny(pkg, g, cfg, "cuda:0")


def test_forward_matches_reference(pkg, tiny):
    g, cfg, net = tiny
    x, xl, spec, sl, sid = inputs(g, "cuda:0")
    with pkg.rng.noise.replay(noise_list(g, "fwd")):
        o, l_length, attn, ids, xm, ym, (z, z_p, m_p, logs_p, m_q, logs_q) = net(x, xl, spec, sl, sid)
    assert np.array_equal(attn.cpu().numpy(), g["fwd/attn"])                # alignment: bit-exact
    assert np.array_equal(ids.cpu().numpy(), g["fwd/ids_slice"])
    for name, t in dict(o=o, l_length=l_length, z=z, z_p=z_p, m_p=m_p, logs_p=logs_p, m_q=m_q, logs_q=logs_q).items():
        assert rel_err(t, g["fwd/" + name]) < TOL, name
    probe = o.pow(2).mean() + l_length.sum() + pkg.losses.kl_loss(z_p, logs_q, m_p, logs_p, ym)
    net.zero_grad()
    probe.backward()
    params = dict(net.named_parameters())
    for k in [k for k in g.files if k.startswith("fwd/grad/")]:
        assert rel_err(params[k[9:]].grad, g[k]) < TOL, k


def test_alignment_from_reference_neg_cent_is_bit_exact(pkg, tiny):
    """End-to-end gate (ii) of SURVEY §7: the path computed by the HIP kernel from the REFERENCE's
    own neg_cent equals the reference's path exactly."""
    g, cfg, net = tiny
    nc = torch.from_numpy(g["fwd/neg_cent"]).cuda()
    mask = (torch.from_numpy(g["fwd/y_mask"]).unsqueeze(-1) * torch.from_numpy(g["fwd/x_mask"]).unsqueeze(2)).squeeze(1).cuda()
    path = pkg.monotonic_align.maximum_path(nc, mask)
    assert np.array_equal(path.cpu().numpy(), g["fwd/attn"][:, 0])


def test_infer_and_voice_conversion(pkg, tiny):
    g, cfg, net = tiny
    x, xl, spec, sl, sid = inputs(g, "cuda:0")
    with torch.no_grad(), pkg.rng.noise.replay(noise_list(g, "infer")):
        o, attn, ym, (z, z_p, m_p, logs_p) = net.infer(x, xl, sid, noise_scale=0.667, length_scale=1.1, noise_scale_w=0.8)
    assert np.array_equal(attn.cpu().numpy(), g["infer/attn"])
    for name, t in dict(o=o, y_mask=ym, z=z, z_p=z_p, m_p=m_p, logs_p=logs_p).items():
        assert rel_err(t, g["infer/" + name]) < TOL, name
    with torch.no_grad(), pkg.rng.noise.replay([torch.from_numpy(g["vc/noise0"])]):
        o, _, (z, z_p, z_hat) = net.voice_conversion(spec, sl, torch.tensor([0, 2]).cuda(), torch.tensor([1, 0]).cuda())
    for name, t in dict(o=o, z=z, z_p=z_p, z_hat=z_hat).items():
        assert rel_err(t, g["vc/" + name]) < TOL, name


def test_discriminator_matches_reference_seeded(pkg):
    import os
    from conftest import ROOT
    ops = np.load(os.path.join(ROOT, "tests", "golden", "ops.npz"))
    torch.manual_seed(99)
    d = pkg.MultiPeriodDiscriminator(False).cuda().eval()
    with torch.no_grad():
        rs, gs, fr, fg = d(torch.from_numpy(ops["disc/y"]).cuda(), torch.from_numpy(ops["disc/y_hat"]).cuda())
    for i in range(6):
        assert rel_err(rs[i], ops[f"disc/logit_r{i}"]) < TOL and rel_err(gs[i], ops[f"disc/logit_g{i}"]) < TOL
        got = np.array([float(f.abs().mean()) for f in fr[i]])
        assert np.allclose(got, ops[f"disc/fmap_r{i}_absmean"], rtol=1e-3)


def test_spectrogram_matches_reference(pkg):
    import os
    from conftest import ROOT
    ops = np.load(os.path.join(ROOT, "tests", "golden", "ops.npz"))
    wav = torch.from_numpy(ops["stft/wav"]).cuda()
    assert rel_err(pkg.mel_processing.spectrogram_torch(wav, 1024, 22050, 256, 1024), ops["stft/spec_1024_256"]) < 1e-4
    assert rel_err(pkg.mel_processing.spectrogram_torch(wav[:, :256], 64, 22050, 16, 64), ops["stft/spec_64_16"]) < 1e-4


def test_train_step_runs_and_learns(pkg):
    """Two fp32 steps of the full-size fine-tune step on a small synthetic batch: finite losses,
    discriminator loss goes down, every G parameter that should train received a gradient."""
    from importlib import import_module
    cfgs = import_module("personalized_text-to-speech_amd.configs")
    tr = import_module("personalized_text-to-speech_amd.train")
    hps = cfgs.get("modified_finetune_speaker")
    ft = tr.FineTuner(hps, "cuda:0", amp=False)
    batch = tr.synthetic_batch(hps, 2, (60, 80), "cuda:0")
    a = {k: float(v) for k, v in ft.step(batch).items()}
    b = {k: float(v) for k, v in ft.step(batch).items()}
    assert all(np.isfinite(list(a.values()))) and all(np.isfinite(list(b.values())))
    assert b["loss_disc"] < a["loss_disc"]
    missing = [n for n, p in ft.net_g.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    assert not missing, missing[:5]


def test_mel_of_generated_audio_matches_oracle(pkg):
    """mel_spectrogram_torch through the exact-fp32 MFMA DFT vs the oracle's torch.stft path, values and
    the gradient wrt the waveform (it sits on the generator's critical path, finetune_speaker_v2.py:192-201)."""
    from oracle import vits_torch as O
    torch.manual_seed(0)
    y = (torch.rand(3, 8192, device="cuda:0") * 1.6 - 0.8)
    ya, yb = y.clone().requires_grad_(True), y.clone().requires_grad_(True)
    got = pkg.mel_processing.mel_spectrogram_torch(ya, 1024, 80, 22050, 256, 1024, 0.0, None)
    basis = O.mel_basis_slaney(22050, 1024, 80, 0.0, None).cuda()
    want = O.spec_to_mel(O.spectrogram(yb, 1024, 256, 1024), basis)
    assert got.shape == want.shape == (3, 80, 32)
    assert rel_err(got, want) < 1e-4
    probe = torch.randn_like(want)
    (got * probe).sum().backward()
    (want * probe).sum().backward()
    assert rel_err(ya.grad, yb.grad) < 1e-3


def test_period_discriminator_hip_path(pkg):
    """The channels-last HIP implementation of DiscriminatorP (off by default) against the oracle on the CPU
    (MIOpen's fp32 solvers are find-mode dependent and too noisy to serve as the yardstick for gradients)."""
    from oracle import vits_torch as O
    torch.manual_seed(4)
    d = pkg.models.DiscriminatorP(3).cuda()
    sd = {"d." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in d.state_dict().items()}
    x = (torch.rand(4, 1, 4096) * 2 - 1)
    xa, xb = x.cuda().requires_grad_(True), x.clone().requires_grad_(True)
    la, fa = d.forward_hip(xa)
    lb, fb = O.disc_p(sd, "d", xb, 3)
    assert rel_err(la, lb) < 1e-4
    for a, b in zip(fa, fb):
        assert a.shape == b.shape and rel_err(a, b) < 1e-4
    la.pow(2).sum().backward()
    lb.pow(2).sum().backward()
    assert rel_err(xa.grad, xb.grad) < 1e-3
    for k, p in d.named_parameters():
        assert rel_err(p.grad, sd["d." + k].grad) < 1e-3, k


def test_captured_step_replays_reproducibly(pkg):
    """The whole fine-tune step captured as one hipGraph: two replays from the same state give the same losses and
    parameters, and agree with the eager step (FineTuner.verify_replay) — the guard bench.py runs before timing."""
    from importlib import import_module
    cfgs = import_module("personalized_text-to-speech_amd.configs")
    tr = import_module("personalized_text-to-speech_amd.train")
    hps = cfgs.get("modified_finetune_speaker")
    ft = tr.FineTuner(hps, "cuda:0", amp=True)
    batch = tr.synthetic_batch(hps, 4, (100, 160), "cuda:0")
    # text encoder, duration predictor and (between them, on the same stream) alignment + prior expansion as side-stream branches
    assert ft.side_branches == {"enc_p", "dp", "prior"}
    ft.capture(batch, warmup=2)
    for _ in range(3):                                # a race between branches would be intermittent
        losses = ft.verify_replay()
        assert all(np.isfinite(list(losses.values())))
        assert max(ft.replay_vs_eager.values()) <= 1e-6, ft.replay_vs_eager
    for _ in range(3):
        out = ft.replay()
    assert all(np.isfinite([float(v) for v in out.values()]))


def test_lr_decay_reaches_the_replayed_graph(pkg):
    """ExponentialLR per epoch (finetune_speaker_v2.py:157-158): after epoch_end() a REPLAYED step must update with the decayed
    learning rate — the AdamW kernel reads lr from device memory, which only FlatAdamW._sync_lr refreshes."""
    from importlib import import_module
    cfgs = import_module("personalized_text-to-speech_amd.configs")
    tr = import_module("personalized_text-to-speech_amd.train")
    hps = cfgs.get("modified_finetune_speaker")
    ft = tr.FineTuner(hps, "cuda:0", amp=False)
    ft.sched_g.gamma = ft.sched_d.gamma = 0.5          # a decay large enough to see in one step
    batch = tr.synthetic_batch(hps, 2, (60, 80), "cuda:0")
    ft.capture(batch, warmup=2)
    ts = ft._state_tensors()
    snap = [t.detach().clone() for t in ts]
    rng = torch.cuda.get_rng_state(ft.device)

    def delta(fn):
        with torch.no_grad():
            for t, s in zip(ts, snap):
                t.copy_(s)
        ft.optim_g._sync_lr(force=True); ft.optim_d._sync_lr(force=True)
        torch.cuda.set_rng_state(rng, ft.device)
        fn()
        torch.cuda.synchronize()
        return (ft.optim_g.flat_p - snap[0]).clone(), (ft.optim_d.flat_p - snap[4]).clone()

    g0, d0 = delta(ft.replay)
    lr0 = ft.optim_g.param_groups[0]["lr"]
    ft.epoch_end()
    assert abs(ft.optim_g.param_groups[0]["lr"] - 0.5 * lr0) < 1e-12
    assert float(ft.optim_g.dev_state[0]) == pytest.approx(0.5 * lr0) and float(ft.optim_d.dev_state[0]) == pytest.approx(0.5 * lr0)
    # from the same state the replay now moves every parameter by what an eager step at the decayed rate moves it
    with torch.no_grad():
        for t, s in zip(ts, snap):
            t.copy_(s)
    ft.optim_g._sync_lr(force=True); ft.optim_d._sync_lr(force=True)
    torch.cuda.set_rng_state(rng, ft.device)
    ft.replay(); torch.cuda.synchronize()
    g1 = (ft.optim_g.flat_p - snap[0]).clone()
    with torch.no_grad():
        for t, s in zip(ts, snap):
            t.copy_(s)
    ft.optim_g._sync_lr(force=True); ft.optim_d._sync_lr(force=True)     # (the snapshot also restored the device copy of the OLD rate)
    torch.cuda.set_rng_state(rng, ft.device)
    ft.step(batch); torch.cuda.synchronize()
    g2 = ft.optim_g.flat_p - snap[0]
    assert float((g1 - g2).abs().max()) <= 1e-6 * float(g2.abs().max())
    # first AdamW step: |delta| ~ lr per element, so the decayed replay moves about half as far as the undecayed one
    ratio = float(g1.abs().sum() / g0.abs().sum())
    assert 0.45 < ratio < 0.55, ratio


def test_text_embedding_gradient_matches_torch(pkg):
    torch.manual_seed(8)
    w = torch.randn(111, 192, device="cuda", requires_grad=True)
    idx = torch.randint(0, 111, (16, 201), device="cuda")
    g = torch.randn(16, 201, 192, device="cuda")
    a = pkg.models._embedding_scaled(idx, w, 13.856)
    b = torch.nn.functional.embedding(idx, w) * 13.856
    assert torch.equal(a, b)
    (ga,) = torch.autograd.grad(a, w, g)
    (gb,) = torch.autograd.grad(b, w, g)
    assert float((ga - gb).abs().max() / gb.abs().max()) < 1e-5


def test_scale_discriminator_and_edge_layers(pkg):
    """DiscriminatorS through disc_cl.DiscFn (first-layer / conv_post bandwidth kernels, grouped layers) against the oracle
    on the CPU: logits, feature maps, input and parameter gradients."""
    from oracle import vits_torch as O
    torch.manual_seed(5)
    d = pkg.models.DiscriminatorS().cuda()
    sd = {"d." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in d.state_dict().items()}
    x = (torch.rand(3, 1, 3000) * 2 - 1)
    xa, xb = x.cuda().requires_grad_(True), x.clone().requires_grad_(True)
    la, fa = d.forward_hip(xa)
    lb, fb = O.disc_s(sd, "d", xb)
    assert rel_err(la, lb) < 1e-4
    for a, b in zip(fa, fb):
        assert a.shape == b.shape and rel_err(a, b) < 1e-4
    (la.pow(2).sum() + sum(f.abs().mean() for f in fa)).backward()
    (lb.pow(2).sum() + sum(f.abs().mean() for f in fb)).backward()
    assert rel_err(xa.grad, xb.grad) < 1e-3
    for k, p in d.named_parameters():
        assert rel_err(p.grad, sd["d." + k].grad) < 1e-3, k


@pytest.mark.parametrize("period", [2, 7, 11])
def test_period_discriminator_odd_lengths(pkg, period):
    """Reflect pad folded into the first-layer kernels: lengths that are not a multiple of the period (8192 is not one of
    3, 5, 7, 11), forward and the gradient wrt the waveform."""
    from oracle import vits_torch as O
    torch.manual_seed(period)
    d = pkg.models.DiscriminatorP(period).cuda()
    sd = {"d." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in d.state_dict().items()}
    x = (torch.rand(2, 1, 8192) * 2 - 1)
    xa, xb = x.cuda().requires_grad_(True), x.clone().requires_grad_(True)
    la, fa = d.forward_hip(xa)
    lb, fb = O.disc_p(sd, "d", xb, period)
    assert la.shape == lb.shape and rel_err(la, lb) < 1e-4
    for a, b in zip(fa, fb):
        assert a.shape == b.shape and rel_err(a, b) < 1e-4
    (la.pow(2).sum() + sum(f.abs().mean() for f in fa)).backward()
    (lb.pow(2).sum() + sum(f.abs().mean() for f in fb)).backward()
    assert rel_err(xa.grad, xb.grad) < 1e-3
    for k, p in d.named_parameters():
        assert rel_err(p.grad, sd["d." + k].grad) < 1e-3, k


def test_mpd_generator_step_half_batch(pkg):
    """Frozen discriminators (the generator step): backward on the generated half only; d/d y_hat equals the oracle's."""
    from oracle import vits_torch as O
    torch.manual_seed(1)
    d = pkg.MultiPeriodDiscriminator(False).cuda()
    sd = {k: v.detach().cpu().clone() for k, v in d.state_dict().items()}
    for p in d.parameters():
        p.requires_grad_(False)
    y = torch.rand(2, 1, 2048) * 2 - 1
    yh = torch.rand(2, 1, 2048) * 2 - 1
    ya, yb = yh.cuda().requires_grad_(True), yh.clone().requires_grad_(True)
    rs, gs, fr, fg = d(y.cuda(), ya)
    (pkg.losses.generator_loss(gs)[0] + pkg.losses.feature_loss(fr, fg)).backward()
    ro, go, fro, fgo = O.mpd(sd, y, yb)
    (O.generator_loss(go) + O.feature_loss(fro, fgo)).backward()
    assert rel_err(ya.grad, yb.grad) < 1e-3


# ---------------------------------------------------------------------------------------------------------------------------
# Fixtures of tools/gen_golden_misc.py (tests/golden/misc.npz + ref_G_tiny.pth, produced by running the reference)
def _misc():
    import os
    from conftest import ROOT
    return np.load(os.path.join(ROOT, "tests", "golden", "misc.npz")), os.path.join(ROOT, "tests", "golden", "ref_G_tiny.pth")


def test_evaluate_matches_reference(pkg):
    """train.evaluate (finetune_speaker_v2.py:313-368) on the checkpoint the reference wrote: the generated waveform cut to
    y_hat_lengths and the alignment equal what the reference's evaluate() computes before its plotting calls; the mels are the
    product's mel of those waveforms (the mel filter bank itself is parity-unpinned)."""
    from importlib import import_module
    tr = import_module("personalized_text-to-speech_amd.train")
    cfgs = import_module("personalized_text-to-speech_amd.configs")
    mel = import_module("personalized_text-to-speech_amd.mel_processing")
    m, ckpt = _misc()
    g, cfg = load_tiny()
    net = pkg.SynthesizerTrn(cfg["n_vocab"], cfg["spec_channels"], cfg["segment_size"], n_speakers=cfg["n_speakers"], **cfg["model"])
    pkg.utils.load_checkpoint(ckpt, net, None)
    net = net.to("cuda:0").train()
    hps = cfgs.HParams({"data": {"hop_length": 16, "filter_length": 32, "win_length": 32, "n_mel_channels": 8, "sampling_rate": 22050,
                                 "mel_fmin": 0.0, "mel_fmax": None}})
    x, x_len, spec, spec_len, sid = inputs(g, "cuda:0")
    torch.manual_seed(0)
    y = torch.rand(2, 1, 24 * 16, device="cuda:0") - 0.5
    y_len = spec_len * 16
    noise = [torch.from_numpy(m[f"eval/noise{i}"]) for i in range(int(m["eval/n_noise"]))]
    with pkg.rng.noise.replay(noise):
        out = tr.evaluate(hps, net, (x, x_len, spec, spec_len, y, y_len, sid))
    assert net.training                                           # back in train mode, like the reference
    assert out["gen/audio"].shape == tuple(m["eval/gen_audio"].shape)
    assert rel_err(out["gen/audio"], m["eval/gen_audio"]) < TOL
    assert np.array_equal(out["attn"].cpu().numpy(), m["eval/attn"])
    assert torch.equal(out["gt/audio"], y[0, :, :int(y_len[0])])
    want = mel.mel_spectrogram_torch(torch.from_numpy(m["eval/gen_audio"]).to("cuda:0").float(), 32, 8, 22050, 16, 32, 0.0, None)[0]
    n = want.size(1)
    assert rel_err(out["gen/mel"][:, :n], want) < TOL
    assert rel_err(out["gt/mel"], mel.spec_to_mel_torch(spec[:1].float(), 32, 8, 22050, 0.0, None)[0]) < 1e-6


def test_spectrograms_on_device_match_per_file_reference(pkg):
    """data_utils.spectrograms_on_device (GPU branch: the DFT product on the convolution kernel) on a zero-padded batch of items
    of different lengths == the reference's spectrogram_torch per FILE (tests/golden/misc.npz loader/*), frame mask included."""
    import types
    from importlib import import_module
    D = import_module("personalized_text-to-speech_amd.data_utils")
    m, _ = _misc()
    nfft, hop = m["loader/hp"].tolist()
    wav, wav_len, spec, spec_len = (torch.from_numpy(m["loader/collate/" + k]).to("cuda:0") for k in ("wav", "wav_len", "spec", "spec_len"))
    hps = types.SimpleNamespace(data=types.SimpleNamespace(filter_length=nfft, hop_length=hop, win_length=nfft, sampling_rate=22050))
    got, got_len = D.spectrograms_on_device(wav, wav_len, hps)
    assert torch.equal(got_len, spec_len)
    t = spec.size(2)
    assert rel_err(got[:, :, :t], spec) < 1e-4
    for i, n in enumerate(spec_len.tolist()):                      # zero beyond each item's own frame count
        assert float(got[i, :, n:].abs().max()) == 0.0 if n < got.size(2) else True


def test_duration_predictor_matches_reference(pkg):
    """models.DurationPredictor (reference models.py:98-132, the use_sdp=False branch) on the convolution / LayerNorm kernels:
    output and every parameter gradient against the reference's own module (fp32)."""
    from importlib import import_module
    M = import_module("personalized_text-to-speech_amd.models")
    m, _ = _misc()
    cin, cf, k, gin = m["durpred/cfg"].tolist()
    dp = M.DurationPredictor(cin, cf, k, 0.0, gin_channels=gin)
    dp.load_state_dict({n[11:]: torch.from_numpy(m[n]) for n in m.files if n.startswith("durpred/sd/")})
    dp = dp.to("cuda:0").train()
    t = lambda n: torch.from_numpy(m["durpred/" + n]).to("cuda:0")
    y = dp(t("x"), t("x_mask"), g=t("g"))
    assert y.shape == tuple(m["durpred/y"].shape) and rel_err(y, m["durpred/y"]) < TOL
    (y * t("w")).sum().backward()
    for n, p in dp.named_parameters():
        assert p.grad is not None, n
        assert rel_err(p.grad, m["durpred/grad/" + n]) < TOL, n


def test_side_branch_refuses_nesting(pkg):
    """kernels.SideBranch: a branch opened inside another open branch (or from a side stream) raises instead of building the
    fork/join pattern that crashed hipStreamEndCapture; sequential branches on different lanes stay allowed."""
    K = pkg.kernels
    x = torch.ones(4, device="cuda:0")
    a = K.SideBranch(x.device, x, lane=0)
    with a:
        y = x * 2
        with pytest.raises(RuntimeError, match="nested"):
            K.SideBranch(x.device, x, lane=1).__enter__()
    b = K.SideBranch(x.device, x, lane=1)
    with b:
        z = x * 3
    y, z = a.join(y), b.join(z)
    torch.cuda.synchronize()
    assert float((y + z).sum()) == 20.0


def test_side_stream_branches_do_not_change_the_step(pkg):
    """The fine-tune step with its side-stream branches (text encoder; alignment + duration predictor + prior expansion on the
    same stream, next to the decoder) against the same step with everything on one stream, from the same parameters, optimizer
    state and generator state: the branches only move launches between streams — every reported scalar and every parameter
    after the update must be identical."""
    from importlib import import_module
    cfgs = import_module("personalized_text-to-speech_amd.configs")
    tr = import_module("personalized_text-to-speech_amd.train")
    hps = cfgs.get("modified_finetune_speaker")
    ft = tr.FineTuner(hps, "cuda:0", amp=True)
    batch = tr.synthetic_batch(hps, 3, (90, 140), "cuda:0")
    ft.step(batch)                                     # (optimizer state, workspaces)
    ts = ft._state_tensors()
    snap = [t.detach().clone() for t in ts]
    rng = torch.cuda.get_rng_state(ft.device)
    results = {}
    for name, branches in (("all", frozenset(("enc_p", "dp", "prior"))), ("two", frozenset(("enc_p", "dp"))), ("none", frozenset())):
        with torch.no_grad():
            for t, s in zip(ts, snap):
                t.copy_(s)
        torch.cuda.set_rng_state(rng, ft.device)
        ft.side_branches = branches
        out = ft.step(batch)
        torch.cuda.synchronize()
        results[name] = ({k: float(v) for k, v in out.items()}, ft.optim_g.flat_p.clone(), ft.optim_d.flat_p.clone())
    ref = results["none"]
    for name in ("all", "two"):
        got = results[name]
        assert got[0] == ref[0], (name, got[0], ref[0])
        assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), name
