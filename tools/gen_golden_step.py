#!/usr/bin/env python3
"""Generate tests/golden/step_tiny.npz by RUNNING THE REFERENCE's modules (/root/reference, imported
read-only, CPU fp32) through one iteration of its fine-tune loop — build container only.

The loop body of finetune_speaker_v2.py:174-232 cannot be imported (the script asserts CUDA at :48 and needs
tensorboard), so this file drives the reference's OWN `models.SynthesizerTrn`, `models.MultiPeriodDiscriminator`,
`mel_processing.spec_to_mel_torch / mel_spectrogram_torch`, `commons.slice_segments / clip_grad_value_`,
`losses.*`, `torch.optim.AdamW` and a disabled GradScaler in that order: G forward, mel targets, D forward on
(y, y_hat.detach()), D backward + AdamW step, D forward again (updated D), generator losses, G backward + AdamW.
fp16_run = False (fp32 parity mode).  Dropout is switched off (modules in eval mode): the stochastic duration
predictor hard-codes p = 0.5 (models.py:452) and a dropout mask cannot be replayed through the product's noise hook.

Stored: data only (tiny-config G state_dict, inputs, the noise tensors drawn, losses, gradient norms, a few
gradients, per-parameter gradient / parameter L2 norms after the update).  The 46.7 M discriminator parameters
are reproduced from the seed, not stored (tests/golden/ops.npz pins that construction).

librosa is not installed: `librosa.filters.mel` is served by the oracle's restatement of librosa 0.9.2's
defaults (mel-basis parity stays UNPINNED); everything else in mel_processing.py is the
reference's own code.
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from oracle import vits_torch as O  # noqa: E402
from tools.gen_golden_model import NoiseTap, TINY, _install_monotonic_align  # noqa: E402


def np_(t):
    """A COPY (numpy would otherwise share memory with parameters the optimizer steps update in place)."""
    return t.detach().cpu().numpy().copy()

DATA = dict(sampling_rate=22050, filter_length=32, hop_length=16, win_length=32, n_mel_channels=8, mel_fmin=0.0, mel_fmax=None)
TRAIN = dict(learning_rate=1e-3, betas=[0.8, 0.99], eps=1e-9, segment_size=128, c_mel=45, c_kl=1.0, seed=1234, lr_decay=0.999875)
D_SEED = 99

G_GRADS = ["enc_p.emb.weight", "enc_p.encoder.attn_layers.0.emb_rel_k", "enc_p.encoder.attn_layers.1.conv_q.weight",
           "enc_p.proj.weight", "enc_q.enc.in_layers.3.weight_v", "enc_q.enc.in_layers.3.weight_g", "enc_q.enc.cond_layer.weight_v",
           "flow.flows.2.post.weight", "flow.flows.0.enc.res_skip_layers.1.weight_v", "dp.flows.1.proj.weight",
           "dp.post_flows.3.convs.convs_sep.1.weight", "dp.flows.0.m", "dec.ups.1.weight_v", "dec.ups.1.weight_g",
           "dec.resblocks.2.convs1.1.weight_v", "dec.conv_post.weight", "dec.conv_pre.bias", "emb_g.weight", "dec.cond.weight",
           "enc_q.pre.weight"]
D_GRADS = ["discriminators.0.convs.0.weight_v", "discriminators.0.convs.1.weight_g", "discriminators.0.conv_post.weight_v",
           "discriminators.1.convs.0.weight_v", "discriminators.1.convs.1.bias", "discriminators.3.convs.0.weight_g",
           "discriminators.5.conv_post.weight_v", "discriminators.5.convs.2.bias"]


def main():
    _install_monotonic_align()
    lib = types.ModuleType("librosa"); lib_util = types.ModuleType("librosa.util"); lib_f = types.ModuleType("librosa.filters")
    lib_util.normalize = lib_util.pad_center = lib_util.tiny = None
    lib_f.mel = lambda sr, n_fft, n_mels, fmin, fmax: O.mel_basis_slaney(sr, n_fft, n_mels, fmin, fmax).numpy()
    lib.util, lib.filters = lib_util, lib_f
    sys.modules.update({"librosa": lib, "librosa.util": lib_util, "librosa.filters": lib_f})
    sys.path.insert(0, "/root/reference")
    import commons, losses, models, mel_processing  # noqa: E401
    from torch.cuda.amp import GradScaler
    from torch.nn import functional as F

    cfg = dict(TINY)
    out = {"config": np.frombuffer(json.dumps(dict(cfg, data=DATA, train=TRAIN, d_seed=D_SEED)).encode(), dtype=np.uint8)}
    torch.manual_seed(1234)
    net_g = models.SynthesizerTrn(cfg["n_vocab"], cfg["spec_channels"], cfg["segment_size"], n_speakers=cfg["n_speakers"], **cfg["model"])
    with torch.no_grad():                  # zero-initialised projections would hide what is behind them
        for p in net_g.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    torch.manual_seed(D_SEED)
    net_d = models.MultiPeriodDiscriminator(False)
    net_g.eval(); net_d.eval()             # = train mode with dropout off (no other mode-dependent layer exists)
    for k, v in net_g.state_dict().items():
        out["sd/" + k] = np_(v)
    out["d/param_checksum"] = np.array([float(p.double().sum()) for p in net_d.parameters()], np.float64)

    B, T_x, T_y, hop = 2, 11, 24, DATA["hop_length"]
    torch.manual_seed(7)
    x_len = torch.tensor([11, 7]); y_len = torch.tensor([24, 17])
    x = torch.randint(1, cfg["n_vocab"], (B, T_x)); x[1, 7:] = 0
    wav = torch.rand(B, 1, T_y * hop) * 1.6 - 0.8
    wav[1, :, 17 * hop:] = 0
    spec = mel_processing.spectrogram_torch(wav.squeeze(1), DATA["filter_length"], DATA["sampling_rate"], hop, DATA["win_length"])
    spec[1, :, 17:] = 0
    sid = torch.tensor([0, 2])
    out.update({"in/x": np_(x), "in/x_lengths": np_(x_len), "in/spec": np_(spec), "in/spec_lengths": np_(y_len),
                "in/y": np_(wav), "in/y_lengths": np_(y_len * hop), "in/sid": np_(sid)})

    kw = dict(betas=TRAIN["betas"], eps=TRAIN["eps"])
    optim_g = torch.optim.AdamW(net_g.parameters(), TRAIN["learning_rate"], **kw)
    optim_d = torch.optim.AdamW(net_d.parameters(), TRAIN["learning_rate"], **kw)
    scaler = GradScaler(enabled=False)
    seg_frames = TRAIN["segment_size"] // hop

    # ---- finetune_speaker_v2.py:180-232
    with NoiseTap() as tap:
        y_hat, l_length, attn, ids_slice, x_mask, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q) = net_g(x, x_len, spec, y_len, sid)
    mel = mel_processing.spec_to_mel_torch(spec, DATA["filter_length"], DATA["n_mel_channels"], DATA["sampling_rate"], DATA["mel_fmin"], DATA["mel_fmax"])
    y_mel = commons.slice_segments(mel, ids_slice, seg_frames)
    y_hat_mel = mel_processing.mel_spectrogram_torch(y_hat.squeeze(1), DATA["filter_length"], DATA["n_mel_channels"], DATA["sampling_rate"],
                                                     hop, DATA["win_length"], DATA["mel_fmin"], DATA["mel_fmax"])
    y = commons.slice_segments(wav, ids_slice * hop, TRAIN["segment_size"])
    y_d_hat_r, y_d_hat_g, _, _ = net_d(y, y_hat.detach())
    loss_disc, _, _ = losses.discriminator_loss(y_d_hat_r, y_d_hat_g)
    optim_d.zero_grad()
    scaler.scale(loss_disc).backward()
    scaler.unscale_(optim_d)
    grad_norm_d = commons.clip_grad_value_(net_d.parameters(), None)
    d_grads = {k: p.grad.detach().clone() for k, p in net_d.named_parameters()}
    with torch.no_grad():                  # what the generator losses would be against the NOT yet updated discriminator
        _, g_stale, fr_s, fg_s = net_d(y, y_hat)
        stale = dict(loss_gen=losses.generator_loss(g_stale)[0], loss_fm=losses.feature_loss(fr_s, fg_s))
    scaler.step(optim_d)

    y_d_hat_r, y_d_hat_g, fmap_r, fmap_g = net_d(y, y_hat)
    loss_dur = torch.sum(l_length.float())
    loss_mel = F.l1_loss(y_mel, y_hat_mel) * TRAIN["c_mel"]
    loss_kl = losses.kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * TRAIN["c_kl"]
    loss_fm = losses.feature_loss(fmap_r, fmap_g)
    loss_gen, _ = losses.generator_loss(y_d_hat_g)
    loss_gen_all = loss_gen + loss_fm + loss_mel + loss_dur + loss_kl
    optim_g.zero_grad()
    scaler.scale(loss_gen_all).backward()
    scaler.unscale_(optim_g)
    grad_norm_g = commons.clip_grad_value_(net_g.parameters(), None)
    g_grads = {k: p.grad.detach().clone() for k, p in net_g.named_parameters()}
    scaler.step(optim_g)
    scaler.update()

    for i, d in enumerate(tap.draws):
        out[f"noise{i}"] = np_(d)
    out["n_noise"] = np.array(len(tap.draws))
    for k, v in dict(loss_disc=loss_disc, loss_gen=loss_gen, loss_fm=loss_fm, loss_mel=loss_mel, loss_dur=loss_dur, loss_kl=loss_kl,
                     grad_norm_d=grad_norm_d, grad_norm_g=grad_norm_g, stale_loss_gen=stale["loss_gen"], stale_loss_fm=stale["loss_fm"]).items():
        out["out/" + k] = np.array(float(v), np.float64)
    out["out/ids_slice"] = np_(ids_slice)
    out["out/y_hat"] = np_(y_hat)
    out["out/y_mel"] = np_(y_mel)
    out["out/y_hat_mel"] = np_(y_hat_mel)
    for k in G_GRADS:
        out["grad_g/" + k] = np_(g_grads[k])
    for k in D_GRADS:
        out["grad_d/" + k] = np_(d_grads[k])
    names_g, names_d = [k for k, _ in net_g.named_parameters()], [k for k, _ in net_d.named_parameters()]
    out["names_g"] = np.frombuffer(json.dumps(names_g).encode(), dtype=np.uint8)
    out["names_d"] = np.frombuffer(json.dumps(names_d).encode(), dtype=np.uint8)
    out["gradnorm_g"] = np.array([float(g_grads[k].double().norm()) for k in names_g])
    out["gradnorm_d"] = np.array([float(d_grads[k].double().norm()) for k in names_d])
    sd0 = {k[3:]: v for k, v in out.items() if k.startswith("sd/")}
    pg = dict(net_g.named_parameters())
    out["upd_g_norm"] = np.array([float((pg[k].detach().double() - torch.from_numpy(sd0[k]).double()).norm()) for k in names_g])
    out["new_g_sum"] = np.array([float(pg[k].detach().double().sum()) for k in names_g])
    out["new_d_sum"] = np.array([float(p.detach().double().sum()) for p in net_d.parameters()])
    for k in G_GRADS:
        out["new_g/" + k] = np_(pg[k])
    pd = dict(net_d.named_parameters())
    for k in D_GRADS:
        out["new_d/" + k] = np_(pd[k])

    path = os.path.join(ROOT, "tests", "golden", "step_tiny.npz")
    np.savez_compressed(path, **out)
    print("step_tiny.npz", os.path.getsize(path), "bytes;",
          {k[4:]: float(v) for k, v in out.items() if k.startswith("out/") and v.ndim == 0})


if __name__ == "__main__":
    main()
