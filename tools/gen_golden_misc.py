#!/usr/bin/env python3
"""Generate tests/golden/misc.npz and tests/golden/ref_G_tiny.pth by RUNNING THE REFERENCE (/root/reference, imported
read-only, CPU fp32) — build container only.  Data only, no reference source:

  ckpt/*      a checkpoint written by the reference's own utils.save_checkpoint (tiny SynthesizerTrn of model_tiny.npz after one
              torch.optim.AdamW step) -> tests/golden/ref_G_tiny.pth; and the REVERSE check done here: a file written by the
              product's utils.save_checkpoint is read back by the reference's utils.load_checkpoint into the reference's model
              (asserted; the result is recorded as ckpt/reverse_ok)
  eval/*      the tensors evaluate() hands to its writer (finetune_speaker_v2.py:313-357) up to the mel / plotting calls: infer() on
              the first item with max_len=1000 and default noise scales (noise captured), cut to y_hat_lengths
  durpred/*   models.DurationPredictor (models.py:98-132): state, inputs, output, gradients of every parameter
  loader/*    data_utils.TextAudioSpeakerLoader (data_utils.py:16-112) over a synthetic file list with `torchaudio.load`
              replaced by a reader of raw float32 files (torchaudio is not installed; only its `load` is used there): which
              entries survive _filter, the shuffled order, the bucketing lengths, and three complete items (text ids, spectrogram
              per file with its own reflect padding, waveform, speaker id); collated by TextAudioSpeakerCollate
"""
import json, os, sys, tempfile, types
import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.dont_write_bytecode = True
from oracle import mas as omas  # noqa: E402

# ---- modules the reference imports that are absent here and unused on these paths
ta = types.ModuleType("torchaudio")


def _load_raw(filename, frame_offset=0, num_frames=-1, normalize=True, channels_first=True):
    a = np.fromfile(filename, dtype=np.int16).astype(np.float32) / 32768.0      # 16-bit PCM payload without a header
    return torch.from_numpy(a).unsqueeze(0), 22050


ta.load = _load_raw
sys.modules["torchaudio"] = ta
for name in ("librosa", "librosa.util", "librosa.filters"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["librosa.util"].normalize = sys.modules["librosa.util"].pad_center = sys.modules["librosa.util"].tiny = None
sys.modules["librosa.filters"].mel = None
sys.modules["librosa"].util, sys.modules["librosa"].filters = sys.modules["librosa.util"], sys.modules["librosa.filters"]
sys.modules["text.cleaners"] = types.ModuleType("text.cleaners")       # (phonemizer back ends; cleaned_text=True never calls them)
m = types.ModuleType("monotonic_align")
m.maximum_path = lambda neg_cent, mask: torch.from_numpy(omas.mas_reference(
    neg_cent.data.cpu().numpy().astype(np.float32), mask.sum(1)[:, 0].data.cpu().numpy().astype(np.int32),
    mask.sum(2)[:, 0].data.cpu().numpy().astype(np.int32))).to(neg_cent)
sys.modules["monotonic_align"] = m
sys.path.insert(0, "/root/reference")
import models, utils as ref_utils, data_utils, text as ref_text  # noqa: E402,E401

np_ = lambda t: t.detach().cpu().numpy()
out = {}
G = os.path.join(ROOT, "tests", "golden")

# ------------------------------------------------------------------ checkpoint written by the reference
g = np.load(os.path.join(G, "model_tiny.npz"))
cfg = json.loads(bytes(g["config"]).decode())
net = models.SynthesizerTrn(cfg["n_vocab"], cfg["spec_channels"], cfg["segment_size"], n_speakers=cfg["n_speakers"], **cfg["model"])
net.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")})
opt = torch.optim.AdamW(net.parameters(), 2e-4, betas=(0.8, 0.99), eps=1e-9)
torch.manual_seed(11)
t = lambda k: torch.from_numpy(g["in/" + k])
o, l_length, *_ = net(t("x"), t("x_lengths"), t("spec"), t("spec_lengths"), t("sid"))
(o.pow(2).mean() + l_length.sum()).backward()
opt.step()
path = os.path.join(G, "ref_G_tiny.pth")
ref_utils.save_checkpoint(net, opt, 1.5e-4, 42, path)
out["ckpt/iteration"] = np.array(42); out["ckpt/learning_rate"] = np.array(1.5e-4)
for k in ("emb_g.weight", "dec.ups.0.weight_v", "dec.ups.0.weight_g", "enc_p.emb.weight", "flow.flows.0.enc.in_layers.0.weight_v"):
    out["ckpt/sd/" + k] = np_(net.state_dict()[k])
sd_opt = opt.state_dict()
out["ckpt/opt_n_state"] = np.array(len(sd_opt["state"]))
out["ckpt/opt_step"] = np.array(float(next(iter(sd_opt["state"].values()))["step"]))
first = min(sd_opt["state"])
out["ckpt/opt_first_exp_avg"] = np_(sd_opt["state"][first]["exp_avg"]); out["ckpt/opt_first_id"] = np.array(first)
# reverse direction: product-written file -> the reference's loader and model
import importlib
P = importlib.import_module("personalized_text-to-speech_amd")
pnet = P.SynthesizerTrn(cfg["n_vocab"], cfg["spec_channels"], cfg["segment_size"], n_speakers=cfg["n_speakers"], **cfg["model"])
pnet.load_state_dict(net.state_dict())
with tempfile.TemporaryDirectory() as d:
    p2 = os.path.join(d, "G_9.pth")
    P.utils.save_checkpoint(pnet, torch.optim.AdamW(pnet.parameters(), 2e-4), 2e-4, 9, p2)
    fresh = models.SynthesizerTrn(cfg["n_vocab"], cfg["spec_channels"], cfg["segment_size"], n_speakers=cfg["n_speakers"] + 1, **cfg["model"])
    _, _, lr, it = ref_utils.load_checkpoint(p2, fresh, None)
    assert (lr, it) == (2e-4, 9)
    for k, v in net.state_dict().items():
        w = fresh.state_dict()[k]
        assert torch.equal(w[:v.shape[0]] if k == "emb_g.weight" else w, v), k
out["ckpt/reverse_ok"] = np.array(1)

# ------------------------------------------------------------------ evaluate(): what finetune_speaker_v2.py:313-332 computes before the plots
from gen_golden_model import NoiseTap  # noqa: E402
net.eval()
with torch.no_grad(), NoiseTap() as tap:
    y_hat, attn, mask, *_ = net.infer(t("x")[:1], t("x_lengths")[:1], t("sid")[:1], max_len=1000)
    y_hat_lengths = mask.sum([1, 2]).long() * 16                   # hop_length of the tiny config = prod(upsample_rates)
for i, dr in enumerate(tap.draws):
    out[f"eval/noise{i}"] = np_(dr)
out["eval/n_noise"] = np.array(len(tap.draws))
out["eval/gen_audio"] = np_(y_hat[0, :, :int(y_hat_lengths[0])]); out["eval/attn"] = np_(attn[0, 0]); out["eval/y_hat_lengths"] = np_(y_hat_lengths)
net.train()

# ------------------------------------------------------------------ DurationPredictor
torch.manual_seed(5)
dp = models.DurationPredictor(16, 32, 3, 0.0, gin_channels=8)
with torch.no_grad():
    for p in dp.parameters():
        p.add_(torch.randn_like(p) * 0.1)
x = torch.randn(2, 16, 13); lens = torch.tensor([13, 9])
x_mask = (torch.arange(13)[None, :] < lens[:, None]).float().unsqueeze(1)
gcond = torch.randn(2, 8, 1)
y = dp(x, x_mask, g=gcond)
wgt = torch.randn_like(y)
(y * wgt).sum().backward()
for k, v in dp.state_dict().items():
    out["durpred/sd/" + k] = np_(v)
for k, p in dp.named_parameters():
    out["durpred/grad/" + k] = np_(p.grad)
out.update({"durpred/x": np_(x), "durpred/x_mask": np_(x_mask), "durpred/g": np_(gcond), "durpred/y": np_(y), "durpred/w": np_(wgt)})
out["durpred/cfg"] = np.array([16, 32, 3, 8])

# ------------------------------------------------------------------ TextAudioSpeakerLoader
symbols = ["_", ",", ".", "!", "?", " "] + list("abcdefghijklmnopqrstuvwxyz")
rng = np.random.default_rng(9)
hop, nfft = 16, 64
with tempfile.TemporaryDirectory() as d:
    lines, texts = [], []
    for i in range(12):
        n = int(rng.integers(40, 400)) * 2 + (i % 2)              # odd sample counts too
        tt = np.arange(n) / 22050.0
        w = 0.4 * np.sin(2 * np.pi * (200 + 37 * i) * tt) + 0.05 * rng.standard_normal(n)
        fn = os.path.join(d, f"u{i}.raw")
        (np.clip(w, -1, 1) * 32767).astype(np.int16).tofile(fn)
        nch = [0, 1, 5, 30, 189, 190, 191, 250, 12, 77, 3, 64][i]
        txt = "".join(rng.choice(list("abc xyz,.Q#"), size=nch))    # Q and # are not symbols: dropped by the id mapping
        lines.append(f"{fn}|{i % 4}|{txt}"); texts.append(txt)
    lst = os.path.join(d, "list.txt")
    open(lst, "w", encoding="utf-8").write("\n".join(lines) + "\n")
    hp = types.SimpleNamespace(text_cleaners=["none"], max_wav_value=32768.0, sampling_rate=22050, filter_length=nfft, hop_length=hop,
                               win_length=nfft, cleaned_text=True, add_blank=True)
    ds = data_utils.TextAudioSpeakerLoader(lst, hp, symbols)
    kept = [int(os.path.basename(a[0])[1:-4]) for a in ds.audiopaths_sid_text]
    out["loader/kept"] = np.array(kept); out["loader/lengths"] = np.array(ds.lengths)
    out["loader/texts"] = np.frombuffer(json.dumps(texts).encode(), dtype=np.uint8)
    out["loader/symbols"] = np.frombuffer(json.dumps(symbols).encode(), dtype=np.uint8)
    out["loader/sizes"] = np.array([os.path.getsize(os.path.join(d, f"u{i}.raw")) for i in range(12)])
    out["loader/hp"] = np.array([nfft, hop])
    items = []
    for j in range(3):
        txt_ids, spec, wav, sid = ds[j]
        items.append((txt_ids, spec, wav, sid))
        out[f"loader/item{j}/text"] = np_(txt_ids); out[f"loader/item{j}/spec"] = np_(spec)
        out[f"loader/item{j}/wav"] = np_(wav); out[f"loader/item{j}/sid"] = np_(sid)
        out[f"loader/item{j}/pcm"] = np.fromfile(ds.audiopaths_sid_text[j][0], dtype=np.int16)
    res = data_utils.TextAudioSpeakerCollate()(items)
    for name, tns in zip(["text", "text_len", "spec", "spec_len", "wav", "wav_len", "sid"], res):
        out["loader/collate/" + name] = np_(tns)

np.savez_compressed(os.path.join(G, "misc.npz"), **out)
print("misc.npz", os.path.getsize(os.path.join(G, "misc.npz")), "bytes; ref_G_tiny.pth", os.path.getsize(path), "bytes; kept", kept)
