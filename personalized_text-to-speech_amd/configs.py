"""Hyper-parameters of the reference's configs (configs/finetune_speaker.json:2-52,
configs/modified_finetune_speaker.json, configs/uma_trilingual.json — identical except n_speakers
and file lists) and the synthetic workloads C1..C5 of SURVEY.md §8(d)."""
import copy

MODEL = dict(inter_channels=192, hidden_channels=192, filter_channels=768, n_heads=2, n_layers=6, kernel_size=3,
             p_dropout=0.1, resblock="1", resblock_kernel_sizes=[3, 7, 11],
             resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], upsample_rates=[8, 8, 2, 2],
             upsample_initial_channel=512, upsample_kernel_sizes=[16, 16, 4, 4], n_layers_q=3,
             use_spectral_norm=False, gin_channels=256)
DATA = dict(sampling_rate=22050, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80,
            mel_fmin=0.0, mel_fmax=None, add_blank=True, n_speakers=999)
TRAIN = dict(seed=1234, learning_rate=2e-4, betas=[0.8, 0.99], eps=1e-9, batch_size=16, fp16_run=True,
             lr_decay=0.999875, segment_size=8192, c_mel=45, c_kl=1.0)
N_SYMBOLS = 68          # len(symbols), configs/finetune_speaker.json:53


class HParams(dict):
    """Attribute-style nested dict, like the reference's utils.HParams (utils.py:405-434)."""

    def __init__(self, d):
        super().__init__({k: HParams(v) if isinstance(v, dict) else v for k, v in d.items()})

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def get(name):
    """name in {finetune_speaker, modified_finetune_speaker, uma_trilingual, hires48k}."""
    cfg = dict(train=copy.deepcopy(TRAIN), data=copy.deepcopy(DATA), model=copy.deepcopy(MODEL), n_symbols=N_SYMBOLS)
    if name == "modified_finetune_speaker":
        cfg["data"]["n_speakers"] = 13
    elif name == "hires48k":            # BASELINE.json configs[4]: synthetic 48 kHz variant
        cfg["data"].update(sampling_rate=48000, filter_length=2048, win_length=2048, hop_length=960)
        cfg["train"].update(segment_size=30720, batch_size=8)
        cfg["model"].update(upsample_rates=[10, 8, 4, 3], upsample_kernel_sizes=[20, 16, 8, 9], upsample_initial_channel=1024)
    elif name not in ("finetune_speaker", "uma_trilingual"):
        raise KeyError(name)
    return HParams(cfg)


# name -> (config, per-rank batch, (T_y lo, hi))
WORKLOADS = {
    "C1": ("finetune_speaker", 2, (320, 400)),
    "C2": ("modified_finetune_speaker", 16, (200, 500)),
    "C3": ("uma_trilingual", 64, (300, 800)),
    "C5": ("hires48k", 8, (300, 600)),
}
# C4 (BASELINE.json configs[3]): inference only — batch 32, 256-phoneme prompts (T_x = 513 with blanks), durations forced to
# the pattern 2,2,1 over the tokens so that every item is 861 frames = 220 416 samples (10.0 s at 22.05 kHz)
C4 = dict(config="finetune_speaker", batch=32, t_x=513, frames=861, noise_scale=0.667, noise_scale_w=0.8)


def c4_inputs(hps, device, seed=1234):
    """(x, x_lengths, sid, durations) of workload C4 (seeded)."""
    import torch
    gen = torch.Generator().manual_seed(seed)
    b, t_x, frames = C4["batch"], C4["t_x"], C4["frames"]
    x = torch.zeros(b, t_x, dtype=torch.long)
    x[:, 1::2] = torch.randint(1, hps.n_symbols, (b, t_x // 2), generator=gen)
    dur = torch.tensor([2, 2, 1] * (t_x // 3 + 1))[:t_x].float()
    extra = frames - int(dur.sum())                       # 513 tokens of 2,2,1 give 855 frames: the first tokens take the rest
    dur[:extra] += 1
    assert int(dur.sum()) == frames
    return (x.to(device), torch.full((b,), t_x, dtype=torch.long, device=device), (torch.arange(b) % 10).to(device),
            dur.view(1, 1, t_x).expand(b, 1, t_x).contiguous().to(device))
