import json
import os

import numpy as np
import torch

from conftest import ROOT


def load_tiny():
    g = np.load(os.path.join(ROOT, "tests", "golden", "model_tiny.npz"))
    cfg = json.loads(bytes(g["config"]).decode())
    return g, cfg


def build_tiny(pkg, g, cfg, device="cpu"):
    net = pkg.SynthesizerTrn(cfg["n_vocab"], cfg["spec_channels"], cfg["segment_size"], n_speakers=cfg["n_speakers"], **cfg["model"])
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    net.load_state_dict(sd, strict=True)
    return net.to(device).eval()


def noise_list(g, prefix):
    n = int(g[prefix + "/n_noise"]) if prefix + "/n_noise" in g.files else 1
    return [torch.from_numpy(g[f"{prefix}/noise{i}"]) for i in range(n)]


def inputs(g, device="cpu"):
    t = lambda k: torch.from_numpy(g["in/" + k]).to(device)
    return t("x"), t("x_lengths"), t("spec"), t("spec_lengths"), t("sid")


def oracle_maximum_path(neg_cent, mask):
    """TEST-ONLY stand-in so the model graph can be exercised without a GPU."""
    from oracle import mas as omas
    t_ys = mask.sum(1)[:, 0].cpu().numpy().astype(np.int32)
    t_xs = mask.sum(2)[:, 0].cpu().numpy().astype(np.int32)
    p = omas.mas_port(neg_cent.detach().cpu().numpy(), t_ys, t_xs)
    return torch.from_numpy(p).to(device=neg_cent.device, dtype=neg_cent.dtype)


def rel_err(a, b):
    if isinstance(a, torch.Tensor):
        a = a.detach().float().cpu().numpy()
    if isinstance(b, torch.Tensor):
        b = b.detach().float().cpu().numpy()
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))
