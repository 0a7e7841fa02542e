import os

import numpy as np

from conftest import ROOT


def load_cases():
    g = np.load(os.path.join(ROOT, "tests", "golden", "mas_cases.npz"))
    names = sorted({k.split("/")[0] for k in g.files})
    return {n: (g[n + "/neg_cent"], g[n + "/t_ys"], g[n + "/t_xs"], g[n + "/idx"]) for n in names}


def path_from_idx(idx, t_s):
    b, t_t = idx.shape
    p = np.zeros((b, t_t, t_s), np.int32)
    bb, yy = np.nonzero(idx >= 0)
    p[bb, yy, idx[bb, yy]] = 1
    return p


def random_case(rng, b, t_t, t_s, kind):
    t_ys = rng.integers(max(1, t_t // 3), t_t + 1, b)
    t_ys[0] = t_t
    t_xs = np.array([rng.integers(1, min(t_s, t) + 1) for t in t_ys])
    t_xs[0] = min(t_s, t_ys[0])
    if kind == "normal":
        nc = rng.standard_normal((b, t_t, t_s)) * 30 - 200
    elif kind == "ties":
        nc = rng.integers(-2, 3, (b, t_t, t_s))
    else:
        nc = np.zeros((b, t_t, t_s))
    return nc.astype(np.float32), t_ys.astype(np.int32), t_xs.astype(np.int32)
