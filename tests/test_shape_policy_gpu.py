"""GPU: the shape policy that connects the bucketed loader to the captured step (train.ShapePolicy, FineTuner.step_padded;
reference finetune_speaker_v2.py:73-83, data_utils.py:170-250): padding a batch up to its bucket's shape changes nothing — every
layer masks by the lengths and the losses take slices and masked sums — so a padded step equals the unpadded eager step, and the
per-shape capture cache (eager -> capture -> replay) gives the same updates as eager steps."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _setup(pkg):
    import dp_child as C
    from importlib import import_module
    cfgs = import_module("personalized_text-to-speech_amd.configs")
    tr = import_module("personalized_text-to-speech_amd.train")
    g = np.load(os.path.join(ROOT, "tests", "golden", "step_tiny.npz"))
    cfg = json.loads(bytes(g["config"]).decode())
    return C, cfgs, tr, g, cfg


def _noise_for(batch, cfg, pad_tx=None, pad_ty=None):
    """The step's three draws (posterior noise [b, t_y, c], duration noise [b, 2, t_x], slice offsets [b]) for the UNPADDED shapes,
    optionally zero-padded to the padded shapes — the same values at the same (item, position)."""
    gen = torch.Generator().manual_seed(77)
    b, t_x, t_y = batch[0].size(0), batch[0].size(1), batch[2].size(2)
    c = cfg["model"]["inter_channels"]
    n_q = torch.randn(b, c, t_y, generator=gen)
    n_w = torch.randn(b, 2, t_x, generator=gen)
    n_s = torch.rand(b, generator=gen)
    if pad_ty:
        n_q = torch.nn.functional.pad(n_q, (0, pad_ty - t_y))
    if pad_tx:
        n_w = torch.nn.functional.pad(n_w, (0, pad_tx - t_x))
    return n_q, n_w, n_s


def test_padded_step_equals_unpadded_step_fp32(pkg):
    C, cfgs, tr, g, cfg = _setup(pkg)
    policy = tr.ShapePolicy([4, 32, 64], cfg["data"]["hop_length"], t_x_steps=(16, 32))
    batch = C.make_batch(pkg, cfg, 0)                               # t_x = 11, t_y = 24
    tx, ty = policy.padded_shape(batch[0].size(1), batch[2].size(2))
    assert (tx, ty) == (16, 32)
    results = []
    for padded in (False, True):
        ft = C.make_tuner(pkg, cfgs, tr, g, cfg)
        bt = policy.pad(batch) if padded else batch
        orig = (pkg.rng.noise.randn_like, pkg.rng.noise.randn, pkg.rng.noise.rand)
        n_q, n_w, n_s = _noise_for(batch, cfg, tx if padded else None, ty if padded else None)

        def serve(shape, device, dtype):
            shape = tuple(shape)
            for cand in (n_q, n_q.transpose(1, 2), n_w, n_s):
                if tuple(cand.shape) == shape:
                    return cand.to(device=device, dtype=dtype).contiguous()
            raise AssertionError(f"unexpected noise request {shape}")

        pkg.rng.noise.randn_like = lambda x: serve(x.shape, x.device, x.dtype)
        pkg.rng.noise.randn = lambda *shape, device=None, dtype=None: serve(shape, device, dtype)
        pkg.rng.noise.rand = lambda *shape, device=None, dtype=None: serve(shape, device, dtype)
        try:
            out = ft.step(bt)
            torch.cuda.synchronize()
        finally:
            pkg.rng.noise.randn_like, pkg.rng.noise.randn, pkg.rng.noise.rand = orig
        results.append(({k: float(v) for k, v in out.items()}, ft.optim_g.flat_p.detach().clone(), ft.optim_d.flat_p.detach().clone()))
    (la, ga, da), (lb, gb, db) = results
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-5 * max(abs(la[k]), 1e-6), (k, la[k], lb[k])
    # gradients that are mathematically zero leave rounding residue that AdamW's first step turns into +-lr: compare the updates
    # where they are significant (as tests/test_step_parity_gpu.py does)
    lr = float(ft.hps.train.learning_rate)
    for a, b in ((ga, gb), (da, db)):
        assert float((a - b).abs().max()) <= 2.05 * lr and float(((a - b).abs() > 1e-6).float().mean()) < 5e-3


def test_capture_cache_per_shape(pkg):
    C, cfgs, tr, g, cfg = _setup(pkg)
    policy = tr.ShapePolicy([4, 32, 64], cfg["data"]["hop_length"], t_x_steps=(16, 32))
    b0 = C.make_batch(pkg, cfg, 0)
    b1 = C.make_batch(pkg, cfg, 1)
    fa, fb = C.make_tuner(pkg, cfgs, tr, g, cfg), C.make_tuner(pkg, cfgs, tr, g, cfg)
    seq = [b0, b1, b0, b1, b0]                                      # one shape: eager, capture + replay, replay, replay, replay
    for i, bt in enumerate(seq):
        torch.manual_seed(500 + i)
        oa = fa.step_padded(bt, policy)
        torch.manual_seed(500 + i)
        ob = fb.step(policy.pad(bt))
        torch.cuda.synchronize()
        for k in oa:
            assert abs(float(oa[k]) - float(ob[k])) <= 1e-5 * max(abs(float(ob[k])), 1e-6), (i, k)
    assert len(fa._shape_graphs) == 1 and "graph" in next(iter(fa._shape_graphs.values()))
    assert float((fa.optim_g.flat_p - fb.optim_g.flat_p).abs().max()) <= 1e-6
