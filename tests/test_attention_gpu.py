"""GPU: relative-position attention on the HIP kernels — the softmax row kernels against the torch statement
(tests/cl_emul.py), and MultiHeadAttention against the fixtures produced by the reference (T = 3, 5, 9, 50:
covers lengths <= window + 1, attentions.py:202-210)."""
import importlib
import os

import numpy as np
import pytest
import torch

import cl_emul
from conftest import ROOT
from model_util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("t", [5, 64, 201])
def test_softmax_row_kernels(pkg, dtype, tol, t):
    A = importlib.import_module("personalized_text-to-speech_amd.attention_cl")
    torch.manual_seed(t)
    b, ld, w = 3, (t + 7) // 8 * 8, 4
    s = (torch.randn(b, t, ld, device=DEV) * 4).to(dtype)
    r = torch.randn(b, t, 16, device=DEV).to(dtype)
    keep = ((torch.rand(b, t, ld, device=DEV) > 0.2).float() / 0.8).to(dtype)
    lens = torch.tensor([t, max(1, t - 2), max(1, t // 2)], device=DEV, dtype=torch.int32)
    got = A.relsoftmax(s, r, keep, lens, w, 0.1, True)
    want = cl_emul.relsoftmax(s.cpu(), r.cpu(), keep.cpu(), lens.cpu(), w, 0.1, True)
    for a, bb in zip(got, want):
        assert rel_err(a, bb) < tol
    dpd = torch.randn(b, t, ld, device=DEV).to(dtype)
    dpb = torch.randn(b, t, 16, device=DEV).to(dtype)
    got = A.relsoftmax_bwd(got[0], dpd, dpb, keep, lens, w, 0.1)
    want = cl_emul.relsoftmax_bwd(want[0], dpd.cpu(), dpb.cpu(), keep.cpu(), lens.cpu(), w, 0.1)
    for a, bb in zip(got, want):
        assert rel_err(a, bb) < tol * 3


@pytest.mark.parametrize("T", [3, 5, 9, 50])
def test_mha_against_reference_fixture(pkg, T):
    ops = np.load(os.path.join(ROOT, "tests", "golden", "ops.npz"))
    tag = f"mha{T}/"
    mha = pkg.attentions.MultiHeadAttention(16, 16, 2, p_dropout=0.0, window_size=4)
    mha.load_state_dict({k[len(tag) + 3:]: torch.from_numpy(ops[k]) for k in ops.files if k.startswith(tag + "sd/")})
    mha = mha.to(DEV).eval()
    x = torch.from_numpy(ops[tag + "x"]).to(DEV)
    lens = torch.from_numpy(ops[tag + "lens"]).to(DEV).to(torch.int32)
    y = mha.forward_cl(x.transpose(1, 2).contiguous(), None, lens).transpose(1, 2)
    valid = (torch.arange(T, device=DEV)[None, :] < lens[:, None])
    # rows beyond an item's length are don't-cares downstream (zeroed by x_mask): compare the valid ones
    assert rel_err(y * valid[:, None, :], torch.from_numpy(ops[tag + "y"]).to(DEV) * valid[:, None, :]) < 2e-5
    p = mha.attn * valid[:, None, :, None]
    assert rel_err(p, torch.from_numpy(ops[tag + "p_attn"]).to(DEV) * valid[:, None, :, None]) < 2e-5
