"""GPU: csrc/reduce.hip (deterministic two-stage reductions) and the fused feature-matching loss against torch."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_sum12_and_sum_all(pkg):
    R = importlib.import_module("personalized_text-to-speech_amd.reduce")
    torch.manual_seed(0)
    for shape in [(16, 192, 500), (3, 2, 7), (16, 1, 201), (1, 513, 4000)]:
        x = torch.randn(*shape, device=DEV, requires_grad=True)
        a, b = R.sum12(x), torch.sum(x.double(), [1, 2])
        assert a.dtype == torch.float32 and torch.allclose(a.double(), b, rtol=1e-5, atol=1e-3)
        w = torch.randn(shape[0], device=DEV)
        (ga,) = torch.autograd.grad((a * w).sum(), x)
        assert torch.equal(ga, w.view(-1, 1, 1).expand(shape))
        s = R.sum_all(x)
        assert s.dim() == 0 and torch.allclose(s.double(), x.double().sum(), rtol=1e-5, atol=1e-3)
        (gs,) = torch.autograd.grad(s * 3.0, x)
        assert torch.equal(gs, torch.full_like(x, 3.0))
    # bitwise reproducible
    x = torch.randn(16, 192, 500, device=DEV)
    assert torch.equal(R.sum12(x), R.sum12(x))
    # transposed (non-contiguous) input
    xt = torch.randn(4, 100, 30, device=DEV).transpose(1, 2)
    assert torch.allclose(R.sum12(xt), xt.sum([1, 2]), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_feature_l1_matches_reference_formula(pkg, dtype):
    """2 * sum_l mean |real_l - generated_l| (reference losses.py:7-15) with gradient only to the generated half."""
    R = importlib.import_module("personalized_text-to-speech_amd.reduce")
    torch.manual_seed(1)
    hs = [torch.randn(2 * n, t, c, device=DEV).to(dtype).requires_grad_(True) for n, t, c in [(32, 1366, 32), (96, 34, 1024), (16, 51, 8), (5, 3, 1)]]
    dens = [h.numel() // 2 for h in hs]
    dens[2] = hs[2].numel() // 2 // 8                      # a padded tensor: only channel 0 is real, the others are equal (zero) in both halves
    with torch.no_grad():
        hs[2][..., 1:] = 0
    loss = R.feature_l1(hs, dens)
    ref = 0
    for h, d in zip(hs, dens):
        half = h.size(0) // 2
        ref = ref + 2 * (h[:half].detach().float() - h[half:].float()).abs().sum() / d
    assert torch.allclose(loss, ref, rtol=2e-5)
    ga = torch.autograd.grad(loss * 1.5, hs)
    gb = torch.autograd.grad(ref * 1.5, hs)
    for a, b in zip(ga, gb):
        assert a.dtype == b.dtype and torch.allclose(a.float(), b.float(), rtol=1e-2 if dtype == torch.bfloat16 else 1e-6, atol=0)
        assert float(a[: a.size(0) // 2].abs().max()) == 0.0
    assert torch.equal(R.feature_l1(hs, dens), loss)       # reproducible


def test_mpd_feature_loss_uses_fused_path_and_matches(pkg):
    torch.manual_seed(2)
    d = pkg.MultiPeriodDiscriminator(False).to(DEV)
    y, y_hat = torch.rand(2, 1, 4096, device=DEV) * 2 - 1, (torch.rand(2, 1, 4096, device=DEV) * 2 - 1).requires_grad_(True)
    _, _, fr, fg = d(y, y_hat)
    assert fg.cl is not None and fr.cl is fg.cl
    fused = pkg.losses.feature_loss(fr, fg)
    plain = pkg.losses.feature_loss([list(f) for f in fr], [list(f) for f in fg])     # plain lists: generic torch path
    assert torch.allclose(fused, plain, rtol=1e-5)
    (ga,) = torch.autograd.grad(fused, y_hat, retain_graph=True)
    (gb,) = torch.autograd.grad(plain, y_hat)
    assert float((ga - gb).norm() / gb.norm()) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_lrelu_mask_bwd(pkg, dtype):
    import cl_emul
    torch.manual_seed(4)
    dy = torch.randn(5, 37, 24, device=DEV).to(dtype)
    y = torch.randn(5, 37, 24, device=DEV).to(dtype)
    lens = torch.tensor([37, 1, 20, 36, 5], device=DEV, dtype=torch.int32)
    for yy, sl, ll in [(y, 0.1, None), (None, 1.0, lens), (y, 0.2, lens)]:
        a, b = pkg.kernels.lrelu_mask_bwd(dy, yy, sl, ll), cl_emul.lrelu_mask_bwd(dy, yy, sl, ll)
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_colsum(pkg, dtype):
    torch.manual_seed(6)
    for shape in [(16, 500, 384), (16, 8192, 32), (3, 7, 5), (16, 256, 256), (1, 1, 8)]:
        x = torch.randn(*shape, device=DEV).to(dtype)
        a, b = pkg.kernels.colsum(x), x.double().sum((0, 1))
        assert a.dtype == torch.float32 and torch.allclose(a.double(), b, rtol=1e-4, atol=1e-2)
        a, b = pkg.kernels.colsum(x, per_item=True), x.double().sum(1)
        assert a.shape == b.shape and torch.allclose(a.double(), b, rtol=1e-4, atol=1e-2)
        assert torch.equal(pkg.kernels.colsum(x), pkg.kernels.colsum(x))
