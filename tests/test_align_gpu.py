"""csrc/align.hip through the C ABI: vits_neg_cent against the oracle's restatement of models.py:470-477 and against the
reference's own `neg_cent` of the whole-model fixture; vits_slice_segments (+ gradient) and vits_generate_path against the
reference fixtures (tests/golden/ops.npz, produced by the reference's commons.py) and against the host implementation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle_neg_cent(z_p, m_p, logs_p):
    import oracle.vits_torch as O
    return O.neg_cent(z_p.double().cpu(), m_p.double().cpu(), logs_p.double().cpu())


@pytest.mark.parametrize("dtypes", [(torch.float32, torch.float32), (torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16)])
def test_neg_cent_matches_oracle(pkg, dtypes):
    torch.manual_seed(3)
    for (b, t_t, t_s, c) in [(2, 37, 11, 192), (3, 130, 70, 64), (16, 500, 201, 192), (1, 64, 64, 32), (2, 45, 9, 24), (2, 70, 33, 100)]:
        # channels-last storage viewed in the reference's [b, c, t] layout; the statistics are the two halves of one tensor
        z = (torch.randn(b, t_t, c, device=DEV) * 1.5).to(dtypes[0])
        stats = torch.randn(b, t_s, 2 * c, device=DEV)
        stats[..., c:] *= 0.3
        stats = stats.to(dtypes[1])
        z_p, m_p, logs_p = z.transpose(1, 2), stats[..., :c].transpose(1, 2), stats[..., c:].transpose(1, 2)
        got = pkg.kernels.neg_cent(z_p, m_p, logs_p)
        want = _oracle_neg_cent(z_p.float(), m_p.float(), logs_p.float())            # the kernel sees the same rounded inputs
        assert got.dtype == torch.float32 and got.shape == (b, t_t, t_s)
        err = float((got.double().cpu() - want).abs().max()) / float(want.abs().max())
        assert err < 5e-6, (b, t_t, t_s, c, err)
        # contiguous [b, c, t] inputs (the reference's own layout) go through a transposing copy and give the same numbers
        again = pkg.kernels.neg_cent(z_p.contiguous(), m_p.contiguous(), logs_p.contiguous())
        assert torch.equal(again, got)


def test_neg_cent_of_the_model_matches_reference_fixture(pkg):
    """The reference's own neg_cent of the tiny whole-model fixture (fp32): recomputed from the fixture's z_p and the text
    encoder's statistics, it must give the reference's alignment bit for bit and its values to 1e-5."""
    from model_util import build_tiny, inputs, load_tiny
    g, cfg = load_tiny()
    net = build_tiny(pkg, g, cfg, DEV)
    x, xl, spec, sl, sid = inputs(g, DEV)
    with torch.no_grad():
        _, m_p, logs_p, x_mask = net.enc_p(x, xl)
        nc = net.neg_cent(torch.from_numpy(g["fwd/z_p"]).to(DEV), m_p, logs_p)
    want = torch.from_numpy(g["fwd/neg_cent"]).to(DEV)
    assert float((nc - want).abs().max()) <= 1e-5 * float(want.abs().max())
    y_mask = torch.from_numpy(g["fwd/y_mask"]).to(DEV)
    mask = (x_mask.unsqueeze(2) * y_mask.unsqueeze(-1)).squeeze(1)
    assert torch.equal(pkg.kernels.maximum_path(nc, mask), torch.from_numpy(g["fwd/attn"]).to(DEV).squeeze(1))


def test_slice_segments_and_generate_path_match_reference_fixtures(pkg, golden_dir):
    import os
    ops = np.load(os.path.join(golden_dir, "ops.npz"))
    c = pkg.commons
    x, ids = torch.from_numpy(ops["slice/x"]).to(DEV), torch.from_numpy(ops["slice/ids"]).to(DEV)
    assert np.array_equal(c.slice_segments(x, ids, 5).cpu().numpy(), ops["slice/y"])
    xt = x.transpose(1, 2).contiguous().transpose(1, 2)                       # channels-last storage, same values
    assert np.array_equal(c.slice_segments(xt, ids, 5).cpu().numpy(), ops["slice/y"])
    path = c.generate_path(torch.from_numpy(ops["genpath/dur"]).to(DEV), torch.from_numpy(ops["genpath/mask"]).to(DEV))
    assert np.array_equal(path.cpu().numpy(), ops["genpath/path"])


def test_slice_segments_layouts_dtypes_and_gradient(pkg):
    c = pkg.commons
    torch.manual_seed(5)
    for (b, d, t, seg, scale, dtype) in [(4, 192, 300, 32, 1, torch.float32), (16, 192, 500, 32, 1, torch.bfloat16), (3, 1, 8192, 512, 256, torch.float32),
                                         (2, 80, 77, 9, 1, torch.float32), (5, 7, 1000, 1000, 1, torch.bfloat16)]:
        t_full = t
        ids = torch.randint(0, (t - seg) // scale + 1, (b,), device=DEV)
        for cl in (False, True):
            x = torch.randn(b, d, t_full, device=DEV).to(dtype)
            if cl:
                x = x.transpose(1, 2).contiguous().transpose(1, 2)
            x.requires_grad_(True)
            y = c.slice_segments(x, ids, seg, ids_scale=scale)
            want = c.slice_segments(x.detach().cpu().float(), ids.cpu(), seg, ids_scale=scale)          # host gather
            assert y.shape == (b, d, seg) and torch.equal(y.detach().cpu().float(), want), (b, d, t, seg, cl)
            w = torch.randn_like(y)
            (y * w).sum().backward()
            xr = x.detach().cpu().float().requires_grad_(True)
            (c.slice_segments(xr, ids.cpu(), seg, ids_scale=scale) * w.cpu().float()).sum().backward()
            assert torch.equal(x.grad.cpu().float(), xr.grad), (b, d, t, seg, cl)


def test_generate_path_sizes(pkg):
    c = pkg.commons
    torch.manual_seed(6)
    for (b, t_x, per) in [(1, 1, 3), (4, 50, 4), (32, 513, 3), (2, 700, 2)]:
        dur = torch.randint(0, per + 1, (b, 1, t_x), device=DEV).float()
        x_len = torch.randint(1, t_x + 1, (b,), device=DEV)
        x_mask = (torch.arange(t_x, device=DEV)[None] < x_len[:, None]).float().unsqueeze(1)
        dur = dur * x_mask
        y_len = dur.sum((1, 2)).clamp_min(1).long()
        t_y = int(y_len.max())
        y_mask = (torch.arange(t_y, device=DEV)[None] < y_len[:, None]).float().unsqueeze(1)
        mask = x_mask.unsqueeze(2) * y_mask.unsqueeze(-1)
        got = c.generate_path(dur, mask)
        want = c.generate_path(dur.cpu(), mask.cpu())                                                  # host implementation (reference formula)
        assert got.shape == mask.shape and torch.equal(got.cpu(), want), (b, t_x)
