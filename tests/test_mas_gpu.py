"""GPU: vits_mas_f32 (through the C ABI) must be bit-identical to the reference's alignment."""
import numpy as np
import pytest
import torch

from mas_util import load_cases, path_from_idx, random_case
from oracle import mas as omas

pytestmark = pytest.mark.gpu
CASES = load_cases()


def run_gpu(pkg, nc, t_ys, t_xs, dtype=torch.int32):
    dev = torch.device("cuda:0")
    out = pkg.monotonic_align.maximum_path_lengths(
        torch.from_numpy(nc).to(dev), torch.from_numpy(t_ys).to(dev), torch.from_numpy(t_xs).to(dev), out_dtype=dtype)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("name", sorted(CASES))
def test_golden(pkg, name):
    nc, t_ys, t_xs, idx = CASES[name]
    assert np.array_equal(run_gpu(pkg, nc, t_ys, t_xs), path_from_idx(idx, nc.shape[2]))


@pytest.mark.parametrize("shape", [(2, 1, 1), (3, 17, 17), (2, 63, 64), (2, 130, 65), (4, 257, 128), (3, 400, 130),
                                   (2, 333, 256), (2, 500, 300), (1, 700, 513), (2, 1000, 381)])
@pytest.mark.parametrize("kind", ["normal", "ties", "zeros"])
def test_vs_oracle(pkg, shape, kind):
    rng = np.random.default_rng(hash((shape, kind)) % (2 ** 31))
    nc, t_ys, t_xs = random_case(rng, *shape, kind)
    assert np.array_equal(run_gpu(pkg, nc, t_ys, t_xs), omas.mas_port(nc, t_ys, t_xs))


def test_reference_wrapper_surface(pkg):
    """maximum_path(neg_cent, mask) as called from models.py:479-480: float path, mask-derived lengths."""
    rng = np.random.default_rng(5)
    nc, t_ys, t_xs = random_case(rng, 4, 200, 60, "normal")
    dev = torch.device("cuda:0")
    y_mask = (torch.arange(200)[None, :] < torch.from_numpy(t_ys)[:, None]).float()
    x_mask = (torch.arange(60)[None, :] < torch.from_numpy(t_xs)[:, None]).float()
    mask = (y_mask[:, :, None] * x_mask[:, None, :]).to(dev)
    out = pkg.monotonic_align.maximum_path(torch.from_numpy(nc).to(dev), mask)
    assert out.dtype == torch.float32 and out.device.type == "cuda"
    assert np.array_equal(out.cpu().numpy().astype(np.int32), omas.mas_port(nc, t_ys, t_xs))


def test_out_of_domain_items_are_zero_and_flagged(pkg):
    rng = np.random.default_rng(9)
    nc = rng.standard_normal((3, 20, 30)).astype(np.float32)
    t_ys = np.array([20, 5, 10], np.int32)
    t_xs = np.array([10, 9, 0], np.int32)        # item 1: t_x > t_y, item 2: t_x < 1
    dev = torch.device("cuda:0")
    status = torch.full((3,), -7, dtype=torch.int32, device=dev)
    out = pkg.monotonic_align.maximum_path_lengths(torch.from_numpy(nc).to(dev), torch.from_numpy(t_ys).to(dev),
                                                   torch.from_numpy(t_xs).to(dev), out_dtype=torch.int32, status=status)
    out = out.cpu().numpy()
    assert status.cpu().tolist() == [0, 1, 1]
    assert out[1].sum() == 0 and out[2].sum() == 0
    assert np.array_equal(out[:1], omas.mas_port(nc[:1], t_ys[:1], t_xs[:1]))


def test_full_size_properties(pkg):
    """BASELINE config C3 size (b=64, 800x321): structural properties + sampled items vs the oracle."""
    rng = np.random.default_rng(11)
    b, t_t, t_s = 64, 800, 321
    t_ys = np.linspace(300, 800, b).round().astype(np.int32)[::-1].copy()
    t_xs = (2 * np.round(t_ys / 5) + 1).astype(np.int32)
    nc = (rng.standard_normal((b, t_t, t_s)) * 40 - 300).astype(np.float32)
    p = run_gpu(pkg, nc, t_ys, t_xs)
    rows = p.sum(2)
    for i in range(b):
        assert (rows[i, : t_ys[i]] == 1).all() and (rows[i, t_ys[i]:] == 0).all()
        cols = p[i, : t_ys[i]].argmax(1)
        assert cols[0] == 0 and cols[-1] == t_xs[i] - 1 and set(np.diff(cols)) <= {0, 1}
    sel = [0, 13, 31, 63]
    assert np.array_equal(p[sel], omas.mas_port(nc[sel], t_ys[sel], t_xs[sel]))
