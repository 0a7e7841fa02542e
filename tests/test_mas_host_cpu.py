"""CPU: vits_mas_f32_cpu, the host twin of the alignment entry point (include/vitsmi.h), through the C ABI
against the golden vectors of the reference's own Cython routine (monotonic_align/core.pyx:5-42, tools/gen_golden_mas.py) and
against the oracle on random / tie / all-zero / out-of-domain items.  The package's operators never call this entry point."""
import ctypes

import numpy as np
import pytest

from mas_util import load_cases, path_from_idx, random_case
from oracle import mas as omas

CASES = load_cases()


def host_mas(pkg, nc, t_ys, t_xs, dtype=np.int32):
    nc = np.ascontiguousarray(nc, np.float32)
    b, t_t, t_s = nc.shape
    path = np.full((b, t_t, t_s), 7, dtype)                         # (must be fully overwritten)
    status = np.full(b, -1, np.int32)
    t_ys, t_xs = np.ascontiguousarray(t_ys, np.int32), np.ascontiguousarray(t_xs, np.int32)
    keep = nc.copy()
    rc = pkg._lib.lib().vits_mas_f32_cpu(nc.ctypes.data, path.ctypes.data, 1 if dtype == np.int32 else 0, t_ys.ctypes.data,
                                         t_xs.ctypes.data, b, t_t, t_s, status.ctypes.data)
    assert rc == 0 and np.array_equal(nc, keep)                     # neg_cent is read-only
    return path, status


@pytest.mark.parametrize("name", sorted(CASES))
def test_host_twin_matches_reference_golden(pkg, name):
    nc, t_ys, t_xs, idx = CASES[name]
    got, status = host_mas(pkg, nc, t_ys, t_xs)
    assert (status == 0).all() and np.array_equal(got, path_from_idx(idx, nc.shape[2]))


def test_host_twin_matches_oracle_on_random_items(pkg):
    rng = np.random.default_rng(11)
    for i in range(40):
        nc, t_ys, t_xs = random_case(rng, 3, int(rng.integers(1, 140)), int(rng.integers(1, 70)), ["normal", "ties", "zeros"][i % 3])
        t_xs = np.minimum(t_xs, t_ys)
        got, status = host_mas(pkg, nc, t_ys, t_xs, np.float32 if i % 2 else np.int32)
        assert (status == 0).all() and np.array_equal(got.astype(np.int32), omas.mas_port(nc, t_ys, t_xs))


def test_host_twin_out_of_domain_items(pkg):
    rng = np.random.default_rng(2)
    nc = rng.standard_normal((3, 20, 9)).astype(np.float32)
    got, status = host_mas(pkg, nc, np.array([20, 5, 20]), np.array([9, 7, 0]))          # item 1: t_x > t_y, item 2: t_x = 0
    assert status.tolist() == [0, 1, 1] and not got[1].any() and not got[2].any() and got[0].sum() == 20
