"""CPU: the C restatement (oracle/mas_ref.c) against the golden vectors produced by the
reference's own Cython routine (tools/gen_golden_mas.py), plus structural properties."""
import numpy as np
import pytest

from mas_util import load_cases, path_from_idx, random_case
from oracle import mas as omas

CASES = load_cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_port_matches_reference_golden(name):
    nc, t_ys, t_xs, idx = CASES[name]
    got = omas.mas_port(nc, t_ys, t_xs)
    assert np.array_equal(got, path_from_idx(idx, nc.shape[2]))


def test_port_matches_reference_build_if_present():
    if not omas.have_reference():
        pytest.skip("oracle/_ref not built (only possible next to /root/reference)")
    rng = np.random.default_rng(7)
    for i in range(60):
        nc, t_ys, t_xs = random_case(rng, 3, int(rng.integers(1, 120)), int(rng.integers(1, 60)),
                                     ["normal", "ties", "zeros"][i % 3])
        t_xs = np.minimum(t_xs, t_ys)
        assert np.array_equal(omas.mas_port(nc, t_ys, t_xs), omas.mas_reference(nc, t_ys, t_xs))


def test_path_properties():
    rng = np.random.default_rng(3)
    nc, t_ys, t_xs = random_case(rng, 4, 150, 40, "normal")
    p = omas.mas_port(nc, t_ys, t_xs)
    for i in range(4):
        rows = p[i].sum(1)
        assert (rows[: t_ys[i]] == 1).all() and (rows[t_ys[i]:] == 0).all()
        cols = p[i, : t_ys[i]].argmax(1)
        assert cols[0] == 0 and cols[-1] == t_xs[i] - 1
        assert ((np.diff(cols) == 0) | (np.diff(cols) == 1)).all()     # monotone, no skips


def test_domain_error():
    with pytest.raises(ValueError):
        omas.mas_port(np.zeros((1, 3, 5), np.float32), [3], [5])
