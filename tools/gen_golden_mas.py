#!/usr/bin/env python3
"""Generate tests/golden/mas_cases.npz from the REFERENCE's own Cython DP.

Runs only in the build container (needs oracle/_ref/core*.so, built by `make -C oracle ref`
from /root/reference/monotonic_align/core.pyx).  Stores, per case, the float32 input
`neg_cent`, the int32 lengths and the expected path as one column index per row
(`idx[b, y]` = the single x with path[b,y,x]==1, -1 for rows y >= t_y).

Cases follow SURVEY.md §8(c): tiny, square, ragged, all-tie, integer-tie, +-1e9 magnitudes.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import mas as omas  # noqa: E402


def make_cases():
    rng = np.random.default_rng(1234)
    cases = {}

    def add(name, nc, t_ys, t_xs):
        cases[name] = (np.asarray(nc, np.float32), np.asarray(t_ys, np.int32), np.asarray(t_xs, np.int32))

    add("one_cell", rng.standard_normal((1, 1, 1)), [1], [1])
    add("square5", rng.standard_normal((1, 5, 5)), [5], [5])
    add("ragged_7x3", rng.standard_normal((2, 7, 3)) * 3, [7, 4], [3, 2])
    add("ragged_64x17", rng.standard_normal((3, 64, 17)) * 10, [64, 40, 17], [17, 9, 17])
    add("ragged_200x60", rng.standard_normal((4, 200, 60)) * 50 - 300, [200, 171, 120, 61], [60, 55, 31, 60])
    add("c1_400x101", rng.standard_normal((2, 400, 101)) * 80 - 500, [400, 320], [101, 81])
    add("all_tie", np.zeros((2, 33, 12)), [33, 20], [12, 12])
    add("integer_tie", rng.integers(-3, 4, (3, 50, 21)).astype(np.float32), [50, 49, 21], [21, 20, 21])
    big = rng.standard_normal((2, 40, 13)).astype(np.float32)
    big[0] *= 1e9
    big[1] = big[1] * 1e9 - 2e9
    add("huge_magnitude", big, [40, 30], [13, 13])
    # lanes/chunk boundaries of the GPU kernel: t_s around 64, 128, 256
    add("wide_70x65", rng.standard_normal((2, 70, 65)) * 5, [70, 66], [65, 64])
    add("wide_300x129", rng.standard_normal((2, 300, 129)) * 5, [300, 131], [129, 128])
    add("wide_520x257", rng.standard_normal((1, 520, 257)) * 5, [520], [257])
    return cases


def main():
    out = {}
    for name, (nc, t_ys, t_xs) in make_cases().items():
        path = omas.mas_reference(nc, t_ys, t_xs)
        b, t_t, t_s = nc.shape
        idx = np.full((b, t_t), -1, np.int32)
        for i in range(b):
            rs = path[i].sum(1)
            assert (rs[: t_ys[i]] == 1).all() and (rs[t_ys[i]:] == 0).all(), name
            idx[i, : t_ys[i]] = path[i, : t_ys[i]].argmax(1)
        out[name + "/neg_cent"] = nc
        out[name + "/t_ys"] = t_ys
        out[name + "/t_xs"] = t_xs
        out[name + "/idx"] = idx
    dst = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "mas_cases.npz")
    np.savez_compressed(dst, **out)
    print("wrote", os.path.abspath(dst), os.path.getsize(dst), "bytes;", len(out) // 4, "cases")


if __name__ == "__main__":
    main()
