#!/usr/bin/env python3
"""Micro-benchmark of vits_mas_f32 on one GPU: time per call and algorithmic GB/s
(4 B read + 4 B written per cell)."""
import argparse
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import ptts_amd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for name, b, lo, hi in [("C2", 16, 200, 500), ("C3", 64, 300, 800), ("max", 64, 1000, 1000)]:
        t_ys = np.linspace(lo, hi, b).round().astype(np.int32)[::-1].copy()
        t_xs = (2 * np.round(t_ys / 5) + 1).astype(np.int32)
        if name == "max":
            t_xs[:] = 381
        t_t, t_s = int(t_ys.max()), int(t_xs.max())
        nc = torch.randn(b, t_t, t_s, device=dev) * 40 - 300
        ty, tx = torch.from_numpy(t_ys).to(dev), torch.from_numpy(t_xs).to(dev)
        for _ in range(3):
            ptts_amd.monotonic_align.maximum_path_lengths(nc, ty, tx)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            ptts_amd.monotonic_align.maximum_path_lengths(nc, ty, tx)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        byts = 8.0 * b * t_t * t_s
        print(f"{name}: b={b} t_t={t_t} t_s={t_s}  {us:8.1f} us/call  {byts/us/1e3:7.1f} GB/s algorithmic "
              f"({b*t_t*t_s/us:.0f} Mcell/s)")


if __name__ == "__main__":
    main()
