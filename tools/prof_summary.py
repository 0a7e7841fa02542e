#!/usr/bin/env python3
"""Steady-state per-step kernel summary from a rocprofv3 --kernel-trace CSV.

The fine-tune step launches the alignment kernel (mas_kernel) exactly once, so consecutive
mas_kernel start times delimit one step.  The last `--steps` such windows are aggregated by
kernel name: calls/step, total and average duration, share of the step's kernel time.
usage: prof_summary.py <kernel_trace.csv> [--steps 3] [--top 40] [--marker mas_kernel]
"""
import argparse
import collections
import csv
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--marker", default="mas_kernel")
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    if len(marks) < a.steps + 1:
        sys.exit(f"only {len(marks)} marker launches")
    lo, hi = marks[-a.steps - 1], marks[-1]
    win = rows[lo:hi]
    wall = (rows[hi][0] - rows[lo][0]) / a.steps
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in win:
        agg[n][0] += 1
        agg[n][1] += e - s
    busy = sum(v[1] for v in agg.values()) / a.steps
    print(f"# steady-state window: {a.steps} steps, wall {wall/1e6:.3f} ms/step, kernel-busy {busy/1e6:.3f} ms/step, "
          f"{len(win)/a.steps:.0f} launches/step, {len(agg)} distinct kernels")
    cats = collections.OrderedDict([
        # every kernel of libvitsmi.so lives in an anonymous namespace (mangled _ZN12_GLOBAL__N_1... or demangled "(anonymous namespace)::");
        # sub-categories below split them by what they do
        ("hip: convolutions fwd / data gradient (tiled, ring, flat-row, grouped, WaveNet layer)", ("conv1d_cl_kernel", "conv1d_ring_kernel", "conv1d_flat_kernel", "small_m_kernel", "grouped_fwd", "grouped_dgrad", "wn_layer", "first_fwd", "first_dgrad", "post_fwd", "post_dgrad", "fold_kernel", "unfold_kernel", "neg_cent_kernel")),
        ("hip: weight / bias gradients (+ slab reductions)", ("wgrad", "reduce_slabs", "reduce_pending")),
        ("hip: weight arena, packing, AdamW", ("prep_fwd", "prep_bwd", "transpose_tiles", "wn_pack", "adamw_kernel", "gradnorm")),
        ("MIOpen convolution (+layout/im2col helpers)", ("igemm_", "naive_conv", "ck::", "_ZN2ck", "Im2d2Col", "Col2Im", "batched_transpose", "SubTensorOp", "miopen", "Im3d", "gridwise")),
        ("rocBLAS/hipBLASLt GEMM", ("Cijk_",)),
        ("aten elementwise/copy/cast", ("elementwise_kernel", "vectorized_elementwise", "CatArrayBatchedCopy", "index", "fill", "copy")),
        ("aten reductions/norms", ("reduce_kernel", "layer_norm", "softmax", "cunn_", "RowwiseMoments", "norm")),
        ("optimizer (multi-tensor)", ("multi_tensor_apply",)),
        ("FFT", ("fft", "rocfft", "transpose_kernel", "real_post", "r2c", "c2r")),
        # (last: torch's own kernels also carry "(anonymous namespace)" inside at::native::...; they were matched above)
        ("hip: row kernels, reductions, alignment, losses (other kernels of this repo)", ("_GLOBAL__N_1", "(anonymous namespace)::")),
    ])
    ctot = collections.OrderedDict((k, [0, 0]) for k in list(cats) + ["other"])
    for n, (c, t) in agg.items():
        for cat, pats in cats.items():
            if any(p_ in n for p_ in pats) and not (cat.startswith("hip:") and "at::" in n):
                ctot[cat][0] += c; ctot[cat][1] += t
                break
        else:
            ctot["other"][0] += c; ctot["other"][1] += t
    print("# by category:")
    for cat, (c, t) in ctot.items():
        print(f"#   {t/a.steps/1e6:9.3f} ms/step {100*t/a.steps/busy:6.2f}%  {c/a.steps:8.0f} launches/step  {cat}")
    print(f"# {'ms/step':>9} {'share':>7} {'calls/step':>10} {'avg_us':>9}  kernel")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[: a.top]:
        print(f"  {t/a.steps/1e6:9.3f} {100*t/a.steps/busy:6.2f}% {c/a.steps:10.1f} {t/c/1e3:9.1f}  {n[:120]}")


if __name__ == "__main__":
    main()
