#!/usr/bin/env python3
"""Locates the irreproducible element of the duration flows' spline backward without touching the kernel:
every vits_flow_spline_bwd call of the captured step gets its inputs (x2, h, mask, dy2, dlogdet) and outputs (dx2, gh) cloned
into static buffers by extra copy nodes of the graph; after each replay from the SAME state the host compares them with the
first replay's.  Prints, per differing tensor, where and by what it differs.  Run against a libvitsmi.so whose rq_spline.hip
was compiled with and without csrc/Makefile's former FLAGS_rq_spline."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from importlib import import_module
P = import_module("personalized_text-to-speech_amd"); cfgs = import_module("personalized_text-to-speech_amd.configs")
tr = import_module("personalized_text-to-speech_amd.train"); rowops = import_module("personalized_text-to-speech_amd.rowops")
K = P.kernels

taps, active = [], [False]
orig_bwd = rowops.FlowTailFn.backward
orig_mask = K.lrelu_mask_bwd


def tapped_bwd(ctx, dout, dlogdet):
    if not active[0]:
        return orig_bwd(ctx, dout, dlogdet)
    xd, hd, md = ctx.saved_tensors
    pre = {"x2": xd.clone(), "h": hd.clone(), "mask": md.clone(), "dy2": dout.float().clone(), "dlogdet": dlogdet.float().clone()}
    res = orig_bwd(ctx, dout, dlogdet)
    rec = {"dx2": res[0].clone(), "gh": res[1].clone()}
    res_b = orig_bwd(ctx, dout, dlogdet)                       # the same launch again, right behind the first, into other buffers
    rec.update({"dx2_again": res_b[0].clone(), "gh_again": res_b[1].clone()})
    rec.update({k + "_before": v for k, v in pre.items()})
    rec.update({"x2_after": xd.clone(), "h_after": hd.clone(), "mask_after": md.clone(), "dy2_after": dout.float().clone(),
                "dlogdet_after": dlogdet.float().clone()})
    taps.append(rec)
    return res


def tapped_mask(dy, y=None, slope=1.0, lengths=None):
    out = orig_mask(dy, y, slope, lengths)
    if active[0] and dy.dim() == 3 and dy.size(2) == 32 and y is None:
        taps.append({"maskbwd_in": dy.clone(), "maskbwd_out": out.clone()})
    return out


rowops.FlowTailFn.backward = staticmethod(tapped_bwd)
K.lrelu_mask_bwd = tapped_mask
import_module("personalized_text-to-speech_amd.wn_cl").K.lrelu_mask_bwd = tapped_mask

cfg_name, batch_size, t_y_range = cfgs.WORKLOADS[os.environ.get("WL", "C2")]
hps = cfgs.get(cfg_name)
ft = tr.FineTuner(hps, "cuda:0", amp=True)
batch = tr.synthetic_batch(hps, batch_size, t_y_range, "cuda:0")
# warm up eagerly (untapped), then capture with the taps on
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        ft.step(batch)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
active[0] = True
ft.capture(batch, warmup=0, verify=False)
active[0] = False
print(f"{len(taps)} tapped calls in the captured step", flush=True)
ts = ft._state_tensors()
snap = [t.detach().clone() for t in ts]
rng = torch.cuda.get_rng_state(ft.device)


def run():
    with torch.no_grad():
        for t, s in zip(ts, snap):
            t.copy_(s)
    torch.cuda.set_rng_state(rng, ft.device)
    ft.replay()
    torch.cuda.synchronize()
    return [{k: v.detach().cpu().clone() for k, v in rec.items()} for rec in taps], ft.optim_g.flat_p.detach().cpu().clone()


def within(cur, tag):
    for ci, rec in enumerate(cur):
        if "gh_again" not in rec:
            continue
        for a, b in (("gh", "gh_again"), ("dx2", "dx2_again")) + tuple((k + "_before", k + "_after") for k in ("x2", "h", "mask", "dy2", "dlogdet")):
            if not torch.equal(rec[a], rec[b]):
                A, B = rec[a].float(), rec[b].float()
                idx = (A != B).nonzero()
                print(f"   WITHIN {tag} call {ci}: {a} != {b} in {idx.size(0)} elements", flush=True)
                if a == "gh":
                    A2, B2 = A.reshape(-1, A.size(-1)), B.reshape(-1, B.size(-1))
                    rows = sorted({int(i[0]) * A.size(1) + int(i[1]) for i in idx.tolist()})
                    print(f"      flat rows {rows[0]}..{rows[-1]} ({len(rows)}); lanes {rows[0] % 64}..{rows[-1] % 64}; channels {sorted({int(i[2]) for i in idx.tolist()})}")
                    for rr in rows[:3]:
                        print(f"      row {rr} first : {[f'{v:.4e}' for v in A2[rr].tolist()]}")
                        print(f"      row {rr} second: {[f'{v:.4e}' for v in B2[rr].tolist()]}")
                        ch = [int(i[2]) for i in idx.tolist() if int(i[0]) * A.size(1) + int(i[1]) == rr]
                        for cc in ch[:2]:
                            for nm, X, Y in (("first", A2, B2), ("second", B2, A2)):
                                same_row = [j for j in range(X.size(1)) if X[rr, j] == Y[rr, cc] and j != cc]
                                other_rows = (X[:, cc] == Y[rr, cc]).nonzero().flatten().tolist()[:6]
                                print(f"         value of the {('second','first')[nm=='second']} run at channel {cc} also found in the {nm} run: same row channels {same_row}, same channel rows {other_rows}")


NR = int(os.environ.get("NREP", "30"))
base, p0 = run()
within(base, "replay 1")
n_bad = 0
for r in range(1, NR):
    cur, p1 = run()
    within(cur, f"replay {r + 1}")
    lines = []
    for ci, (a, b) in enumerate(zip(base, cur)):
        for k in a:
            if not torch.equal(a[k], b[k]):
                A, B = a[k].float(), b[k].float()
                idx = (A != B).nonzero()
                rows = sorted({tuple(i[:-1]) for i in idx.tolist()})
                lines.append(f"   call {ci} {k} shape {tuple(A.shape)}: {idx.size(0)} elements differ; first {idx[:6].tolist()}; rows {rows[0]}..{rows[-1]} ({len(rows)} rows)")
                for ix in idx[:2].tolist():
                    row = ix[0]
                    lines.append(f"      at {ix}: base {A[tuple(ix)].item():.6e}  now {B[tuple(ix)].item():.6e}")
                    if A.dim() == 2:
                        lines.append(f"      base row {row}: {[f'{v:.3e}' for v in A[row].tolist()]}")
                        lines.append(f"      now  row {row}: {[f'{v:.3e}' for v in B[row].tolist()]}")
                        if row + 1 < A.size(0):
                            lines.append(f"      base row {row + 1}: {[f'{v:.3e}' for v in A[row + 1].tolist()]}")
                        if row >= 1:
                            lines.append(f"      base row {row - 1}: {[f'{v:.3e}' for v in A[row - 1].tolist()]}")
                        # is the odd value present anywhere in the base tensor's same column?
                        col = ix[1]
                        hits = (A[:, col] == B[tuple(ix)]).nonzero().flatten().tolist()[:8]
                        lines.append(f"      rows of base whose column {col} equals the odd value: {hits}")
    pd = int((p0 != p1).sum())
    if lines or pd:
        n_bad += 1
        print(f"--- replay {r + 1}: {pd} parameter elements differ from replay 1", flush=True)
        print("\n".join(lines), flush=True)
print(f"{n_bad} of {NR - 1} replays differ from the first", flush=True)
