"""Debug: out-of-bounds write check of the flat-row kernels at the period-discriminator shapes (sentinel guards)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ptts_amd as pkg
K = pkg.kernels
dev = "cuda"
dt = torch.bfloat16
G = 4096


def guarded(shape, dtype):
    n = 1
    for s in shape:
        n *= s
    buf = torch.full((n + 2 * G,), 7.0, device=dev, dtype=dtype)
    return buf, buf[G:G + n].view(shape)


def check(buf, n, what):
    ok = bool((buf[:G] == 7).all() and (buf[G + n:] == 7).all())
    if not ok:
        lo = (buf[:G] != 7).nonzero().flatten(); hi = (buf[G + n:] != 7).nonzero().flatten()
        print("  OOB WRITE", what, "below:", lo[:4].tolist(), len(lo), "above:", hi[:4].tolist(), len(hi))
    return ok


orig_ws = K.workspace
state = {}
def ws_guard(nbytes, device):
    buf = torch.full((nbytes + 2 * G,), 7, device=dev, dtype=torch.uint8)
    state["ws"] = (buf, nbytes)
    return buf[G:G + nbytes]
K.workspace = ws_guard

n_items = int(os.environ.get("N", 32))
for p in (2, 3, 5, 7, 11):
    rows = (8192 + p - 1) // p
    b = n_items * p
    t = rows
    for (ci, co, k, st, pd) in [(8, 32, 5, 3, 2), (32, 128, 5, 3, 2), (128, 512, 5, 3, 2), (512, 1024, 5, 3, 2), (1024, 1024, 5, 1, 2), (1024, 8, 3, 1, 1)]:
        t_out = (t + 2 * pd - (k - 1) - 1) // st + 1
        x = torch.randn(b, t, ci, device=dev).to(dt)
        w = (torch.randn(k, co, ci, device=dev) / (ci * k) ** 0.5).to(dt)
        bias = torch.randn(co, device=dev)
        ybuf, y = guarded((b, t_out, co), dt)
        K.conv1d_cl_raw(x, w, bias, out=y, pad=pd, stride=st, out_slope=0.1)
        ok1 = check(ybuf, y.numel(), f"fwd p{p} {ci}->{co}")
        dy = torch.randn(b, t_out, co, device=dev).to(dt)
        dxbuf, dx = guarded((b, t, ci), dt)
        wb = w.flip(0).transpose(1, 2).contiguous()
        K.conv1d_cl_raw(dy, wb, out=dx, pad=k - 1 - pd, in_div=st, t_out=t) if st > 1 else K.conv1d_cl_raw(dy, wb, out=dx, pad=k - 1 - pd)
        ok2 = check(dxbuf, dx.numel(), f"dgrad p{p} {ci}->{co}")
        dwbuf, dw = guarded((k, co, ci), torch.float32)
        dbbuf, db = guarded((co,), torch.float32)
        K.conv1d_cl_wgrad_raw(x, dy, k, pad=pd, stride=st, out=dw, dbias=db)
        ok3 = check(dwbuf, dw.numel(), f"wgrad p{p} {ci}->{co}") and check(dbbuf, co, "dbias")
        wsb, nb = state["ws"]
        ok4 = bool((wsb[:G] == 7).all() and (wsb[G + nb:] == 7).all())
        if not ok4:
            print("  OOB WRITE workspace", p, ci, co, nb)
        torch.cuda.synchronize()
        print(f"p{p} b{b} t{t}->{t_out} {ci}->{co} k{k} s{st}:", "ok" if (ok1 and ok2 and ok3 and ok4) else "BAD", flush=True)
        t = t_out
