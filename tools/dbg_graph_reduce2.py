"""Debug: feature-map style reduction of a kernel output under graph replay."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ptts_amd as pkg
K = pkg.kernels
dev = "cuda"
dt = torch.bfloat16
torch.manual_seed(0)
n, p_ = 32, 2
b, t, ci, co, k = n * p_, 1366, 32, 128, 5
x = torch.randn(b, t, ci, device=dev).to(dt)
w = (torch.randn(k, co, ci, device=dev) / (ci * k) ** 0.5).to(dt)


def term(h, mode):
    f = h.view(n, p_, h.size(1), h.size(2)).permute(0, 3, 2, 1)
    a, c = f[:16], f[16:]
    if mode == "views":
        return (a.float() - c.float()).abs().mean()
    if mode == "cl":
        return (h[:b // 2].float() - h[b // 2:].float()).abs().mean()
    if mode == "nofloat":
        return (a - c).abs().mean()
    if mode == "sum":
        return (a.float() - c.float()).abs().sum()
    if mode == "double":
        return (a.double() - c.double()).abs().mean()


for src in ("kernel", "clone"):
    for mode in ("views", "cl", "nofloat", "sum", "double"):
        def run():
            h = K.conv1d_cl_raw(x, w, pad=2, stride=3, out_slope=0.1) if src == "kernel" else (x * 2)
            return term(h, mode)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                r = run()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        ref = r.item()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            r = run()
        vals = []
        for it in range(3):
            junk = torch.full((1 << 26,), float("nan"), device=dev); del junk
            g.replay(); torch.cuda.synchronize(); vals.append(round(r.item(), 5))
        print(f"{src:7s} {mode:8s} ref {ref:.5f} replays {vals}", flush=True)
