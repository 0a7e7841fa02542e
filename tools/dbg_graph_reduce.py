"""Debug: does a torch global reduction survive graph replay next to each of this repo's kernels?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ptts_amd as pkg
K = pkg.kernels
dev = "cuda"
dt = torch.bfloat16
torch.manual_seed(0)
big = torch.rand(64, 1366, 32, device=dev)

b, t, ci, co, k = 64, 456, 128, 512, 5
x = torch.randn(b, t, ci, device=dev).to(dt)
w = (torch.randn(k, co, ci, device=dev) / (ci * k) ** 0.5).to(dt)
wb = w.flip(0).transpose(1, 2).contiguous()
t_out = (t + 4 - 4 - 1) // 3 + 1
dy = torch.randn(b, t_out, co, device=dev).to(dt)


def nothing():
    return None
def conv_flat():
    return K.conv1d_cl_raw(x, w, pad=2, stride=3, out_slope=0.1)
def conv_classic():
    return K.conv1d_cl_raw(x, w, pad=2)
def dgrad_div():
    return K.conv1d_cl_raw(dy, wb, pad=2, in_div=3, t_out=t)
def wgrad_flat():
    return K.conv1d_cl_wgrad_raw(x, dy, k, pad=2, stride=3, dbias=torch.empty(co, device=dev))
dy1 = torch.randn(b, t, co, device=dev).to(dt)
def wgrad_classic():
    return K.conv1d_cl_wgrad_raw(x, dy1, k, pad=2, dbias=torch.empty(co, device=dev))


for name, fn in [("nothing", nothing), ("conv_flat", conv_flat), ("conv_classic", conv_classic), ("dgrad_div", dgrad_div),
                 ("wgrad_flat", wgrad_flat), ("wgrad_classic", wgrad_classic)]:
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn(); r = big.abs().mean()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    ref = r.item()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o = fn()
        r = big.abs().mean()
        r2 = (big - 0.5).abs().mean()
    vals = []
    for it in range(3):
        g.replay(); torch.cuda.synchronize(); vals.append((r.item(), r2.item()))
    print(f"{name:14s} ref {ref:.6f} replays {vals}", flush=True)
