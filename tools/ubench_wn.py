#!/usr/bin/env python3
"""Times the one-launch WaveNet layer kernels (csrc/wn_layer.hip) and the stack's batched weight gradient on the step's shapes,
back to back on one stream (warm caches) and interleaved with a 512 MiB copy (cold L2 / Infinity Cache), HIP events.
  python3 tools/ubench_wn.py [b t H k L]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from importlib import import_module
P = import_module("personalized_text-to-speech_amd")
K = P.kernels
b, t, H, k, L = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else (16, 500, 192, 5, 4)
dev, dt = "cuda:0", torch.bfloat16
torch.manual_seed(0)
lengths = torch.linspace(t, t * 0.4, b).round().to(torch.int32).to(dev)
x = torch.randn(b, t, H, device=dev).to(dt)
w_in = [torch.randn(k, 2 * H, H, device=dev).mul_(0.05).to(dt) for _ in range(L)]
w_rs = [torch.randn(1, H if i == L - 1 else 2 * H, H, device=dev).mul_(0.05).to(dt) for i in range(L)]
w_in_t = [w.flip(0).transpose(1, 2).contiguous() for w in w_in]
w_rs_t = [w.transpose(1, 2).contiguous() for w in w_rs]
b_in = [torch.randn(2 * H, device=dev) * 0.1 for _ in range(L)]
b_rs = [torch.randn(w.size(1), device=dev) * 0.1 for w in w_rs]
cond = torch.randn(L, b, 2 * H, device=dev) * 0.1
packed = K.WnPacked(H, k, L, dt, dev)
pre = [torch.empty(b, t, 2 * H, device=dev, dtype=dt) for _ in range(L)]
acts = [torch.empty(b, t, H, device=dev, dtype=dt) for _ in range(L)]
hs = [x] + [None] * L
skip = torch.empty(b, t, H, device=dev, dtype=dt)
d_o = torch.randn(b, t, H, device=dev).to(dt)
d_pre = torch.empty(L, b, t, 2 * H, device=dev, dtype=dt)
dh = [torch.empty(b, t, H, device=dev, dtype=dt) for _ in range(L)]
dw_in = [torch.empty(k, 2 * H, H, device=dev) for _ in range(L)]
dw_rs = [torch.empty(1, w.size(1), H, device=dev) for w in w_rs]
db_in = [torch.empty(2 * H, device=dev) for _ in range(L)]
db_rs = [torch.empty(w.size(1), device=dev) for w in w_rs]
junk_a = torch.empty(128 << 20, dtype=torch.float32, device=dev); junk_b = torch.empty_like(junk_a)


def pack():
    packed.fill(list(zip(w_in, w_rs)), list(zip(w_rs_t, w_in_t)))


def fwd():
    for i in range(L):
        hs[i + 1] = K.wn_layer_fwd(hs[i], packed, i, b_in[i], cond[i], b_rs[i], lengths, 1, skip, accumulate=i > 0, last=i == L - 1, pre=pre[i], acts=acts[i])


def bwd():
    d_h = None
    for i in reversed(range(L)):
        K.wn_layer_bwd(d_h, d_o, pre[i], packed, i, lengths, 1, i == L - 1, d_pre[i], dh[i])
        d_h = dh[i]


def wgrad():
    batch = []
    for i in range(L):
        last = i == L - 1
        if last:
            batch.append(dict(x=acts[i], dy=d_o, k=1, out=dw_rs[i], dbias=db_rs[i]))
        else:
            batch.append(dict(x=acts[i], dy=dh[i + 1], k=1, out=dw_rs[i][:, :H], dbias=db_rs[i][:H]))
            batch.append(dict(x=acts[i], dy=d_o, k=1, out=dw_rs[i][:, H:], dbias=db_rs[i][H:]))
        batch.append(dict(x=hs[i], dy=d_pre[i], k=k, pad=(k - 1) // 2, out=dw_in[i], dbias=db_in[i]))
    assert K.conv1d_cl_wgrad_batch(batch, None)


def timeit(fn, n=20, cold=False):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        if cold:
            junk_b.copy_(junk_a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3


pack(); fwd(); bwd(); torch.cuda.synchronize()
fl = 2.0 * b * t * H * (2 * H * k + 2 * H)
for name, fn, per in (("pack (stack)", pack, 1), ("fwd / layer", fwd, L), ("bwd / layer", bwd, L), ("wgrad batch / layer", wgrad, L)):
    w, c = timeit(fn) / per, timeit(fn, cold=True) / per
    extra = f"  {fl / w / 1e6:7.1f} TFLOP/s warm" if "layer" in name and "wgrad" not in name else ""
    print(f"{name:22s} warm {w:8.1f} us   cold {c:8.1f} us{extra}", flush=True)
