#!/usr/bin/env python3
"""Back-to-back timing of vits_conv1d_cl_wgrad on the layer shapes that dominate the step (HIP events, 20 launches each)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from importlib import import_module
P = import_module("personalized_text-to-speech_amd"); K = P.kernels
shapes = [  # b, t, ci, co, k, stride, dil, pad, in_slope
    (352, 10, 1024, 1024, 5, 1, 1, 2, 1.0), (64, 51, 1024, 1024, 5, 1, 1, 2, 1.0), (352, 28, 512, 1024, 5, 3, 1, 2, 1.0), (64, 152, 512, 1024, 5, 3, 1, 2, 1.0),
    (96, 304, 128, 512, 5, 3, 1, 2, 1.0), (16, 2048, 128, 128, 11, 1, 1, 5, 0.1), (16, 2048, 128, 128, 7, 1, 1, 3, 0.1), (16, 2048, 128, 128, 3, 1, 1, 1, 0.1),
    (16, 256, 256, 256, 11, 1, 1, 5, 0.1), (16, 256, 256, 256, 3, 1, 1, 1, 0.1), (16, 4096, 64, 64, 11, 1, 1, 5, 0.1), (16, 4096, 64, 64, 3, 1, 1, 1, 0.1),
    (16, 201, 768, 192, 3, 1, 1, 1, 1.0), (16, 201, 192, 768, 3, 1, 1, 1, 1.0), (16, 500, 192, 384, 5, 1, 1, 2, 1.0), (16, 8192, 32, 32, 11, 1, 1, 5, 0.1),
]
only = os.environ.get('UBW_ONLY')
FLAGS = int(os.environ.get('UBW_FLAGS', '0'))
for idx, (b, t, ci, co, kk, st, dl, pd, sl) in enumerate(shapes):
    if only and str(idx) not in only.split(','):
        continue
    x = torch.randn(b, t, ci, device="cuda").bfloat16()
    t_out = (t + 2 * pd - dl * (kk - 1) - 1) // st + 1
    dy = torch.randn(b, t_out, co, device="cuda").bfloat16()
    db = torch.empty(co, device="cuda")
    out = torch.empty(kk, co, ci, device="cuda")
    for _ in range(3):
        K.conv1d_cl_wgrad_raw(x, dy, kk, pad=pd, stride=st, dil=dl, in_slope=sl, dbias=db, out=out, flags=FLAGS)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        K.conv1d_cl_wgrad_raw(x, dy, kk, pad=pd, stride=st, dil=dl, in_slope=sl, dbias=db, out=out, flags=FLAGS)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * b * t_out * ci * co * kk
    print(f"b{b:4d} t{t:5d} ci{ci:5d} co{co:5d} k{kk:3d} s{st} d{dl}  {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s (incl. second stage)", flush=True)
