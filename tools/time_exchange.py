#!/usr/bin/env python3
"""Where the data-parallel (three-graph) form of the captured step spends its time at ONE rank: device events after every
graph replay and every bucket exchange + host timestamps of the enqueue calls, C2 workload, one-rank RCCL group.

    python tools/time_exchange.py [steps] [plain]        plain: no process group, no buckets — the three graphs alone"""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.distributed as dist
from importlib import import_module

plain = "plain" in sys.argv[2:]
torch.cuda.set_device(0)
if not plain:
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29671", rank=0, world_size=1)
cfgs = import_module("personalized_text-to-speech_amd.configs"); tr = import_module("personalized_text-to-speech_amd.train")
cfg_name, batch_size, t_y_range = cfgs.WORKLOADS["C2"]
hps = cfgs.get(cfg_name)
ft = tr.FineTuner(hps, "cuda:0", amp=True, force_exchange=not plain)
batch = tr.synthetic_batch(hps, batch_size, t_y_range, "cuda:0")
ft.capture_segments(batch, warmup=2, verify=False)
ga, gb, gc = ft._graph
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(3):
    ft.replay()
torch.cuda.synchronize()
names = ("graph A", "all-reduce D", "graph B", "all-reduce G", "graph C")
calls = (ga.replay, ft.buckets_d.all_reduce, gb.replay, ft.buckets_g.all_reduce, gc.replay)
dev = [0.0] * 5; host = [0.0] * 5
t_all = time.perf_counter()
for _ in range(steps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    for i, f in enumerate(calls):
        t0 = time.perf_counter(); f(); host[i] += time.perf_counter() - t0
        ev[i + 1].record()
    torch.cuda.synchronize()
    for i in range(5):
        dev[i] += ev[i].elapsed_time(ev[i + 1])
t_all = (time.perf_counter() - t_all) / steps * 1e3
print(f"{steps} steps, {t_all:.2f} ms/step wall (with a sync per step)")
for i, n in enumerate(names):
    print(f"  {n:14s} device {dev[i] / steps:7.3f} ms   host enqueue {host[i] / steps * 1e3:7.3f} ms")
print(f"  buckets: D {[b[0].numel() * 4 >> 20 for b in ft.buckets_d.buckets]} MiB, G {[b[0].numel() * 4 >> 20 for b in ft.buckets_g.buckets]} MiB")
if not plain:
    dist.destroy_process_group()
