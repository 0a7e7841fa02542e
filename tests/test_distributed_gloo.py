"""CPU, 2 processes over gloo: distributed.GradBuckets averages gradients like DDP and
broadcast_parameters aligns the replicas (the N>1 path of bench.py / train.FineTuner)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    from importlib import import_module
    D = import_module("personalized_text-to-speech_amd.distributed")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # different init per rank on purpose
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Linear(16, 2))
    unused = torch.nn.Parameter(torch.ones(3))          # never receives a gradient
    D.broadcast_parameters(net)
    params = list(net.parameters()) + [unused]
    buckets = D.GradBuckets(params, bucket_bytes=600)   # several buckets
    assert len(buckets.buckets) > 2
    torch.manual_seed(7 + rank)
    x = torch.randn(5, 6)
    res = {}
    for step in range(2):                               # second step: grads re-zeroed, hooks re-armed
        buckets.zero_grad()
        loss = net(x * (step + 1)).pow(2).mean()
        loss.backward()
        local = [p.grad.clone() for p in net.parameters()]      # before reduction? (may already be reduced)
        buckets.finish()
        res[step] = [p.grad.clone() for p in net.parameters()]
    # reference: manual average of per-rank local gradients recomputed without the reducer
    ref = []
    for step in range(2):
        gs = torch.autograd.grad(net(x * (step + 1)).pow(2).mean(), list(net.parameters()))
        gs = [g.clone() for g in gs]
        for g in gs:
            dist.all_reduce(g)
            g.div_(world)
        ref.append(gs)
    ok = all(torch.allclose(a, b, atol=1e-6) for s in range(2) for a, b in zip(res[s], ref[s]))
    # manual mode (the captured, three-graph step): hooks off, pack() then all_reduce() give the same averages
    buckets.manual(True)
    buckets.zero_grad()
    net(x).pow(2).mean().backward()
    buckets.finish()                                   # a no-op in manual mode
    buckets.pack()
    buckets.all_reduce()
    ok &= all(torch.allclose(p.grad, b, atol=1e-6) for p, b in zip(net.parameters(), ref[0]))
    ok &= float(unused.grad.abs().sum()) == 0.0
    inside = lambda g: any(f.data_ptr() <= g.data_ptr() and g.data_ptr() + g.numel() * 4 <= f.data_ptr() + f.numel() * 4 for f in buckets._exchange)
    placed = all(inside(p.grad) for p in list(net.parameters()) + [unused])
    # gradients that already lie back to back in one allocation (the weight arenas' layout) are exchanged where they are
    plist = list(net.parameters())
    arena = torch.zeros(sum(p.numel() for p in plist))
    buckets.zero_grad()
    buckets.INPLACE_MIN, off = 16, 0
    local = torch.autograd.grad(net(x).pow(2).mean(), plist)
    for p, g in zip(plist, local):
        v = arena[off:off + p.numel()].view_as(p); v.copy_(g); p.grad = v; off += p.numel()
    buckets.pack()
    placed &= any(f.data_ptr() == arena.data_ptr() and f.numel() == arena.numel() for f in buckets._exchange)
    placed &= all(p.grad.data_ptr() >= arena.data_ptr() and p.grad.data_ptr() < arena.data_ptr() + arena.numel() * 4 for p in plist)
    buckets.all_reduce()
    ok &= all(torch.allclose(p.grad, b, atol=1e-6) for p, b in zip(plist, ref[0]))
    buckets.manual(False)
    same_params = True
    for p in net.parameters():
        t = p.data.clone()
        dist.broadcast(t, 0)
        same_params &= bool(torch.equal(t, p.data))
    out[rank] = (ok, same_params, float(unused.grad.abs().sum()), placed)
    dist.destroy_process_group()


def test_grad_buckets_two_ranks():
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        for r in range(2):
            ok, same, unused, flat_views = out[r]
            assert ok and same and unused == 0.0 and flat_views
