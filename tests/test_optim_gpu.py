"""optim.FlatAdamW / csrc/adamw.hip against torch.optim.AdamW (what the reference constructs, finetune_speaker_v2.py:113-120)
and the gradient norm against commons.clip_grad_value_'s definition (reference commons.py:149-164)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _nets(seed=0):
    torch.manual_seed(seed)
    shapes = [(1,), (3,), (7, 5), (4096,), (4097,), (64, 33, 5), (300000,), (1, 1, 1), (129, 1), (2, 3, 4, 5)]
    a = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    return a, b


def _grads(params, step, flat=None, unaligned=False):
    gen = torch.Generator(device=DEV).manual_seed(100 + step)
    gs = [torch.randn(p.shape, device=DEV, generator=gen) * (10.0 ** ((i % 5) - 3)) for i, p in enumerate(params)]
    return gs


def test_matches_torch_adamw_over_steps(pkg):
    optim = pkg.optim
    a, b = _nets()
    kw = dict(lr=2e-4, betas=(0.8, 0.99), eps=1e-9, weight_decay=0.01)
    ref = torch.optim.AdamW(b, **kw)
    runs = [[a[5], a[2], a[6]]]                                     # packed back to back, in this order
    opt = optim.FlatAdamW(a, runs=runs, **kw)
    assert all(p.data.untyped_storage().data_ptr() == opt.flat_p.untyped_storage().data_ptr() for p in a)
    # gradients of the run in ONE buffer in the run's order (as a weight arena hands them out); one gradient at an odd address
    run_buf = torch.empty(sum(p.numel() for p in runs[0]) + 1, device=DEV)
    for step in range(6):
        gs = _grads(a, step)
        off = 0
        for p in runs[0]:
            i = [id(q) for q in a].index(id(p))
            v = run_buf[off:off + p.numel()].view(p.shape); v.copy_(gs[i]); gs[i] = v; off += p.numel()
        odd = torch.empty(a[3].numel() + 1, device=DEV)[1:]
        odd.copy_(gs[3]); gs[3] = odd.view(a[3].shape)               # 4-byte aligned only: the kernel's scalar path
        for p, q, g in zip(a, b, gs):
            p.grad, q.grad = g, g.clone()
        want_norm = torch.sqrt(sum((q.grad.double() ** 2).sum() for q in b if q.grad is not None))
        opt.step(); ref.step()
        assert opt.last_entries <= len(a) - 2                       # the run merged into one entry
        assert abs(float(opt.grad_norm) - float(want_norm)) <= 1e-6 * float(want_norm)
        for p, q in zip(a, b):
            assert torch.allclose(p, q, rtol=2e-6, atol=1e-7), (step, p.shape, float((p - q).abs().max()))
    for p, q in zip(a, b):
        st, rt = opt.state[p], ref.state[q]
        scale = float(rt["exp_avg"].abs().max())
        assert float((st["exp_avg"] - rt["exp_avg"]).abs().max()) <= 2e-6 * scale, p.shape
        assert torch.allclose(st["exp_avg_sq"], rt["exp_avg_sq"], rtol=1e-5, atol=1e-12), p.shape
    assert float(opt.dev_state[1]) == 6.0
    # a parameter without a gradient is left alone (its moments too)
    before = [t.clone() for t in (a[7], opt.state[a[7]]["exp_avg"], a[0])]
    for p, g in zip(a, _grads(a, 9)):
        p.grad = g
    a[7].grad = None
    opt.step()
    assert torch.equal(a[7], before[0]) and torch.equal(opt.state[a[7]]["exp_avg"], before[1]) and not torch.equal(a[0], before[2])


def test_state_dict_round_trips_with_torch_adamw(pkg):
    optim = pkg.optim
    a, b = _nets(1)
    kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    opt, ref = optim.FlatAdamW(a, **kw), torch.optim.AdamW(b, **kw)
    for step in range(2):
        for p, q, g in zip(a, b, _grads(a, step)):
            p.grad, q.grad = g, g.clone()
        opt.step(); ref.step()
    # torch -> flat (a checkpoint written by the reference), flat -> torch
    sd_ref, sd_flat = copy.deepcopy(ref.state_dict()), copy.deepcopy(opt.state_dict())
    assert sd_ref["state"].keys() == sd_flat["state"].keys() and set(sd_flat["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    a2, b2 = _nets(1)
    opt2, ref2 = optim.FlatAdamW(a2, **kw), torch.optim.AdamW(b2, **kw)
    with torch.no_grad():
        for p, q, r in zip(a2, b2, b):
            p.copy_(r); q.copy_(r)
    sd_ref["param_groups"][0]["lr"] = 5e-4
    sd_flat["param_groups"][0]["lr"] = 5e-4
    opt2.load_state_dict(sd_ref); ref2.load_state_dict(sd_flat)
    assert float(opt2.dev_state[1]) == 2.0 and abs(float(opt2.dev_state[0]) - 5e-4) < 1e-10
    assert all(opt2.state[p]["exp_avg"].untyped_storage().data_ptr() == opt2.flat_m.untyped_storage().data_ptr() for p in a2)
    for p, q, g in zip(a2, b2, _grads(a2, 7)):
        p.grad, q.grad = g, g.clone()
    opt2.step(); ref2.step()
    for p, q in zip(a2, b2):
        assert torch.allclose(p, q, rtol=2e-6, atol=1e-7)


def test_grad_norm_l2_matches_definition(pkg):
    a, _ = _nets(2)
    for p, g in zip(a, _grads(a, 3)):
        p.grad = g
    a[1].grad = None
    want = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in a if p.grad is not None))
    got = pkg.commons.grad_norm_l2(a)
    assert got.dim() == 0 and abs(float(got) - float(want)) <= 1e-6 * float(want)
    assert abs(float(pkg.commons.clip_grad_value_(a, None)) - float(want)) <= 1e-6 * float(want)
    assert torch.equal(pkg.commons.grad_norm_l2(a), got)                      # fixed-order sums: reproducible
