"""Debug: MultiPeriodDiscriminator forward+backward captured in a graph and replayed, vs eager."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ptts_amd as pkg

torch.manual_seed(0)
dev = "cuda"
d = pkg.MultiPeriodDiscriminator(False).to(dev)
if os.environ.get("SKIP_S") == "1":
    d.discriminators = torch.nn.ModuleList(list(d.discriminators)[1:])
if os.environ.get("NP"):
    d.discriminators = torch.nn.ModuleList(list(d.discriminators)[:int(os.environ["NP"])])
if os.environ.get("ONLY_S") == "1":
    d.discriminators = torch.nn.ModuleList(list(d.discriminators)[:1])
b = 16
y = torch.rand(b, 1, 8192, device=dev) * 2 - 1
y_hat = (torch.rand(b, 1, 8192, device=dev) * 2 - 1).requires_grad_(True)


MODE = os.environ.get("FM", "plain")
def fm(a, c):
    if MODE == "contig":
        return (a.contiguous().float() - c.contiguous().float()).abs().mean()
    if MODE == "sum":
        return (a.float() - c.float()).abs().sum() / a.numel()
    if MODE == "flat":
        return (a.float() - c.float()).abs().reshape(-1).mean()
    if MODE == "twostage":
        ab = (a.float() - c.float()).abs()
        return ab.flatten(0, 1).sum(dim=(1, 2)).sum() / ab.numel()
    if MODE == "keep":
        af, cf = a.float(), c.float()
        df = af - cf
        ab = df.abs()
        m = ab.mean()
        KEEP.append((af, cf, df, ab, m))
        return m
    return (a.float() - c.float()).abs().mean()


KEEP = []
def run():
    KEEP.clear()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        if os.environ.get("INLINE") == "1":
            import importlib
            WA = importlib.import_module("personalized_text-to-speech_amd.weight_arena")
            dp = d.discriminators[0]
            with WA.scope(d, type(d)._arena_specs):
                out, fmap = dp(torch.cat([y, y_hat], 0))
            rs, gs, fr, fg = [out[:16]], [out[16:]], [[f[:16] for f in fmap]], [[f[16:] for f in fmap]]
        else:
            rs, gs, fr, fg = d(y, y_hat)
        terms = [((1 - r.float()) ** 2).mean() for r in rs] + [(g.float() ** 2).mean() for g in gs] + \
            [fm(a, c) for fa, fc in zip(fr, fg) for a, c in zip(fa, fc)]
        run.terms = terms
        loss = sum(terms)
    if os.environ.get('NOBWD') == '1':
        grads = [loss]
    else:
        grads = torch.autograd.grad(loss, [y_hat] + list(d.parameters()))
    outs = list(rs) + list(gs) + [f for fm in fr for f in fm] + [f for fm in fg for f in fm]
    run.outs = outs
    return loss, grads


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(int(os.environ.get('WARM', 3))):
        l0, g0 = run()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
ref = [g.clone() for g in g0]; lref = l0.item(); oref = [o.detach().clone() for o in run.outs]; tref = [t.item() for t in run.terms]; kref = [[q.clone() for q in tup] for tup in KEEP]
graph = torch.cuda.CUDAGraph()
if os.environ.get('DOT'):
    graph.enable_debug_mode()
with torch.cuda.graph(graph):
    l1, g1 = run()
if os.environ.get('DOT'):
    graph.debug_dump(os.environ['DOT'])
outs1 = run.outs; terms1 = run.terms; keep1 = list(KEEP)
names = ["y_hat"] + [k for k, _ in d.named_parameters()]
for it in range(3):
    # poison freed memory between replays
    if os.environ.get("NOJUNK") != "1":
        junk = torch.full((1 << 28,), float("nan"), device=dev); del junk
    graph.replay(); torch.cuda.synchronize()
    print("replay", it, "loss", l1.item(), "ref", lref)
    for i, (a, c) in enumerate(zip(tref, terms1)):
        if abs(a - c.item()) > 0.02 * abs(a) + 1e-6:
            print(f"   term[{i}] ref {a:.5f} got {c.item():.5f}")
    for i, (ra, rc) in enumerate(zip(kref, keep1)):
        es = [((u.float() - v.float()).norm() / (u.float().norm() + 1e-12)).item() for u, v in zip(ra, rc)]
        if max(es) > 1e-3:
            print(f"   keep[{i}] af,cf,df,ab,m rel errs {[f'{e:.2e}' for e in es]} shape {tuple(ra[0].shape)} strides {rc[0].stride()} ab strides {rc[3].stride()}")
    for i, (a, c) in enumerate(zip(oref, outs1)):
        e = ((a.float() - c.float()).norm() / (a.float().norm() + 1e-12)).item()
        if not (e < 0.02):
            print(f"   out[{i}] shape {tuple(a.shape)} strides {c.stride()} rel {e:.3e}")
    for n, a, c in zip(names, ref, g1):
        e = ((a.float() - c.float()).norm() / (a.float().norm() + 1e-12)).item()
        if not (e < 0.02):
            print(f"   {n:50s} rel {e:.3e}")
