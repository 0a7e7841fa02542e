"""Debug: are torch global reductions graph-replay safe on this stack when pool blocks are recycled dirty? (no repo kernels)"""
import torch
dev = "cuda"
torch.manual_seed(0)
xs = [torch.rand(16, c, t, 2, device=dev).permute(0, 1, 2, 3) for c, t in [(32, 1366), (128, 456), (512, 152), (1024, 51), (1024, 51), (1, 51)]]
dirty = int(__import__("os").environ.get("DIRTY", "1"))


def run():
    out = []
    for x in xs:
        if dirty:
            for n in (8, 64, 512, 4096):
                tmp = torch.full((n,), 7, device=dev, dtype=torch.int32); del tmp       # dirty small blocks
        a = x.float() * 1.0
        out.append(a.abs().mean())
        out.append((a - 0.5).abs().sum())
    return out


s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        r = run()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
ref = [q.item() for q in r]
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    r = run()
for it in range(4):
    g.replay(); torch.cuda.synchronize()
    bad = [(i, ref[i], r[i].item()) for i in range(len(ref)) if abs(ref[i] - r[i].item()) > 1e-3 * abs(ref[i])]
    print("replay", it, "bad:", bad[:6])
