"""Debug: MultiPeriodDiscriminator at bench size, HIP period discriminators vs the library path (grads per parameter)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ptts_amd as pkg

torch.manual_seed(0)
dev = "cuda"
d = pkg.MultiPeriodDiscriminator(False).to(dev)
b = int(os.environ.get("B", 16))
y, y_hat = torch.rand(b, 1, 8192, device=dev) * 2 - 1, (torch.rand(b, 1, 8192, device=dev) * 2 - 1).requires_grad_(True)
res = {}
for mode in (False, True):
    pkg.models.DiscriminatorP.use_hip = mode
    d.__dict__.pop("_weight_arenas", None)
    for p in d.parameters():
        p.grad = None
    y_hat.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        rs, gs, fr, fg = d(y, y_hat)
        loss = sum(((1 - r.float()) ** 2).mean() + (g.float() ** 2).mean() for r, g in zip(rs, gs)) + \
            sum((a.float() - c.float()).abs().mean() for fa, fc in zip(fr, fg) for a, c in zip(fa, fc))
    loss.backward()
    res[mode] = ({k: p.grad.clone() for k, p in d.named_parameters()}, y_hat.grad.clone(), loss.item())
print("loss", res[False][2], res[True][2])
ga, gb = res[False][0], res[True][0]
for k in ga:
    a, c = ga[k].float(), gb[k].float()
    e = ((a - c).norm() / (a.norm() + 1e-12)).item()
    bad = not torch.isfinite(c).all().item()
    if e > 0.05 or bad:
        print(f"{k:50s} rel {e:.3e} finite={not bad} |lib|={a.norm().item():.3e} |hip|={c.norm().item():.3e}")
a, c = res[False][1], res[True][1]
print("d y_hat rel", ((a - c).norm() / a.norm()).item(), torch.isfinite(c).all().item())
