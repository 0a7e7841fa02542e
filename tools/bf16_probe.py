import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import torch, ptts_amd as P
from model_util import *
from oracle import vits_torch as O
from importlib import import_module
DEV="cuda:0"
def run(net, cfgm, z, g):
    dec=net.dec
    sd={("dec."+k):v.detach().clone().requires_grad_(True) for k,v in dec.state_dict().items()}
    out={}
    for mode in ("fp32","autocast"):
        for v in sd.values(): v.grad=None
        z_o,g_o=z.clone().requires_grad_(True),g.clone().requires_grad_(True)
        if mode=="autocast":
            with torch.autocast("cuda",dtype=torch.bfloat16): y=O.generator(sd,cfgm,z_o,g_o)
        else: y=O.generator(sd,cfgm,z_o,g_o)
        torch.manual_seed(5); probe=torch.randn_like(y.float())
        (y.float()*probe).sum().backward()
        out[mode]=(y.float().detach(), z_o.grad.clone(), g_o.grad.clone(), {k:v.grad.clone() for k,v in sd.items()})
    z_p,g_p=z.clone().requires_grad_(True),g.clone().requires_grad_(True)
    dec.zero_grad()
    with torch.autocast("cuda",dtype=torch.bfloat16): y=dec(z_p,g_p)
    (y*probe).sum().backward()
    out["mine"]=(y.detach(), z_p.grad, g_p.grad, {"dec."+k:p.grad for k,p in dec.named_parameters()})
    ref=out["fp32"]
    for m in ("autocast","mine"):
        o=out[m]
        ws=max(rel_err(o[3][k],ref[3][k]) for k in ref[3])
        print(m, "y",rel_err(o[0],ref[0]),"dz",rel_err(o[1],ref[1]),"dg",rel_err(o[2],ref[2]),"worst dW",ws)
g_,cfg=load_tiny(); net=build_tiny(P,g_,cfg,DEV)
torch.manual_seed(0)
run(net,cfg["model"],torch.randn(2,16,13,device=DEV),torch.randn(2,8,1,device=DEV))
cfgs=import_module("personalized_text-to-speech_amd.configs"); hps=cfgs.get("finetune_speaker")
torch.manual_seed(1)
net=P.SynthesizerTrn(hps.n_symbols,513,32,n_speakers=4,**hps.model).to(DEV)
run(net,dict(hps.model),torch.randn(2,192,6,device=DEV),torch.randn(2,256,1,device=DEV))
