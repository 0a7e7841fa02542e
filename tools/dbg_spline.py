import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch, importlib
import ptts_amd
R = importlib.import_module("personalized_text-to-speech_amd.rowops")
ops = np.load("tests/golden/ops.npz"); DEV="cuda:0"
for tag, inv in (("spline_fwd/", False), ("spline_inv/", True)):
    x = torch.from_numpy(ops[tag+"x"]).to(DEV); n = x.numel()
    h = torch.cat([torch.from_numpy(ops[tag+k]) for k in ("uw","uh","ud")], -1).to(DEV)
    y, lad = R.rq_spline(x.reshape(n), h.reshape(n,29), 1.0, inv, 5.0)
    yr = torch.from_numpy(ops[tag+"y"]).reshape(n).to(DEV); lr = torch.from_numpy(ops[tag+"lad"]).reshape(n).to(DEV)
    ey = (y-yr).abs(); el=(lad-lr).abs()
    i = int(ey.argmax()); j=int(el.argmax())
    print(tag, "max|dy|", float(ey.max()), "at x=", float(x.reshape(n)[i]), "y", float(y[i]), "ref", float(yr[i]), "| max|dlad|", float(el.max()), "x=", float(x.reshape(n)[j]), float(lad[j]), float(lr[j]))
