#!/usr/bin/env python3
"""Two eager fine-tune steps of the bench workload (C2) and nothing else: the target of the rocprofv3 --pmc passes
(counter collection serialises every dispatch, so the run is kept as short as a steady-state step allows)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from importlib import import_module
P = import_module("personalized_text-to-speech_amd"); cfgs = import_module("personalized_text-to-speech_amd.configs"); tr = import_module("personalized_text-to-speech_amd.train")
cfg_name, batch_size, t_y_range = cfgs.WORKLOADS["C2"]
hps = cfgs.get(cfg_name)
ft = tr.FineTuner(hps, "cuda:0", amp=True)
batch = tr.synthetic_batch(hps, batch_size, t_y_range, "cuda:0")
for _ in range(2):
    out = ft.step(batch)
torch.cuda.synchronize()
print({k: round(float(v), 4) for k, v in out.items()})
