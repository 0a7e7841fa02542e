#!/usr/bin/env python3
"""Generate tests/golden/data_pipeline.npz by RUNNING the reference's DistributedBucketSampler and TextAudioSpeakerCollate
(/root/reference/data_utils.py:115-276, imported read-only; `torchaudio` — not installed, used only by the file loader —
is stubbed in sys.modules).  Data only: lengths, sampler parameters, the batches of several (epoch, rank) pairs, and one
collated batch."""
import os, sys, types
import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.dont_write_bytecode = True
for name in ("torchaudio", "librosa", "librosa.util", "librosa.filters"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["librosa.util"].normalize = sys.modules["librosa.util"].pad_center = sys.modules["librosa.util"].tiny = None
sys.modules["librosa.filters"].mel = None
sys.modules["librosa"].util, sys.modules["librosa"].filters = sys.modules["librosa.util"], sys.modules["librosa.filters"]
text_stub = types.ModuleType("text"); text_stub.text_to_sequence = text_stub.cleaned_text_to_sequence = None
sys.modules.setdefault("text", text_stub)
sys.path.insert(0, "/root/reference")
import data_utils  # noqa: E402


class DS:
    def __init__(self, lengths): self.lengths = lengths
    def __len__(self): return len(self.lengths)


out = {}
rng = np.random.default_rng(5)
lengths = rng.integers(20, 1100, size=700).tolist()
boundaries = [32, 300, 400, 500, 600, 700, 800, 900, 1000]          # finetune_speaker_v2.py:77
out["lengths"] = np.array(lengths); out["boundaries"] = np.array(boundaries)
cases = []
for (bs, nrep, rank, epoch, shuffle) in [(16, 1, 0, 0, True), (16, 8, 3, 0, True), (4, 2, 1, 5, True), (8, 2, 0, 0, False)]:
    s = data_utils.DistributedBucketSampler(DS(list(lengths)), bs, list(boundaries), num_replicas=nrep, rank=rank, shuffle=shuffle)
    s.set_epoch(epoch)
    batches = list(iter(s))
    tag = f"s{len(cases)}"
    cases.append([bs, nrep, rank, epoch, int(shuffle), len(s)])
    out[tag + "/batches"] = np.array(batches, dtype=np.int64)
out["cases"] = np.array(cases)
# sparse lengths: empty buckets get removed
lengths2 = [50, 60, 650, 655, 660, 990, 40, 45]
s = data_utils.DistributedBucketSampler(DS(list(lengths2)), 2, list(boundaries), num_replicas=1, rank=0, shuffle=True)
out["sparse/lengths"] = np.array(lengths2); out["sparse/batches"] = np.array(list(iter(s)), dtype=np.int64)
out["sparse/boundaries_after"] = np.array(s.boundaries)
# collate
torch.manual_seed(3)
items = []
for i, (tx, ty) in enumerate([(7, 12), (11, 20), (5, 9), (9, 20)]):
    items.append((torch.randint(0, 30, (tx,)), torch.rand(6, ty), torch.rand(1, ty * 4) - 0.5, torch.LongTensor([i + 3])))
res = data_utils.TextAudioSpeakerCollate(return_ids=True)(items)
for i, it in enumerate(items):
    out[f"collate/in{i}/text"] = it[0].numpy(); out[f"collate/in{i}/spec"] = it[1].numpy(); out[f"collate/in{i}/wav"] = it[2].numpy(); out[f"collate/in{i}/sid"] = it[3].numpy()
for name, t in zip(["text", "text_len", "spec", "spec_len", "wav", "wav_len", "sid", "ids"], res):
    out["collate/out/" + name] = t.numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "data_pipeline.npz"), **out)
print("data_pipeline.npz", os.path.getsize(os.path.join(ROOT, "tests", "golden", "data_pipeline.npz")), "bytes", cases)
