"""CPU: bucket sampler and collate against tests/golden/data_pipeline.npz, produced by running the
reference's own DistributedBucketSampler / TextAudioSpeakerCollate (tools/gen_golden_data.py): identical batches for several
(batch size, replicas, rank, epoch, shuffle) settings, bucket removal on sparse data, identical padded tensors."""
import os

import numpy as np
import torch

from conftest import ROOT


def _g():
    return np.load(os.path.join(ROOT, "tests", "golden", "data_pipeline.npz"))


def test_bucket_sampler_matches_reference(pkg):
    from importlib import import_module
    D = import_module("personalized_text-to-speech_amd.data_utils")
    g = _g()
    for i, (bs, nrep, rank, epoch, shuffle, n_batches) in enumerate(g["cases"].tolist()):
        s = D.DistributedBucketSampler(g["lengths"].tolist(), bs, g["boundaries"].tolist(), num_replicas=nrep, rank=rank, shuffle=bool(shuffle))
        s.set_epoch(epoch)
        got = np.array(list(iter(s)), dtype=np.int64)
        assert len(s) == n_batches and np.array_equal(got, g[f"s{i}/batches"]), i
    s = D.DistributedBucketSampler(g["sparse/lengths"].tolist(), 2, g["boundaries"].tolist())
    assert np.array_equal(np.array(list(iter(s)), dtype=np.int64), g["sparse/batches"])
    assert s.boundaries == g["sparse/boundaries_after"].tolist()


def test_ranks_partition_every_padded_bucket(pkg):
    from importlib import import_module
    D = import_module("personalized_text-to-speech_amd.data_utils")
    g = _g()
    seen = []
    for rank in range(4):
        s = D.DistributedBucketSampler(g["lengths"].tolist(), 8, g["boundaries"].tolist(), num_replicas=4, rank=rank)
        seen.append([i for b in iter(s) for i in b])
    assert len({len(x) for x in seen}) == 1                       # same number of samples on every rank
    lens = g["lengths"]
    inside = {i for i in range(len(lens)) if 32 < lens[i] <= 1000}
    assert set().union(*map(set, seen)) == inside                 # every in-range item is visited, none outside


def test_collate_matches_reference(pkg):
    from importlib import import_module
    D = import_module("personalized_text-to-speech_amd.data_utils")
    g = _g()
    items = [(torch.from_numpy(g[f"collate/in{i}/text"]), torch.from_numpy(g[f"collate/in{i}/spec"]), torch.from_numpy(g[f"collate/in{i}/wav"]),
              torch.from_numpy(g[f"collate/in{i}/sid"])) for i in range(4)]
    res = D.TextAudioSpeakerCollate(return_ids=True)(items)
    for name, t in zip(["text", "text_len", "spec", "spec_len", "wav", "wav_len", "sid", "ids"], res):
        assert np.array_equal(t.numpy(), g["collate/out/" + name]), name
