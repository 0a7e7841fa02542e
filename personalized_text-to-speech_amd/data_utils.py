"""Input pipeline around the hot path (reference data_utils.py:115-276): the bucket sampler and the
collate function with the reference's exact semantics, and the spectrogram of a batch of waveforms on the GPU (the reference
computes it per file on DataLoader workers, data_utils.py:60-75, which starves a step that takes tens of milliseconds).

File parsing (`TextAudioSpeakerLoader`: torchaudio + the text front-end, data_utils.py:16-112) is out of scope (SURVEY §8(f)
f-4: its dependencies are not installable offline); anything that yields (text ids, spectrogram, waveform, speaker id)
tuples can feed the collate below.
"""
import numpy as np
import torch

from .mel_processing import spectrogram_torch


class TextAudioSpeakerCollate:
    """Zero-pads a list of (text [t_x] long, spec [F, t_y] float, wav [1, n] float, sid) to the batch maxima, items sorted by
    DECREASING spectrogram length (reference data_utils.py:115-167; same return tuple, `return_ids` adds the permutation)."""

    def __init__(self, return_ids=False):
        self.return_ids = return_ids

    def __call__(self, batch):
        spec_len = torch.tensor([x[1].size(1) for x in batch], dtype=torch.long)
        _, order = torch.sort(spec_len, dim=0, descending=True)
        n = len(batch)
        text_padded = torch.zeros(n, max(len(x[0]) for x in batch), dtype=torch.long)
        spec_padded = torch.zeros(n, batch[0][1].size(0), int(spec_len.max()), dtype=torch.float32)
        wav_padded = torch.zeros(n, 1, max(x[2].size(1) for x in batch), dtype=torch.float32)
        text_lengths, spec_lengths, wav_lengths, sid = (torch.zeros(n, dtype=torch.long) for _ in range(4))
        for i, src in enumerate(order.tolist()):
            text, spec, wav, s = batch[src]
            text_padded[i, :text.size(0)] = text
            spec_padded[i, :, :spec.size(1)] = spec
            wav_padded[i, :, :wav.size(1)] = wav
            text_lengths[i], spec_lengths[i], wav_lengths[i], sid[i] = text.size(0), spec.size(1), wav.size(1), int(s)
        out = (text_padded, text_lengths, spec_padded, spec_lengths, wav_padded, wav_lengths, sid)
        return out + (order,) if self.return_ids else out


class DistributedBucketSampler(torch.utils.data.Sampler):
    """Batches of similar spectrogram length (reference data_utils.py:170-276).  Item i falls into bucket j when
    boundaries[j] < lengths[i] <= boundaries[j+1]; items outside every bucket are dropped; empty buckets are removed; every
    bucket is padded (by repeating its permuted ids) to a multiple of num_replicas * batch_size; rank r takes ids[r::num_replicas];
    the per-bucket permutations and the order of the batches come from one torch.Generator seeded with the epoch — the same
    draws in the same order as the reference, so both produce the same batches."""

    def __init__(self, lengths, batch_size, boundaries, num_replicas=1, rank=0, shuffle=True):
        self.lengths = list(lengths)
        self.batch_size, self.num_replicas, self.rank, self.shuffle, self.epoch = batch_size, num_replicas, rank, shuffle, 0
        edges = np.asarray(boundaries)
        which = np.searchsorted(edges, np.asarray(self.lengths), side="left") - 1        # edges[j] < x <= edges[j+1]
        buckets = [[] for _ in range(len(edges) - 1)]
        for i, j in enumerate(which.tolist()):
            if 0 <= j < len(buckets):
                buckets[j].append(i)
        # an empty bucket j disappears together with its UPPER edge boundaries[j+1] (reference data_utils.py:197-210)
        self.boundaries = [boundaries[0]] + [boundaries[j + 1] for j, b in enumerate(buckets) if b]
        self.buckets = [b for b in buckets if b]
        total = num_replicas * batch_size
        self.num_samples_per_bucket = [len(b) + (total - len(b) % total) % total for b in self.buckets]
        self.total_size = sum(self.num_samples_per_bucket)
        self.num_samples = self.total_size // num_replicas

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)
        perms = [torch.randperm(len(b), generator=g).tolist() if self.shuffle else list(range(len(b))) for b in self.buckets]
        batches = []
        for bucket, ids, padded in zip(self.buckets, perms, self.num_samples_per_bucket):
            rem = padded - len(bucket)
            ids = ids + ids * (rem // len(bucket)) + ids[:rem % len(bucket)]
            ids = ids[self.rank::self.num_replicas]
            batches += [[bucket[i] for i in ids[j:j + self.batch_size]] for j in range(0, len(ids) - self.batch_size + 1, self.batch_size)]
        if self.shuffle:
            batches = [batches[i] for i in torch.randperm(len(batches), generator=g).tolist()]
        assert len(batches) * self.batch_size == self.num_samples
        self.batches = batches
        return iter(batches)

    def __len__(self):
        return self.num_samples // self.batch_size


def spectrograms_on_device(wav_padded, wav_lengths, hps, device=None):
    """Linear spectrograms of a padded waveform batch [b, 1, n] in ONE launch chain on the GPU (the STFT-as-convolution kernel of
    mel_processing.spectrogram_torch) instead of per file on CPU workers (reference data_utils.py:60-75: `spectrogram_torch` per
    item, cached as .spec.pt).  Frames beyond an item's length are zeroed; returns (spec [b, F, t], spec_lengths)."""
    dev = device or wav_padded.device
    hop = hps.data.hop_length
    wav = wav_padded.to(dev).squeeze(1)
    spec = spectrogram_torch(wav, hps.data.filter_length, hps.data.sampling_rate, hop, hps.data.win_length)
    spec_lengths = (wav_lengths.to(dev) // hop).clamp(max=spec.size(2))
    frame = torch.arange(spec.size(2), device=dev)[None, :] < spec_lengths[:, None]
    return spec * frame[:, None, :], spec_lengths
