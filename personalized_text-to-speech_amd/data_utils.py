"""Input pipeline around the hot path (reference data_utils.py:16-276): the dataset with the
reference's filter / length-estimate / per-file spectrogram semantics, the bucket sampler and the collate function with the
reference's exact batches, and the spectrograms of a batch of waveforms on the GPU (the reference computes them per file on
DataLoader workers, data_utils.py:60-75, which starves a step that takes tens of milliseconds).

Two things the reference's loader pulls in are not installable offline and are therefore INJECTED instead of imported:
`torchaudio.load` (`audio_reader`, default: a 16-bit PCM WAV reader on the standard library) and the phonemizing text
front-end `text.text_to_sequence` (`text_to_sequence=`, no default).  File lists with CLEANED text
(`cleaned_text: true`, what preprocess_v2.py writes) need neither: `cleaned_text_to_sequence` is the symbol-table lookup.
"""
import os
import random

import numpy as np
import torch

from . import commons
from .mel_processing import spectrogram_torch


def load_filepaths_and_text(filename, split="|"):
    """reference utils.py:290-293"""
    with open(filename, encoding="utf-8") as f:
        return [line.strip().split(split) for line in f]


def cleaned_text_to_sequence(cleaned_text, symbols):
    """reference text/__init__.py:34-43: ids of the characters that are symbols, the others are dropped."""
    symbol_to_id = {s: i for i, s in enumerate(symbols)}
    return [symbol_to_id[ch] for ch in cleaned_text if ch in symbol_to_id]


def read_wav_pcm16(filename):
    """(audio [channels, n] float32 in [-1, 1), sampling rate) of a 16-bit PCM WAV file — what the reference's
    `torchaudio.load(filename, normalize=True, channels_first=True)` (data_utils.py:78) returns for such a file."""
    import wave
    with wave.open(filename, "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError(f"{filename}: {8 * w.getsampwidth()}-bit samples (16-bit PCM expected)")
        ch, sr = w.getnchannels(), w.getframerate()
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    audio = torch.from_numpy(pcm.astype(np.float32) / 32768.0).view(-1, ch).t().contiguous()
    return audio, sr


class TextAudioSpeakerLoader(torch.utils.data.Dataset):
    """(text ids, linear spectrogram [F, t], waveform [1, n], speaker id) per line `path|speaker|text` of a file list, with the
    reference's semantics (data_utils.py:16-112): the list is shuffled with `random.seed(1234)`; an entry survives when
    min_text_len <= len(text) <= max_text_len (defaults 1, 190; the length of the RAW text field); `lengths` estimates the frame
    count from the FILE SIZE as size // (2 * hop_length) (16-bit mono payload, header included) for the bucket sampler; the
    spectrogram is taken per file, reflect-padded at that file's own ends."""

    def __init__(self, audiopaths_sid_text, hparams, symbols, audio_reader=None, text_to_sequence=None):
        self.audiopaths_sid_text = load_filepaths_and_text(audiopaths_sid_text)
        self.text_cleaners = hparams.text_cleaners
        self.max_wav_value = hparams.max_wav_value
        self.sampling_rate = hparams.sampling_rate
        self.filter_length, self.hop_length, self.win_length = hparams.filter_length, hparams.hop_length, hparams.win_length
        self.cleaned_text = getattr(hparams, "cleaned_text", False)
        self.add_blank = hparams.add_blank
        self.min_text_len = getattr(hparams, "min_text_len", 1)
        self.max_text_len = getattr(hparams, "max_text_len", 190)
        self.symbols = symbols
        self.audio_reader = audio_reader or read_wav_pcm16
        self.text_to_sequence = text_to_sequence
        random.seed(1234)
        random.shuffle(self.audiopaths_sid_text)
        self._filter()

    def _filter(self):
        kept, lengths = [], []
        for audiopath, sid, text in self.audiopaths_sid_text:
            if self.min_text_len <= len(text) <= self.max_text_len:
                kept.append([audiopath, sid, text])
                lengths.append(os.path.getsize(audiopath) // (2 * self.hop_length))
        self.audiopaths_sid_text, self.lengths = kept, lengths

    def get_audio(self, filename):
        audio_norm, _ = self.audio_reader(filename)
        spec = spectrogram_torch(audio_norm, self.filter_length, self.sampling_rate, self.hop_length, self.win_length, center=False)
        return spec.squeeze(0), audio_norm

    def get_text(self, text):
        if self.cleaned_text:
            ids = cleaned_text_to_sequence(text, self.symbols)
        elif self.text_to_sequence is not None:
            ids = self.text_to_sequence(text, self.text_cleaners)
        else:
            raise NotImplementedError("uncleaned text needs the phonemizing front-end (reference text/cleaners.py: pypinyin, jieba, "
                                      "pyopenjtalk ...), which is injected as text_to_sequence=; file lists written by the "
                                      "reference's preprocess_v2.py are already cleaned (cleaned_text: true)")
        if self.add_blank:
            ids = commons.intersperse(ids, 0)
        return torch.LongTensor(ids)

    def get_sid(self, sid):
        return torch.LongTensor([int(sid)])

    def get_audio_text_speaker_pair(self, audiopath_sid_text):
        audiopath, sid, text = audiopath_sid_text[0], audiopath_sid_text[1], audiopath_sid_text[2]
        text = self.get_text(text)
        spec, wav = self.get_audio(audiopath)
        return (text, spec, wav, self.get_sid(sid))

    def __getitem__(self, index):
        return self.get_audio_text_speaker_pair(self.audiopaths_sid_text[index])

    def __len__(self):
        return len(self.audiopaths_sid_text)


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
    """Linear spectrograms of a zero-padded waveform batch [b, 1, n] in ONE launch chain on the GPU (the STFT-as-convolution
    kernel of mel_processing.spectrogram_torch) instead of per file on CPU workers (reference data_utils.py:71-92: `spectrogram_torch`
    per item).  Every item is reflect-padded at ITS OWN ends — a gather builds the padded batch from the per-item lengths —
    so the frames of a short item equal what the reference computes from that file alone; frames beyond an item's length are
    zero.  Returns (spec [b, F, t], spec_lengths = wav_lengths // hop)."""
    dev = device or wav_padded.device
    n_fft, hop = hps.data.filter_length, hps.data.hop_length
    pad = int((n_fft - hop) / 2)
    wav = wav_padded.to(dev).squeeze(1).float()
    n = wav_lengths.to(dev).long()
    if int(n.min()) <= pad:
        raise ValueError("spectrograms_on_device: an item is shorter than the reflect padding")
    j = torch.arange(wav.size(1) + 2 * pad, device=dev)[None, :] - pad                 # source index before reflection
    src = torch.where(j < 0, -j, j)
    src = torch.where(src >= n[:, None], 2 * (n[:, None] - 1) - src, src)
    live = (j < (n[:, None] + pad)) & (src >= 0)
    yp = torch.gather(wav, 1, src.clamp(0, wav.size(1) - 1)) * live
    spec = spectrogram_torch(yp, n_fft, hps.data.sampling_rate, hop, hps.data.win_length, prepadded=True)
    spec_lengths = (n // hop).clamp(max=spec.size(2))
    frame = torch.arange(spec.size(2), device=dev)[None, :] < spec_lengths[:, None]
    return spec * frame[:, None, :], spec_lengths
