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


# ---------------------------------------------------------------------------------------------------------------------------
# The dataset (reference data_utils.py:16-112) against tests/golden/misc.npz `loader/*`: produced by running the reference's
# TextAudioSpeakerLoader over a synthetic file list with torchaudio.load replaced by a raw 16-bit reader (tools/gen_golden_misc.py).
def _misc():
    return np.load(os.path.join(ROOT, "tests", "golden", "misc.npz"))


def _raw_reader(filename):
    a = np.fromfile(filename, dtype=np.int16).astype(np.float32) / 32768.0
    return torch.from_numpy(a).unsqueeze(0), 22050


def _build_dataset(D, g, d):
    import json
    import types
    texts = json.loads(bytes(g["loader/texts"]).decode()); symbols = json.loads(bytes(g["loader/symbols"]).decode())
    sizes = g["loader/sizes"].tolist(); kept = g["loader/kept"].tolist()
    nfft, hop = g["loader/hp"].tolist()
    lines = []
    for i, (txt, size) in enumerate(zip(texts, sizes)):
        fn = os.path.join(d, f"u{i}.raw")
        if i in kept[:3]:
            g[f"loader/item{kept.index(i)}/pcm"].tofile(fn)
        else:
            open(fn, "wb").write(b"\0" * size)                  # only the SIZE of the other files is looked at
        assert os.path.getsize(fn) == size
        lines.append(f"{fn}|{i % 4}|{txt}")
    lst = os.path.join(d, "list.txt")
    open(lst, "w", encoding="utf-8").write("\n".join(lines) + "\n")
    hp = types.SimpleNamespace(text_cleaners=["none"], max_wav_value=32768.0, sampling_rate=22050, filter_length=nfft, hop_length=hop,
                               win_length=nfft, cleaned_text=True, add_blank=True)
    return D.TextAudioSpeakerLoader(lst, hp, symbols, audio_reader=_raw_reader), hp


def test_dataset_filter_order_lengths_and_items_match_reference(pkg, tmp_path):
    from importlib import import_module
    D = import_module("personalized_text-to-speech_amd.data_utils")
    g = _misc()
    ds, hp = _build_dataset(D, g, str(tmp_path))
    kept = [int(os.path.basename(a[0])[1:-4]) for a in ds.audiopaths_sid_text]
    assert kept == g["loader/kept"].tolist()                     # seeded shuffle order + min/max text length filter (0, 191, 250 dropped)
    assert ds.lengths == g["loader/lengths"].tolist()            # file size // (2 * hop)
    items = [ds[j] for j in range(3)]
    for j, (txt, spec, wav, sid) in enumerate(items):
        assert torch.equal(txt, torch.from_numpy(g[f"loader/item{j}/text"]))           # symbol lookup + interspersed blanks
        assert torch.equal(wav, torch.from_numpy(g[f"loader/item{j}/wav"])) and int(sid) == int(g[f"loader/item{j}/sid"])
        ref = torch.from_numpy(g[f"loader/item{j}/spec"])
        assert spec.shape == ref.shape and float((spec - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    res = D.TextAudioSpeakerCollate()(items)
    for name, t in zip(["text", "text_len", "spec", "spec_len", "wav", "wav_len", "sid"], res):
        ref = torch.from_numpy(g["loader/collate/" + name])
        assert t.shape == ref.shape and float((t.double() - ref.double()).abs().max()) <= 1e-4, name


def test_uncleaned_text_needs_an_injected_front_end(pkg, tmp_path):
    from importlib import import_module
    D = import_module("personalized_text-to-speech_amd.data_utils")
    ds, hp = _build_dataset(D, _misc(), str(tmp_path))
    ds.cleaned_text = False
    try:
        ds.get_text("abc")
        raise AssertionError("expected NotImplementedError")
    except NotImplementedError:
        pass
    ds.text_to_sequence = lambda text, cleaners: [3, 4]
    assert ds.get_text("abc").tolist() == [0, 3, 0, 4, 0]


def test_wav_reader_pcm16(pkg, tmp_path):
    import wave
    from importlib import import_module
    D = import_module("personalized_text-to-speech_amd.data_utils")
    pcm = (np.sin(np.arange(500) / 7.0) * 20000).astype(np.int16)
    fn = str(tmp_path / "a.wav")
    with wave.open(fn, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(22050); w.writeframes(pcm.tobytes())
    audio, sr = D.read_wav_pcm16(fn)
    assert sr == 22050 and audio.shape == (1, 500) and torch.equal(audio[0], torch.from_numpy(pcm.astype(np.float32) / 32768.0))


def test_batched_spectrograms_pad_every_item_at_its_own_end(pkg):
    """spectrograms_on_device on the padded batch == the reference's per-file spectrograms (fixture items of different lengths),
    including the last frames of the shorter items, which see reflected samples, not the batch's zero padding.  (Host tensors:
    the torch.stft branch of the same function; tests/test_model_gpu.py runs the GPU branch.)"""
    import types
    from importlib import import_module
    D = import_module("personalized_text-to-speech_amd.data_utils")
    g = _misc()
    nfft, hop = g["loader/hp"].tolist()
    wav, wav_len, spec, spec_len = (torch.from_numpy(g["loader/collate/" + k]) for k in ("wav", "wav_len", "spec", "spec_len"))
    hps = types.SimpleNamespace(data=types.SimpleNamespace(filter_length=nfft, hop_length=hop, win_length=nfft, sampling_rate=22050))
    got, got_len = D.spectrograms_on_device(wav, wav_len, hps)
    assert torch.equal(got_len, spec_len)
    t = spec.size(2)
    assert float((got[:, :, :t] - spec).abs().max()) <= 1e-4 * float(spec.abs().max())
    assert float(got[:, :, t:].abs().max()) == 0.0 if got.size(2) > t else True
