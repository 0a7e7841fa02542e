"""Mirror of the reference's mel_processing.py (spectrogram_torch :51-70, spec_to_mel_torch :73-82,
mel_spectrogram_torch :85-112): same names, arguments and results.

The mel filterbank comes from third-party librosa==0.9.2 in the reference
(`librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)`, mel_processing.py:78,96), which is neither
vendored there nor installed here: `mel_filterbank` restates librosa's published defaults (Slaney
mel scale, htk=False, norm='slaney', float32).  No reference fixture covers it — PARITY UNPINNED
for the filterbank values; the STFT magnitude and the log-clamp around it are pinned
(tests/golden/ops.npz `stft/*`).
"""
import numpy as np
import torch

from . import kernels

MAX_WAV_VALUE = 32768.0

mel_basis = {}
hann_window = {}


def dynamic_range_compression_torch(x, C=1, clip_val=1e-5):
    return torch.log(torch.clamp(x, min=clip_val) * C)


def dynamic_range_decompression_torch(x, C=1):
    return torch.exp(x) / C


def spectral_normalize_torch(magnitudes):
    return dynamic_range_compression_torch(magnitudes)


def spectral_de_normalize_torch(magnitudes):
    return dynamic_range_decompression_torch(magnitudes)


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """[n_mels, 1 + n_fft//2] float32 triangular filters, Slaney scale and area normalisation."""
    if fmax is None:
        fmax = float(sr) / 2
    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0, float(sr) / 2, n_bins)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, n_bins), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return weights.astype(np.float32)


def _window(win_size, dtype, device):
    key = f"{win_size}_{dtype}_{device}"
    if key not in hann_window:
        hann_window[key] = torch.hann_window(win_size).to(dtype=dtype, device=device)
    return hann_window[key]


def _basis(sampling_rate, n_fft, num_mels, fmin, fmax, dtype, device):
    key = f"{fmax}_{dtype}_{device}_{sampling_rate}_{n_fft}_{num_mels}_{fmin}"
    if key not in mel_basis:
        mel_basis[key] = torch.from_numpy(mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax)).to(dtype=dtype, device=device)
    return mel_basis[key]


def spectrogram_torch(y, n_fft, sampling_rate, hop_size, win_size, center=False, prepadded=False):
    """y [b, t] in [-1, 1] -> linear magnitude [b, n_fft/2+1, frames]; reflect pad (n_fft-hop)/2,
    periodic Hann, sqrt(re^2 + im^2 + 1e-6) (reference mel_processing.py:51-70).  The reference's
    range check prints (two host syncs per call, :52-55) are dropped.  prepadded: `y` already carries the padding
    (data_utils.spectrograms_on_device pads every item of a batch at its own ends)."""
    if center:
        raise NotImplementedError("the reference always calls with center=False")
    return kernels.stft_magnitude(y, n_fft, hop_size, win_size, _window(win_size, y.dtype, y.device), prepadded=prepadded)


def spec_to_mel_torch(spec, n_fft, num_mels, sampling_rate, fmin, fmax):
    # reference mel_processing.py:73-82: mel basis @ linear-magnitude spec, then log(clamp(., 1e-5))
    return _mel(spec, _basis(sampling_rate, n_fft, num_mels, fmin, fmax, spec.dtype, spec.device))


def _mel(spec, basis):
    """log(clamp(basis @ spec, 1e-5)) in the dtype of `spec` even inside an autocast region: the c_mel = 45 loss is taken on
    this product, and the reference's mel path is fp32 (mel_processing.py:104 `y.float()`)."""
    if spec.is_cuda and torch.is_autocast_enabled():
        with torch.autocast("cuda", enabled=False):
            return spectral_normalize_torch(torch.matmul(basis, spec))
    return spectral_normalize_torch(torch.matmul(basis, spec))


def mel_spectrogram_torch(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
    # reference mel_processing.py:85-112 (computes in fp32: `y.float()` at :104)
    y = y.float()
    if y.is_cuda and not center:
        # DFT product + one fused magnitude / mel filter bank / log-clamp launch (csrc/stft_mel.hip), fp32 like the reference
        basis = _basis(sampling_rate, n_fft, num_mels, fmin, fmax, y.dtype, y.device)
        with torch.autocast("cuda", enabled=False):
            mel = kernels.stft_mel(y, n_fft, hop_size, win_size, _window(win_size, y.dtype, y.device), basis)
        if mel is not None:
            return mel
    spec = spectrogram_torch(y, n_fft, sampling_rate, hop_size, win_size, center)
    return _mel(spec, _basis(sampling_rate, n_fft, num_mels, fmin, fmax, spec.dtype, spec.device))
