"""STFT-window + mel-filterbank transform, drop-in for the reference's Audio2Mel
(featuresynth/feature/feature.py:11-59): same constructor defaults, same registered buffers
(`mel_basis` (n_mel, n_fft/2+1), `window` (win_length,)), same output
log10(clamp(mel_basis @ |STFT(audio)|, 1e-5)) of shape (B, n_mel, frames).

The reference takes its filterbank from librosa.filters.mel (third-party, un-pinned; called at
feature.py:27-29 with librosa < 0.10 positional arguments, i.e. htk=False, norm='slaney').
`slaney_mel_basis` restates that published algorithm in numpy on the host, as librosa itself
does; the per-frame arithmetic (window, FFT, magnitude, mel projection, log10) runs in one
HIP kernel (csrc/audio2mel.hip).

The reference's dataset-side helpers in the same file (`audio`, `spectrogram`, the LMDB cache,
:62-85) are file I/O and out of scope of this build.
"""
import numpy as np
import torch
from torch import nn

from .._ops import prims as P


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log_part = min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log_part, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_basis(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """(n_mels, 1 + n_fft//2) float32 triangular filters on the Slaney mel scale, each scaled by
    2 / (f[i+2] - f[i]) (area normalisation)."""
    if fmax is None:
        fmax = float(sr) / 2
    nb = 1 + n_fft // 2
    fft_freqs = np.linspace(0.0, float(sr) / 2, nb)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(edges)
    ramps = edges[:, None] - fft_freqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0.0, np.minimum(lower, upper))
    weights *= (2.0 / (edges[2:n_mels + 2] - edges[:n_mels]))[:, None]
    return weights.astype(np.float32)


class Audio2Mel(nn.Module):
    def __init__(self, n_fft=1024, hop_length=256, win_length=1024, sampling_rate=22050,
                 n_mel_channels=80, mel_fmin=0.0, mel_fmax=None):
        super().__init__()
        if win_length != n_fft:
            raise NotImplementedError("Audio2Mel kernel assumes win_length == n_fft (the reference "
                                      "only ever uses 1024/1024)")
        n = torch.arange(win_length, dtype=torch.float64)
        window = (0.5 - 0.5 * torch.cos(2.0 * np.pi * n / win_length)).float()  # periodic hann
        mel_basis = torch.from_numpy(
            slaney_mel_basis(sampling_rate, n_fft, n_mel_channels, mel_fmin, mel_fmax))
        self.register_buffer("mel_basis", mel_basis)
        self.register_buffer("window", window)
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.win_length = win_length
        self.sampling_rate = sampling_rate
        self.n_mel_channels = n_mel_channels

    def forward(self, audio):
        if isinstance(audio, np.ndarray):
            audio = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32)).view(1, 1, -1)
            audio = audio.to(self.window.device)
        if audio.dim() != 3 or audio.shape[1] != 1:
            raise RuntimeError("Audio2Mel expects (B, 1, N) audio, got %s" % (tuple(audio.shape),))
        a = audio.reshape(audio.shape[0], audio.shape[2]).contiguous().float()
        return P.audio2mel(a, self.window, self.mel_basis, self.n_fft, self.hop_length)
