"""STFT-window + mel-filterbank transform, drop-in for the reference's Audio2Mel
(featuresynth/feature/feature.py:11-59): same constructor defaults, same registered buffers
(`mel_basis` (n_mel, n_fft/2+1), `window` (win_length,)), same output
log10(clamp(mel_basis @ |STFT(audio)|, 1e-5)) of shape (B, n_mel, frames).

The reference takes its filterbank from librosa.filters.mel (third-party, un-pinned; called at
feature.py:27-29 with librosa < 0.10 positional arguments, i.e. htk=False, norm='slaney').
`slaney_mel_basis` restates that published algorithm in numpy on the host, as librosa itself
does; the per-frame arithmetic (window, FFT, magnitude, mel projection, log10) runs in one
HIP kernel (csrc/audio2mel.hip).

The reference's dataset-side helpers in the same file (:62-85) are `audio` (read a file chunk with zounds,
librosa.resample to the target rate, librosa.util.normalize * 0.95) and `spectrogram` (Audio2Mel of that).
The file / LMDB part is out of scope; the arithmetic -- resampling and peak
normalisation -- is `audio_from_samples` below, on the device: librosa's default resampler is resampy's
'kaiser_best' interpolated Kaiser-windowed sinc, restated from its published algorithm (resampy and librosa are
third-party, un-pinned and absent here: parity unpinned, checked against a numpy restatement and against
scipy.signal.resample_poly).
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


# resampy 'kaiser_best' (librosa.resample's default res_type in librosa < 0.10): published filter parameters
KAISER_BEST = dict(num_zeros=64, precision=9, rolloff=0.9475937167399596, beta=14.769656459379492)


def sinc_window(num_zeros, precision, rolloff, beta):
    """Right half of the Kaiser-windowed sinc low-pass, 2**precision samples per zero crossing
    (resampy.filters.sinc_window) -> (interp_win float64, samples per zero crossing)."""
    from scipy.signal.windows import kaiser
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


_FILTERS = {}


def _filter_for(ratio, device):
    key = (round(float(ratio), 12), str(device))
    if key not in _FILTERS:
        win, num_table = sinc_window(**KAISER_BEST)
        if ratio < 1:
            win = win * ratio
        delta = np.zeros_like(win)
        delta[:-1] = np.diff(win)
        _FILTERS[key] = (torch.from_numpy(win.astype(np.float32)).to(device),
                         torch.from_numpy(delta.astype(np.float32)).to(device), num_table)
    return _FILTERS[key]


def resample(samples, orig_sr, target_sr):
    """librosa.resample(samples, orig_sr, target_sr) (res_type 'kaiser_best', fix=True, scale=False) on the
    device: (B, N) or (B, 1, N) float tensor -> same rank with ceil(N * target_sr / orig_sr) samples."""
    if orig_sr == target_sr:
        return samples
    ratio = float(target_sr) / float(orig_sr)
    shape = samples.shape
    x = samples.reshape(-1, shape[-1]).contiguous().float()
    win, delta, num_table = _filter_for(ratio, x.device)
    y = P.resample_sinc(x, ratio, win, delta, num_table)
    return y.reshape(shape[:-1] + (y.shape[-1],))


def audio_from_samples(samples, orig_sr, samplerate):
    """The arithmetic of the reference's `audio()` (feature.py:64-71) after the file read: resample to
    `samplerate`, then librosa.util.normalize(axis=-1) * 0.95 (peak 0.95 per chunk)."""
    y = resample(samples, orig_sr, samplerate)
    shape = y.shape
    y2 = y.reshape(-1, shape[-1]).contiguous().clone() if y is samples else y.reshape(-1, shape[-1])
    P.peak_normalize_(y2, 0.95)
    return y2.reshape(shape)
