"""Log-mel front end for reference audio (host side, NumPy): the step in front of the speaker encoder.

Restates ``get_mel_from_wav`` (mels.py:357-395): reflect-pad (n_fft-hop)/2, STFT n_fft/hop/hann with
center=False, magnitude, Slaney mel basis (librosa.filters.mel defaults: htk=False, norm='slaney'),
log(clip(., 1e-5)).  librosa is not installed in the build image, so parity of this file against
librosa is UNPINNED; the formulas follow librosa's published definitions.
"""
import numpy as np


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    fftfreqs = np.linspace(0, sr / 2.0, n_fft // 2 + 1)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]        # slaney area normalisation
    return w.astype(np.float32)


_basis_cache = {}


def get_mel_from_wav(audio, sampling_rate, fft_size, hop_size, win_length, num_mels, fmin, fmax):
    """-> (log-mel [num_mels, frames], energy [frames])."""
    audio = np.asarray(audio, np.float32)
    key = (sampling_rate, fft_size, num_mels, fmin, fmax)
    if key not in _basis_cache:
        _basis_cache[key] = mel_filterbank(sampling_rate, fft_size, num_mels, fmin, fmax)
    basis = _basis_cache[key]
    pad = (fft_size - hop_size) // 2
    x = np.pad(audio, (pad, pad), mode="reflect")
    n = 1 + (len(x) - fft_size) // hop_size
    idx = np.arange(fft_size)[None, :] + hop_size * np.arange(n)[:, None]
    win = np.hanning(win_length + 1)[:-1].astype(np.float32)            # periodic hann (fftbins=True)
    if win_length < fft_size:
        lp = (fft_size - win_length) // 2
        win = np.pad(win, (lp, fft_size - win_length - lp))
    mag = np.abs(np.fft.rfft(x[idx] * win[None, :], axis=1)).T.astype(np.float32)   # [n_fft/2+1, frames]
    spec = np.log(np.clip(basis @ mag, 1e-5, None))
    return spec.astype(np.float32), np.linalg.norm(mag, axis=0)


def trim_silence(wav, top_db=40, frame_length=2048, hop_length=512):
    """librosa.effects.trim restated: keep [first, last] frame whose RMS is within top_db of the peak."""
    wav = np.asarray(wav, np.float32)
    if len(wav) < frame_length:
        return wav
    pad = frame_length // 2
    x = np.pad(wav, (pad, pad), mode="constant")
    n = 1 + (len(x) - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n)[:, None]
    rms = np.sqrt(np.mean(x[idx] ** 2, axis=1))
    db = 20 * np.log10(np.maximum(rms, 1e-10)) - 20 * np.log10(max(rms.max(), 1e-10))
    nz = np.flatnonzero(db > -top_db)
    if nz.size == 0:
        return wav[:0]
    return wav[nz[0] * hop_length: min(len(wav), (nz[-1] + 1) * hop_length)]
