"""CPU oracle of the log-mel front end (TEST INFRASTRUCTURE: imported by tests/ only, never by zerovox_amd/).

Restates `get_mel_from_wav` (/root/reference/zerovox/tts/mels.py:357-395):
    wav -> np.pad(reflect, (n_fft - hop)/2) -> librosa.stft(n_fft, hop, win, window='hann', center=False) -> |.|
        -> librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) @ |S|   (mels.py:374-376; defaults htk=False, norm='slaney')
        -> log(clip(., 1e-5))                                        (mels.py:350-351, 386-388)
and `librosa.effects.trim(top_db=40)` as used by ZeroVoxTTS.speaker_embed (synthesize.py:126).

librosa (>= 0.10.2, pyproject.toml:34) is not installable in the build image, so the filterbank is written here from
librosa's PUBLISHED definitions, deliberately in a different form from the product's `zerovox_amd/mels.py` (explicit
per-filter triangles in closed form, float64, no shared helper), and pinned in tests/test_mel_oracle.py against the values
librosa's documentation prints (hz_to_mel / mel_to_hz / mel_frequencies examples, the `filters.mel` example entry) and
against torch.stft for the STFT half.
"""
import math

import numpy as np

F_SP = 200.0 / 3.0                 # Slaney: linear below 1 kHz, 66.67 Hz per mel
MIN_LOG_HZ = 1000.0
MIN_LOG_MEL = MIN_LOG_HZ / F_SP    # = 15
LOGSTEP = math.log(6.4) / 27.0     # log-spaced above: 27 mels per factor 6.4


def hz_to_mel(f):
    f = float(f)
    return f / F_SP if f < MIN_LOG_HZ else MIN_LOG_MEL + math.log(f / MIN_LOG_HZ) / LOGSTEP


def mel_to_hz(m):
    m = float(m)
    return F_SP * m if m < MIN_LOG_MEL else MIN_LOG_HZ * math.exp(LOGSTEP * (m - MIN_LOG_MEL))


def mel_frequencies(n_mels, fmin, fmax):
    lo, hi = hz_to_mel(fmin), hz_to_mel(fmax)
    return [mel_to_hz(lo + (hi - lo) * i / (n_mels - 1)) for i in range(n_mels)]


def mel_basis(sr, n_fft, n_mels, fmin, fmax):
    """[n_mels][n_fft/2+1]: filter i is the triangle with corners f[i] < f[i+1] < f[i+2] (n_mels + 2 mel-spaced frequencies),
    scaled to area 1 in Hz ('slaney': height 2 / (f[i+2] - f[i]))."""
    f = mel_frequencies(n_mels + 2, fmin, fmax)
    nbin = n_fft // 2 + 1
    w = np.zeros((n_mels, nbin), np.float64)
    for i in range(n_mels):
        left, centre, right = f[i], f[i + 1], f[i + 2]
        height = 2.0 / (right - left)
        for k in range(nbin):
            fk = k * sr / float(n_fft)
            if left < fk <= centre:
                w[i, k] = height * (fk - left) / (centre - left)
            elif centre < fk < right:
                w[i, k] = height * (right - fk) / (right - centre)
    return w


def stft_magnitude(x, n_fft, hop, win_length):
    """|librosa.stft(x, center=False, window='hann')|: frames x[t*hop : t*hop + n_fft] * periodic hann -> rfft.  [n_fft/2+1][frames]"""
    n = np.arange(win_length)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)            # scipy.signal.get_window('hann', fftbins=True)
    if win_length < n_fft:
        lp = (n_fft - win_length) // 2
        win = np.concatenate([np.zeros(lp), win, np.zeros(n_fft - win_length - lp)])
    nfr = 1 + (len(x) - n_fft) // hop
    out = np.empty((n_fft // 2 + 1, nfr), np.float64)
    for t in range(nfr):
        out[:, t] = np.abs(np.fft.rfft(np.asarray(x[t * hop: t * hop + n_fft], np.float64) * win))
    return out


def get_mel_from_wav(audio, sampling_rate, fft_size, hop_size, win_length, num_mels, fmin, fmax):
    """mels.py:357-395 -> (log-mel [num_mels][frames] f32, energy [frames])."""
    audio = np.asarray(audio, np.float32)
    p = (fft_size - hop_size) // 2
    assert len(audio) > p, "np.pad(mode='reflect') needs more samples than the padding"
    x = np.concatenate([audio[1:p + 1][::-1], audio, audio[-p - 1:-1][::-1]])     # reflect: no edge repeat
    mag = stft_magnitude(x, fft_size, hop_size, win_length).astype(np.float32)     # librosa returns complex64 -> f32 magnitudes
    mel = mel_basis(sampling_rate, fft_size, num_mels, fmin, fmax).astype(np.float32) @ mag
    return np.log(np.clip(mel, 1e-5, None)).astype(np.float32), np.linalg.norm(mag, axis=0)


def trim(wav, top_db=40, frame_length=2048, hop_length=512):
    """librosa.effects.trim: RMS per centred frame (zero padding), keep frames within top_db of the loudest."""
    wav = np.asarray(wav, np.float32)
    half = frame_length // 2
    x = np.concatenate([np.zeros(half, np.float32), wav, np.zeros(half, np.float32)])
    nfr = 1 + (len(x) - frame_length) // hop_length
    rms = np.array([math.sqrt(float(np.mean(np.square(x[t * hop_length: t * hop_length + frame_length], dtype=np.float64)))) for t in range(nfr)])
    ref = max(rms.max(), 1e-10)
    keep = [t for t in range(nfr) if 20.0 * math.log10(max(rms[t], 1e-10) / ref) > -top_db]
    if not keep:
        return wav[:0]
    return wav[keep[0] * hop_length: min(len(wav), (keep[-1] + 1) * hop_length)]
