"""Pins of the log-mel front end that need no GPU.

librosa is not installable here, so `oracle/mel_oracle.py` (written from librosa's published definitions, independently of
the product's zerovox_amd/mels.py) is pinned against the numbers librosa's own documentation prints, and the STFT half
against torch.stft; the product's weight-matrix builders are then checked against that oracle."""
import numpy as np
import torch

from oracle import mel_oracle as M
from zerovox_amd import mels as P


def test_mel_scale_matches_librosa_documentation():
    # librosa.hz_to_mel / mel_to_hz docstring examples
    assert abs(M.hz_to_mel(60) - 0.9) < 1e-12
    assert np.allclose([M.hz_to_mel(f) for f in (110, 220, 440)], [1.65, 3.3, 6.6], atol=1e-12)
    assert abs(M.mel_to_hz(3) - 200.0) < 1e-9
    assert np.allclose([M.mel_to_hz(m) for m in (1, 2, 3, 4, 5)], [66.667, 133.333, 200., 266.667, 333.333], atol=5e-4)
    # librosa.mel_frequencies(n_mels=40) docstring output (fmin=0, fmax=11025, htk=False), all 40 values
    doc = [0., 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855, 853.173, 938.49, 1024.856,
           1119.114, 1222.042, 1334.436, 1457.167, 1591.187, 1737.532, 1897.337, 2071.84, 2262.393, 2470.47, 2697.686,
           2945.799, 3216.731, 3512.582, 3835.643, 4188.417, 4573.636, 4994.285, 5453.621, 5955.205, 6502.92, 7101.009,
           7754.107, 8467.272, 9246.028, 10096.408, 11025.]
    assert np.allclose(M.mel_frequencies(40, 0.0, 11025.0), doc, atol=6e-4)


def test_filterbank_matches_librosa_documentation_and_definition():
    fb = M.mel_basis(22050, 2048, 128, 0.0, 11025.0)                      # librosa.filters.mel(sr=22050, n_fft=2048) example
    assert fb.shape == (128, 1025) and round(float(fb[0, 1]), 3) == 0.016 and fb[0, 0] == 0.0 and fb[-1, -1] == 0.0
    # 'slaney' normalisation: every triangle has unit area in Hz; below 1 kHz the filters are equal-width linear triangles
    fb = M.mel_basis(22050, 1024, 80, 0.0, 8000.0)
    area = fb.sum(axis=1) * (22050 / 1024.0)
    assert np.allclose(area[5:], 1.0, atol=0.12) and (fb >= 0).all()
    d = M.mel_to_hz(M.hz_to_mel(8000.0) / 81.0)                           # spacing of the first (linear) centres in Hz
    k = np.arange(513) * 22050 / 1024.0
    for i in (0, 3, 10):
        c = (i + 1) * d
        assert np.allclose(fb[i], np.maximum(0.0, 1.0 - np.abs(k - c) / d) / d, atol=1e-12)


def test_stft_half_matches_torch_stft():
    rng = np.random.default_rng(2)
    x = rng.standard_normal(5000).astype(np.float32)
    ref = torch.stft(torch.from_numpy(x), n_fft=1024, hop_length=256, win_length=1024, window=torch.hann_window(1024, periodic=True),
                     center=False, return_complex=True).abs().numpy()
    got = M.stft_magnitude(x, 1024, 256, 1024)
    assert got.shape == ref.shape and np.abs(got - ref).max() < 2e-3 * ref.max()


def test_product_weight_matrices_equal_the_oracle():
    """pack.py builds the device's `mel.basis` / `mel.dft` tensors with zerovox_amd/mels.py; both must equal the independent oracle."""
    assert np.abs(P.mel_filterbank(22050, 1024, 80, 0, 8000) - M.mel_basis(22050, 1024, 80, 0, 8000)).max() < 1e-8
    rng = np.random.default_rng(3)
    frame = rng.standard_normal((7, 1024))
    basis = P.stft_basis(1024, 1024)
    spec = frame @ basis.T
    mag = np.sqrt(spec[:, :513] ** 2 + spec[:, 513:1026] ** 2)
    ref = M.stft_magnitude(frame.reshape(-1), 1024, 1024, 1024).T
    assert np.abs(mag - ref).max() < 1e-9 * ref.max() + 1e-9
    w = rng.standard_normal(9000).astype(np.float32)
    a, ea = P.get_mel_from_wav(w, 22050, 1024, 256, 1024, 80, 0, 8000)
    b, eb = M.get_mel_from_wav(w, 22050, 1024, 256, 1024, 80, 0, 8000)
    assert a.shape == b.shape and np.abs(a - b).max() < 1e-5 and np.allclose(ea, eb, rtol=1e-5)
    sig = np.concatenate([np.zeros(5000, np.float32), 0.3 * w, np.zeros(3000, np.float32)])
    assert np.array_equal(P.trim_silence(sig), M.trim(sig))
