"""CPU-side checks: host logic, weight packing, and that libzvx.so loads and exports the whole C-ABI.
No compute call is made here (no GPU in the build container)."""
import ctypes
import os
import re

import numpy as np
import pytest

from zerovox_amd import _lib, config as zcfg, pack, weights as zw
from zerovox_amd.dist import shard_range
from zerovox_amd.symbols import Symbols
from zerovox_amd.synthesize import ZeroVoxTTS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _synth_stub():
    s = ZeroVoxTTS.__new__(ZeroVoxTTS)
    s._symbols = Symbols(zcfg.PHONES, zcfg.PUNCTS)
    from zerovox_amd.normalize import ZeroVoxNormalizer
    s._normalizer = ZeroVoxNormalizer("en")
    s._verbose = False
    s._model = None
    return s


def test_symbols_contract():
    sy = Symbols(zcfg.PHONES, zcfg.PUNCTS)                      # symbols.py:2-48
    assert sy.num_phones == 28 and sy.num_puncts == 10
    assert sy.encode_phone("'") == 0 and sy.encode_phone("z") == 27
    assert sy.encode_punct(" ") == 1 and sy.encode_punct(Symbols.NO_PUNCT) == 0 and sy.encode_punct('"') == 9
    assert sy.decode_phone(2) == "a" and sy.decode_punct(2) == ","
    assert sy.is_punct("-") and sy.is_phone("-")


def test_transcript2phonemids_reference_example():
    # the worked example in the reference's comment, synthesize.py:201-203
    s = _synth_stub()
    ph, pu = s.transcript2phonemids("entweder zu helfen, wenn")
    assert ph == [6, 15, 21, 24, 6, 5, 6, 19, 27, 22, 9, 6, 13, 7, 6, 15, 24, 6, 15, 15]
    assert pu == [0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0]


@pytest.mark.parametrize("text", ["a. # , b", "it's - ok!? yes", "  hi", "x,,;y ", "", "...", "a-b"])
def test_transcript2phonemids_matches_oracle(text):
    from oracle.zvx_oracle import transcript2phonemids
    s = _synth_stub()
    assert s.transcript2phonemids(text) == tuple(transcript2phonemids(text, zcfg.PHONES, zcfg.PUNCTS))


def test_empty_text_sentinel():
    # synthesize.py:219-220: no phones -> ([[0.0]], [[0]], 0, [[0.0]]) without touching the model
    wav, ph, length, mel = _synth_stub().tts_ex("  ?!  ", spkemb=None)
    assert wav.shape == (1, 1) and wav.dtype == np.float32 and ph.shape == (1, 1) and ph.dtype == np.int32
    assert length == 0 and mel.shape == (1, 1)


def test_unknown_decoder_kind_raises():
    with pytest.raises(Exception, match="unknown decoder kind"):         # model.py:244
        zcfg.medium_modelcfg("tacotron")


def test_weight_norm_fold_matches_definition():
    r = np.random.default_rng(0)
    v = r.standard_normal((6, 5, 3)).astype(np.float32)
    g = r.uniform(0.5, 2, (6, 1, 1)).astype(np.float32)
    w = zw.fold_weight_norm(g, v)
    for i in range(6):
        np.testing.assert_allclose(w[i], g[i, 0, 0] * v[i] / np.linalg.norm(v[i]), rtol=1e-6)


def test_pack_manifest_and_polyphase():
    cfg = zcfg.medium_modelcfg("styletts")
    h = zcfg.hifigan_config("tiny")
    hsd = zw.hifigan_state_dict(h, 0)
    man, blob = pack.pack_model(cfg, zw.tts_state_dict(cfg, 0), h, hsd, "bf16")
    lines = man.splitlines()
    assert lines[0] == "zvx_manifest 1" and "cfg precision bf16" in lines and "cfg dec_kind styletts" in lines
    tens = {l.split()[1]: l.split() for l in lines if l.startswith("tensor ")}
    assert blob.dtype == np.float32
    for name, f in tens.items():
        nd = int(f[3]); dims = [int(x) for x in f[4:4 + nd]]; off = int(f[4 + nd])
        assert off % 4 == 0 and off + int(np.prod(dims)) <= blob.size, name
    # polyphase ConvTranspose1d weights reproduce the oracle's transposed convolution
    from oracle import zvx_oracle as O
    f = tens["voc.up0_w"]; dims = [int(x) for x in f[4:7]]; off = int(f[7])
    poly = blob[off: off + int(np.prod(dims))].reshape(dims)              # [3][u*Cout][Cin]
    u, k = h["upsample_rates"][0], h["upsample_kernel_sizes"][0]
    x = np.random.default_rng(1).standard_normal((dims[2], 9)).astype(np.float32)
    ref = O.conv_transpose1d(x, O.fold_wn(hsd, "ups.0"), None, stride=u, padding=(k - u) // 2)
    cout = dims[1] // u
    xp = np.pad(x, ((0, 0), (1, 1)))
    got = np.zeros((cout, 9 * u), np.float32)
    for t in range(9):
        acc = sum(poly[ti] @ xp[:, t + 1 + dv] for ti, dv in enumerate((-1, 0, 1)))   # [u*Cout]
        got[:, t * u:(t + 1) * u] = acc.reshape(u, cout).T
    np.testing.assert_allclose(got, ref, atol=1e-4)


def test_shard_range_partitions():
    for n, w in ((256, 8), (7, 3), (5, 8), (32, 1)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_mel_frontend_shapes_and_filterbank():
    from zerovox_amd.mels import get_mel_from_wav, mel_filterbank
    fb = mel_filterbank(22050, 1024, 80, 0, 8000)
    assert fb.shape == (80, 513) and (fb >= 0).all() and fb[:, 400:].sum() == 0      # nothing above fmax=8 kHz
    wav = 0.1 * np.sin(2 * np.pi * 440 * np.arange(22050) / 22050).astype(np.float32)
    mel, energy = get_mel_from_wav(wav, 22050, 1024, 256, 1024, 80, 0, 8000)
    assert mel.shape == (80, 22050 // 256) and energy.shape == (22050 // 256,)          # center=False framing, mels.py:383-386
    assert int(mel.mean(axis=1).argmax()) in range(8, 20)                               # 440 Hz lands in a low mel band


def test_mel_scale_matches_librosa_documented_examples():
    """Slaney mel scale against the worked examples in librosa's own docstrings (hz_to_mel / mel_to_hz, htk=False) and the
    'slaney' area normalisation against its definition (every triangle integrates to 1 over Hz)."""
    from zerovox_amd.mels import _hz_to_mel, _mel_to_hz, mel_filterbank
    assert abs(float(_hz_to_mel(60)) - 0.9) < 1e-12
    assert np.allclose(_hz_to_mel([110, 220, 440]), [1.65, 3.3, 6.6], atol=1e-12)
    assert abs(float(_mel_to_hz(3)) - 200.0) < 1e-9
    assert np.allclose(_mel_to_hz([1, 2, 3, 4, 5]), [66.667, 133.333, 200.0, 266.667, 333.333], atol=1e-3)
    assert abs(float(_hz_to_mel(1000.0)) - 15.0) < 1e-12 and abs(float(_mel_to_hz(_hz_to_mel(6400.0))) - 6400.0) < 1e-6
    fb = mel_filterbank(22050, 1024, 80, 0, 8000).astype(np.float64)
    area = fb.sum(axis=1) * (22050 / 1024)
    assert np.all(np.abs(area[10:] - 1.0) < 0.08)          # wide triangles: the bin sum approximates the integral
    assert np.all(fb.argmax(axis=1)[1:] >= fb.argmax(axis=1)[:-1])


def test_stft_basis_matches_torch_stft():
    """The windowed DFT basis behind zvx_melspec, pinned against an independent STFT (torch.stft, periodic hann,
    center=False) -- the STFT half of get_mel_from_wav (mels.py:383-386); the Slaney mel basis stays unpinned (no librosa)."""
    import torch
    from zerovox_amd.mels import stft_basis
    rng = np.random.default_rng(0)
    x = rng.standard_normal(1024 + 256 * 9).astype(np.float32)
    ref = torch.stft(torch.from_numpy(x), n_fft=1024, hop_length=256, win_length=1024, window=torch.hann_window(1024, periodic=True),
                     center=False, return_complex=True).numpy()            # [513, frames]
    frames = np.stack([x[i * 256:i * 256 + 1024] for i in range(ref.shape[1])]).astype(np.float64)
    out = frames @ stft_basis(1024, 1024).T
    assert np.abs(out[:, :513] - ref.real.T).max() < 2e-3 and np.abs(out[:, 513:1026] - ref.imag.T).max() < 2e-3
    assert not out[:, 1026:].any()


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "zvx.h")).read()
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)                  # strip comments
    declared = set(re.findall(r"\b(zvx_[a-z0-9_]+)\s*\(", code))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.load()
    for name in declared:
        assert getattr(lib, name) is not None


def test_create_fails_loudly_without_gpu_or_with_bad_manifest():
    import torch
    lib = _lib.load()
    h = ctypes.c_void_p()
    blob = np.zeros(4, np.float32)
    rc = lib.zvx_create(b"not a manifest\n", blob.ctypes.data_as(ctypes.c_void_p), blob.nbytes, 0, ctypes.byref(h))
    assert rc != 0 and not h.value
    msg = lib.zvx_last_error(None).decode()
    if torch.cuda.is_available():
        assert rc == _lib.ZVX_E_MANIFEST and "magic" in msg
    else:
        assert rc == _lib.ZVX_E_HIP and "HIP device" in msg      # no CPU fallback: the product path refuses to run


def test_cpu_device_is_refused():
    from zerovox_amd.model import parse_device
    with pytest.raises(_lib.ZvxError):
        parse_device("cpu")
    assert parse_device("cuda:3") == 3 and parse_device("cuda") == 0


def test_committed_bench_line_follows_the_contract():
    """The bench line committed under profiles/ (produced by `python bench.py` on the GPU box) carries every field the
    driver and the judge read."""
    import json
    path = os.path.join(ROOT, "profiles", "r03_bench_n1.json")
    if not os.path.exists(path):
        pytest.skip("no committed bench line")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "samples/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["unit"] in ("GB/s", "TFLOP/s")
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert abs(d["value"] - d["config"]["global_batch"] * d["config"]["samples_per_utt"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_write_wav_to_file_is_int16_pcm_x32760(tmp_path):
    """demo.py:29-35 / model.py:44-63: (wav * 32760).astype(int16), cut to length * hop, 16-bit mono."""
    import wave
    from zerovox_amd.synthesize import write_wav_to_file
    rng = np.random.default_rng(1)
    wav = np.clip(rng.standard_normal(5000) * 0.4, -1, 1).astype(np.float32)
    f = tmp_path / "t.wav"
    write_wav_to_file(wav, length=7, filename=f, sample_rate=22050, hop_length=256)
    with wave.open(str(f), "rb") as w:
        assert (w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()) == (22050, 1, 2, 7 * 256)
        pcm = np.frombuffer(w.readframes(w.getnframes()), np.int16)
    assert np.array_equal(pcm, (wav * 32760).astype("int16")[: 7 * 256])


def test_get_speakerref_resamples_to_the_model_rate(tmp_path):
    """synthesize.py:112-121 loads the reference voice at the model's sampling rate (librosa.load(sr=...)): a 16 kHz file comes
    back at 22.05 kHz with its pitch preserved; a file already at the model rate comes back sample-exact."""
    import wave
    n = 16000
    tone = (0.5 * np.sin(2 * np.pi * 440 * np.arange(n) / 16000.0)).astype(np.float32)
    f = tmp_path / "ref16k.wav"
    with wave.open(str(f), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes((tone * 32767).astype(np.int16).tobytes())
    a = ZeroVoxTTS.get_speakerref(f, 22050)
    assert abs(len(a) - 22050) <= 1 and a.dtype == np.float32
    spec = np.abs(np.fft.rfft(a * np.hanning(len(a))))
    assert abs(np.argmax(spec) * 22050.0 / len(a) - 440.0) < 2.0
    g = tmp_path / "ref22k.wav"
    pcm = (np.random.default_rng(0).standard_normal(3000) * 3000).astype(np.int16)
    with wave.open(str(g), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(22050); w.writeframes(pcm.tobytes())
    assert np.array_equal(ZeroVoxTTS.get_speakerref(g, 22050), pcm.astype(np.float32) / 32768.0)
