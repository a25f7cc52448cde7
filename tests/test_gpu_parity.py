"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI (ctypes -> libzvx.so), against
(1) the golden fixtures produced by the reference itself and (2) the CPU oracle on seeded inputs.

Stated tolerances
  f32 mode  (exact-f32 MFMA everywhere): max|err| <= 2e-4 * max(1, max|ref|) on every float output
            (measured ~3e-5: fp32 summation-order noise); lengths / durations / bucket ids exact.
  bf16 mode (bf16 activations+weights, fp32 accumulate, the benchmarked mode).  The phoneme encoder + variance adaptor stay
            f32 in this mode: the variance predictors run on the exact-f32 MFMA, the encoder's FFT blocks as 3-plane split
            products on IEEE-HALF planes (hi.wh + hi.wl + lo.wh with 11-bit significands: every operand is carried to 2^-24,
            an f32 half-ulp, see ops.hip), so log-duration / pitch / energy / features keep the f32 tolerance (2e-4) and every
            discrete decision (duration, pitch / energy bucket) is held to the SAME 1e-3 ambiguity margin as the f32 mode
            (test_predicted_durations_and_buckets_exact[bf16], test_headline_shape_discrete_decisions...).
            Both mel decoders AND (round 5) the HiFi-GAN vocoder run in IEEE HALF in this mode (f16 weights + activations on the
            f16 MFMA; bf16 for the speaker encoder): the decoders' single-product bf16 floor was 7e-2 (StyleTTS) / 2.9e-2 (FS2) max
            on the log-mel, above SURVEY 8c's 2e-2; in half they measure <= 1.02e-2 / 0.22 % rms and <= 3.8e-3 / 0.09 % over every
            fixture.  Mel limits = SURVEY.md 8c's own: max|err| <= 2e-2, rms <= 0.4 % of the reference rms.  Waveform in [-1, 1]:
            SURVEY 8c allows 1e-2 / 2e-3 rms; the half vocoder is held to 4e-3 / 8e-4 end to end and 2e-3 / 4e-4 alone (measured
            <= 1.97e-3 / 4.2e-4 and <= 9.8e-4 / 1.6e-4; the bf16 vocoder of rounds 1-4 -- zvx_set_int("voc_f16", 0) -- measured
            <= 8.5e-3 / 1.9e-3 and <= 6.3e-3 / 1.3e-3).  ZVX_ERR_LOG=<file> makes every comparison append what it measured.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import zvx_oracle as O                      # checker only
from zerovox_amd import _lib, config as zcfg, pack, synthetic, weights as zw

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
_sd, _ctx = {}, {}


def tts_sd(kind):
    if kind not in _sd:
        cfg = zcfg.medium_modelcfg(kind)
        _sd[kind] = (cfg, zw.tts_state_dict(cfg, 0))
    return _sd[kind]


def voc_sd(name):
    if name not in _sd:
        h = zcfg.hifigan_config(name)
        _sd[name] = (h, zw.hifigan_state_dict(h, 0))
    return _sd[name]


_last_voc = [None]                                       # the generator of the context a test last asked for (check_wav picks its limits by it)


def ctx_for(kind, voc, prec):
    key = (kind, voc, prec)
    _last_voc[0] = voc
    if key not in _ctx:
        if len(_ctx) >= 4:                                # keep device memory bounded
            k0 = next(iter(_ctx))
            _ctx.pop(k0).close()
        cfg, sd = tts_sd(kind)
        h, hsd = voc_sd(voc)
        man, blob = pack.pack_model(cfg, sd, h, hsd, prec)
        _ctx[key] = _lib.Context(man, blob, 0)
    return _ctx[key]


def stats(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max()), float(np.sqrt(np.mean((a - b) ** 2))), float(np.sqrt(np.mean(b ** 2))), float(np.abs(b).max())


def check_f32(a, b, what, tol=2e-4):
    mx, _, _, bm = stats(a, b)
    assert mx <= tol * max(1.0, bm), f"{what}: max err {mx:.3e} (ref max {bm:.3g})"


def _errlog(kind, what, *vals):
    """ZVX_ERR_LOG=<file>: every 16-bit-mode comparison appends its measured errors (tolerances are set from these with margin)."""
    path = os.environ.get("ZVX_ERR_LOG")
    if path:
        with open(path, "a") as f:
            f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]} | {kind} {what} | " + " ".join(f"{v:.3e}" for v in vals) + "\n")


def check_mel(a, b, prec, what, kind="fastspeech2"):
    """16-bit mode: both mel decoders run in IEEE half and are held to SURVEY 8c's 2e-2 abs on the log-mel (measured over every
    fixture: StyleTTS <= 1.02e-2 / 0.22 % rms, FS2 / SCLN <= 3.6e-3 / 0.09 %).  kind "bf16": a decoder explicitly run in bf16
    (zvx_set_int flash / dec_f16 0, measured <= 2.9e-2 / 0.7 % for FS2): 4e-2 / 1 %."""
    if prec == "f32":
        return check_f32(a, b, what)
    mx, rms, ref_rms, _ = stats(a, b)
    _errlog("mel", what, mx, rms, ref_rms)
    lim_mx, lim_rel = (4e-2, 0.01) if kind == "bf16" else (2e-2, 0.004)
    assert rms <= lim_rel * ref_rms and mx <= lim_mx, f"{what}: mel err max {mx:.3e} rms {rms:.3e} (ref rms {ref_rms:.3g})"


def check_embed16(e, ref, what):
    """16-bit mode (the speaker encoder runs in bf16): a unit-norm 528-vector is held to cosine >= 0.9999 with the f32 reference
    and max |err| <= 2.5e-3 (SURVEY 8c gives no figure for this path; measured over every fixture, raw-audio input and the sampled
    clips of configs[4]: 1 - cosine <= 4.0e-5, max |err| <= 1.2e-3 -- ZVX_ERR_LOG prints them)."""
    e, ref = np.asarray(e, np.float64), np.asarray(ref, np.float64)
    cos = float(np.dot(e, ref) / (np.linalg.norm(e) * np.linalg.norm(ref)))
    mx = float(np.abs(e - ref).max())
    _errlog("embed", what, 1.0 - cos, mx)
    assert cos >= 0.9999 and mx <= 2.5e-3, f"{what}: cosine {cos:.6f}, max err {mx:.3e}"


def check_wav(a, b, prec, what, e2e=True, voc=None):
    """16-bit mode.  SURVEY 8c's waveform bound is 1e-2 abs / 2e-3 rms.  The generator's arithmetic is chosen per stage (round 6,
    zvx_set_int("voc_f16_stages")): IEEE half everywhere except a ResBlock1 stage of 128 channels -- HiFi-GAN V1's stage 2, the pair kernel,
    43 % of the generator's matrix work for 1/5 of its error budget -- which runs in bf16.  Limits = what that measures, with ~1.5x margin:
      V1 (voc "v1"):             end to end 6e-3 / 1.2e-3 (measured at the headline shape 3.55e-3 / 6.6e-4), the vocoder alone 5e-3 / 9e-4 (3.33e-3 / 6.1e-4);
      every other generator (all of it in half): end to end 4e-3 / 8e-4 (measured <= 1.97e-3 / 4.2e-4), alone 2e-3 / 4e-4 (<= 9.8e-4 / 1.6e-4).
    V1 all in half (voc_f16_stages 31) measures 1.64e-3 / 3.1e-4 and 1.02e-3 / 1.7e-4 (test_vocoder_in_ieee_half...), all in bf16
    (voc_f16 0) 7.3e-3 / 1.36e-3 and 6.7e-3 / 1.33e-3.  voc: the generator's name (default: the one of the last ctx_for call)."""
    if prec == "f32":
        return check_f32(a, b, what)
    mx, rms, _, _ = stats(a, b)
    _errlog("wav-e2e" if e2e else "wav-voc", what, mx, rms)
    mixed = (voc if voc is not None else _last_voc[0]) == "v1"
    if mixed:
        lim_mx, lim_rms = (6e-3, 1.2e-3) if e2e else (5e-3, 9e-4)
    else:
        lim_mx, lim_rms = (4e-3, 8e-4) if e2e else (2e-3, 4e-4)
    assert mx <= lim_mx and rms <= lim_rms, f"{what}: wav err max {mx:.3e} rms {rms:.3e} (limits {lim_mx:.0e} / {lim_rms:.0e})"


E2E = ["e2e_styletts_tiny_T8", "e2e_fs2_tiny_T8", "e2e_styletts_tiny_T16_pred", "e2e_fs2_tiny_T16_pred",
       "e2e_fs2_tiny2_T12_ragged", "e2e_styletts_tiny2_T12_ragged", "e2e_fs2_v2_T24", "e2e_styletts_v1_T64"]


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("name", E2E)
def test_e2e_against_reference_golden(name, prec):
    """ZeroVox.inference_ex (model.py:308-347) fixtures, staged API: encode -> decode -> vocode."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    ctx = ctx_for(str(g["decoder_kind"]), str(g["vocoder"]), prec)
    T = len(g["phoneme"])
    dur = g["duration"][None] if bool(g["forced"]) else None
    mel_len, logd, pitch, energy = ctx.encode(g["phoneme"][None], g["puncts"][None], np.array([T], np.int32), g["spk"][None], dur)
    ml = int(g["mel_len"])
    assert int(mel_len[0]) == ml                                         # exact: discrete decision
    check_f32(logd[0], g["log_duration"], "log_duration")
    check_f32(pitch[0], g["pitch"], "pitch")
    check_f32(energy[0], g["energy"], "energy")
    check_f32(ctx.fetch("encoder_out", (1, T, 528))[0] - g["spk"][None], g["encoder_raw"], "encoder_out")
    check_f32(ctx.fetch("features", (1, ml, 528))[0], g["features"], "features")
    mel = ctx.decode(1, ml)
    check_mel(mel[0, :ml], g["mel"].T, prec, "mel", str(g["decoder_kind"]))
    wav = ctx.vocode(1, mel_len, np.array([int(g["pad_to"])], np.int32))
    assert wav.shape[1] == ml * 256
    check_wav(wav[0], g["wav"], prec, "wav")


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_one_shot_synthesize_matches_golden(prec):
    g = np.load(os.path.join(GOLDEN, "e2e_fs2_tiny2_T12_ragged.npz"))      # durations with zeros and a long one
    ctx = ctx_for("fastspeech2", "tiny2", prec)
    out = ctx.synthesize(g["phoneme"][None], g["puncts"][None], np.array([12], np.int32), g["spk"][None],
                         g["duration"][None], np.array([int(g["pad_to"])], np.int32))
    ml = int(g["mel_len"])
    assert int(out["mel_len"][0]) == ml == int(np.maximum(g["duration"], 0).sum())
    check_mel(out["mel"][0, :ml], g["mel"].T, prec, "mel")
    check_wav(out["wav"][0, : ml * 256], g["wav"], prec, "wav")
    check_f32(out["log_duration"][0], g["log_duration"], "log_duration")


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("kind,voc", [("styletts", "tiny"), ("fastspeech2", "tiny2")])
def test_ragged_batch_equals_independent_oracle_calls(kind, voc, prec):
    """A padded batch must equal B independent batch-1 reference calls: no statistic sees padding."""
    ctx = ctx_for(kind, voc, prec)
    cfg, sd = tts_sd(kind)
    h, hsd = voc_sd(voc)
    Ts = [16, 5, 11]
    B, Tmax = len(Ts), max(Ts)
    ph = np.zeros((B, Tmax), np.int32); pu = np.zeros((B, Tmax), np.int32); dur = np.zeros((B, Tmax), np.int32)
    spk = np.zeros((B, 528), np.float32)
    for b, T in enumerate(Ts):
        p, q, s, d = synthetic.utterance(T, 40 + b, "uniform")
        ph[b, :T], pu[b, :T], dur[b, :T], spk[b] = p, q, d, s
    pad_to = np.array([40, 100, 8], np.int32)
    out = ctx.synthesize(ph, pu, np.array(Ts, np.int32), spk, dur, pad_to)
    for b, T in enumerate(Ts):
        ref = O.inference_ex(sd, hsd, cfg, h, ph[b, :T], pu[b, :T], spk[b], duration=dur[b, :T], pad_to=int(pad_to[b]))
        ml = ref["mel_len"]
        assert int(out["mel_len"][b]) == ml
        check_mel(out["mel"][b, :ml], ref["mel"].T, prec, f"mel[{b}]", kind)
        check_wav(out["wav"][b, : ml * 256], ref["wav"], prec, f"wav[{b}]")


@pytest.mark.parametrize("prec", ["f32", "bf16", "bf16-exact-encoder", "bf16-planes-bf16"])
def test_predicted_durations_and_buckets_exact(prec):
    """Discrete decisions (pitch / energy bucket ids, durations, mel_len) against the oracle, at the fp32 ambiguity margin of 1e-3
    for f32 mode, for the DEFAULT 16-bit mode (the encoder's split products on IEEE-half planes: 2^-24-class) and for the 16-bit
    mode with the encoder on the exact-f32 MFMA (`zvx_set_int("enc_split", 0)`).  The bf16-plane split of rounds 2-3
    (`enc_split 1`, kept as an A/B: 5e-5 on the encoder output) is held to 2e-2 bucket units / 5e-3 frames."""
    mode = {"f32": None, "bf16": 2, "bf16-exact-encoder": 0, "bf16-planes-bf16": 1}[prec]
    ctx = ctx_for("styletts", "tiny", "f32" if prec == "f32" else "bf16")
    if mode is not None:
        ctx.set_int("enc_split", mode)
    try:
        _durations_and_buckets(ctx, prec, mode != 1)
    finally:
        if mode is not None:
            ctx.set_int("enc_split", 2)


def _durations_and_buckets(ctx, prec, exact_enc):
    cfg, sd = tts_sd("styletts")
    ph, pu, T, spk, _ = synthetic.batch(3, 20, 70, None)
    mel_len, logd, pitch, energy = ctx.encode(ph, pu, T, spk)
    pidx = ctx.fetch("pitch_idx", (3, 20)); eidx = ctx.fetch("energy_idx", (3, 20)); dur = ctx.fetch("duration", (3, 20))
    for b in range(3):
        ref = O.fs2_encoder(ph[b], pu[b], spk[b], sd, cfg)
        # a rounding boundary closer than 1e-3 would make the discrete outcome legitimately ambiguous in fp32 (every mode but the
        # bf16-plane A/B is held to that); the bf16-plane split products are 5e-5-class: 2e-2 bucket units / 5e-3 frames
        mb, md = (1e-3, 1e-3) if exact_enc else (2e-2, 5e-3)
        safe_p = np.abs((ref["pitch"] * 255) % 1 - 0.5) > mb
        safe_e = (np.abs((ref["energy"] * 255) % 1 - 0.5) > mb) & (pidx[b] == ref["pitch_idx"])     # energy sees the pitch-embedded input
        safe_d = np.abs((np.exp(ref["log_duration"]) - 1) % 1 - 0.5) > md
        assert safe_p.mean() >= 0.8
        assert np.array_equal(pidx[b][safe_p], ref["pitch_idx"][safe_p])
        assert np.array_equal(eidx[b][safe_e], ref["energy_idx"][safe_e])
        assert np.array_equal(dur[b][safe_d], ref["duration"][safe_d])
        if safe_d.all() and safe_p.all() and safe_e.all():
            assert int(mel_len[b]) == ref["mel_len"]


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("voc", ["tiny", "tiny2"])
def test_vocoder_alone_against_golden(voc, prec):
    g = np.load(os.path.join(GOLDEN, "blocks_hifigan.npz"))
    ctx = ctx_for("styletts", voc, prec)
    mel = g[f"{voc}_mel"].T[None]                                         # [1][12][80]
    wav = ctx.vocode_mel(mel, np.array([12], np.int32))
    check_wav(wav[0], g[f"{voc}_wav"], prec, "wav", e2e=False)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_vocoder_v1_ragged_batch_equals_independent_oracle_calls(prec):
    """HiFi-GAN V1 (every kernel of the benchmarked vocoder: fused ResBlock pairs for C = 128/64/32 with k = 3/7/11, all three
    running-sum modes) on a ragged batch: each utterance must equal its own batch-1 oracle call."""
    h, hsd = voc_sd("v1")
    ctx = ctx_for("styletts", "v1", prec)
    P = np.array([23, 9, 17], np.int32)
    rng = np.random.default_rng(11)
    mel = np.zeros((3, int(P.max()), 80), np.float32)
    for b in range(3):
        mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
    wav = ctx.vocode_mel(mel, P)
    for b in range(3):
        ref = O.hifigan_generator(mel[b, :P[b]].T, hsd, h)
        check_wav(wav[b, :P[b] * 256], ref, prec, f"utt {b}", e2e=False)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_vocoder_v3_resblock2_ragged_batch_equals_independent_oracle_calls(prec):
    """HiFi-GAN V3 (hifigan.py:65-99: ResBlock2 = single convolutions with a residual, k = 3 / 5 / 7, dilations up to 12,
    256 -> 128 -> 64 -> 32 channels, three upsamplers 8 x 8 x 4) at its published width: each utterance of a ragged batch
    against its own batch-1 oracle call."""
    h, hsd = voc_sd("v3")
    ctx = ctx_for("styletts", "v3", prec)
    P = np.array([19, 7, 12], np.int32)
    rng = np.random.default_rng(13)
    mel = np.zeros((3, int(P.max()), 80), np.float32)
    for b in range(3):
        mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
    wav = ctx.vocode_mel(mel, P)
    for b in range(3):
        ref = O.hifigan_generator(mel[b, :P[b]].T, hsd, h)
        check_wav(wav[b, :P[b] * 256], ref, prec, f"utt {b}", e2e=False)
        assert not wav[b, P[b] * 256:].any()


def _variants_of(ctx, fn):
    """kernel variants (names) the conv / GEMM launches of one call went to"""
    ctx.set_int("profile", 2)
    ctx.reset_stats()
    try:
        fn()
        return {k["name"]: k["launches"] for k in ctx.kernel_stats()}
    finally:
        ctx.set_int("profile", 0)


def test_vocoder_v3_routing_no_gathered_row_fallback():
    """HiFi-GAN V3's k = 7 / dilation-12 convolutions span 72 rows of taps: they used to fall through to the generic gathered-row GEMM
    (0.99 + 0.63 + 0.22 ms of a 5.1 ms vocoder at the benchmark shape) because the slab kernels stage 64 halo rows.  Every
    convolution of the V3 generator must run on a slab / register-weight kernel (results: the parity tests above)."""
    ctx = ctx_for("styletts", "v3", "bf16")
    rng = np.random.default_rng(5)
    mel = rng.standard_normal((4, 96, 80)).astype(np.float32)
    v = _variants_of(ctx, lambda: ctx.vocode_mel(mel, np.full(4, 96, np.int32)))
    assert v and not [n for n in v if n.startswith("gemm_")], v
    # round 6: a ResBlock2 is ONE launch where that measured faster (rb2fuse_kernel: lrelu(x1) stays in LDS): all three blocks of the C = 32
    # stage, the k = 3 / 5 blocks of the C = 64 stage (k = 7 with its 36-row halo stays two launches)
    assert v.get("rb2fuse_bf16_c64") == 2 and v.get("rb2fuse_bf16_c32") == 3, v


@pytest.mark.parametrize("voc", ["v3", "tiny2"])
def test_fused_resblock2_equals_the_per_convolution_launches(voc):
    """rb2fuse_kernel (a whole ResBlock2 per launch for C = 32 / 64: x1 = x + c_0(lrelu(x)) kept in LDS as lrelu(x1), x2 = x1 + c_1(lrelu(x1)),
    k = 3 / 5 / 7 with second dilations up to 12 = 36 halo rows either side) against zvx_set_int("rb2fuse", 0), every convolution its own
    launch: the same 16-bit rounding of the intermediate and the same accumulation order -- bit-identical waveforms, in half and in bf16;
    ragged batches, one-frame utterances, utterances shorter than a tile's halo and longer than several tiles."""
    ctx = ctx_for("styletts", voc, "bf16")
    rng = np.random.default_rng(71)
    try:
        for f16 in (1, 0):
            ctx.set_int("voc_f16", f16)
            for B, Pmax in ((3, 23), (1, 1), (2, 300), (5, 70), (1, 2)):
                P = rng.integers(1, Pmax + 1, B).astype(np.int32); P[0] = Pmax
                mel = np.zeros((B, Pmax, 80), np.float32)
                for b in range(B):
                    mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
                ctx.set_int("rb2fuse", 0); ref = ctx.vocode_mel(mel, P)
                ctx.set_int("rb2fuse", 1); got = ctx.vocode_mel(mel, P)
                assert np.isfinite(got).all() and np.array_equal(got, ref), (voc, f16, B, Pmax, float(np.abs(got - ref).max()))
    finally:
        ctx.set_int("rb2fuse", 1); ctx.set_int("voc_f16", 1)


def test_speaker_encoder_routing_persistent_kernels():
    """3 s clips: the C = 32 / 64 levels run on conv2d_persist_kernel, both covered level transitions on conv2d_s2_kernel; what is left on
    the gathered-row GEMM is the 128 -> 256 transition's conv1 and two shortcut convolutions (+ the attention's second projection)."""
    ctx = ctx_for("styletts", "tiny", "bf16")
    rng = np.random.default_rng(6)
    mels = rng.standard_normal((4, 258, 80)).astype(np.float32)
    v = _variants_of(ctx, lambda: ctx.spkemb(mels, np.full(4, 258, np.int32)))
    assert v.get("conv2d_persist_c32") == 6 and v.get("conv2d_persist_c64") == 7, v
    assert v.get("conv2d_s2_c32") == 1 and v.get("conv2d_s2_c64") == 1, v
    assert "convreg_bf16_c32" not in v and "convreg_bf16_c64" not in v, v
    assert sum(n for k, n in v.items() if k.startswith("gemm_bf16")) <= 4, v


def test_vocoder_many_short_utterances():
    """130 utterances of 1-4 frames: more utterances than the fused kernel's LDS length table holds (scalar-load path),
    fewer tiles than CUs, row tiles entirely past an utterance's end."""
    h, hsd = voc_sd("tiny")
    ctx = ctx_for("styletts", "tiny", "bf16")
    rng = np.random.default_rng(5)
    B = 130
    P = rng.integers(1, 5, B).astype(np.int32)
    mel = np.zeros((B, 4, 80), np.float32)
    for b in range(B):
        mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
    wav = ctx.vocode_mel(mel, P)
    for b in range(0, B, 7):
        ref = O.hifigan_generator(mel[b, :P[b]].T, hsd, h)
        check_wav(wav[b, :P[b] * 256], ref, "bf16", f"utt {b}", e2e=False)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_decoders_alone_against_golden(prec):
    g = np.load(os.path.join(GOLDEN, "blocks_tts.npz"))
    for kind, key in (("styletts", "dec_styletts_y"), ("fastspeech2", "dec_fs2_y")):
        ctx = ctx_for(kind, "tiny", prec)
        mel = ctx.decode_features(g["dec_x"][None], np.array([20], np.int32), g["spk"][None])
        check_mel(mel[0], g[key], prec, kind, kind)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("name", ["spkemb_T96", "spkemb_T258"])
def test_speaker_encoder_against_golden(name, prec):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    ctx = ctx_for("styletts", "tiny", prec)
    e = ctx.spkemb(g["ref_mel"][None], np.array([g["ref_mel"].shape[0]], np.int32))[0]
    assert abs(np.linalg.norm(e) - 1.0) < 1e-4
    if prec == "f32":
        mx, _, _, bm = stats(e, g["embed"])
        assert mx <= 2e-5, f"embed err {mx:.3e} (ref max {bm:.3g})"
    else:
        check_embed16(e, g["embed"], name)


def test_speaker_encoder_ragged_batch():
    ctx = ctx_for("styletts", "tiny", "f32")
    cfg, sd = tts_sd("styletts")
    r = np.random.default_rng(3)
    lens = np.array([64, 37, 50], np.int32)
    mels = r.standard_normal((3, 64, 80)).astype(np.float32)
    e = ctx.spkemb(mels, lens)
    for b in range(3):
        check_f32(e[b], O.resnet_se34v2(mels[b, :lens[b]], sd, cfg), f"embed[{b}]", 2e-5)


def test_full_size_properties_config2():
    """BASELINE config #2 (B=32 x T=128, styledec + V1, bf16): size-independent properties."""
    ctx = ctx_for("styletts", "v1", "bf16")
    ph, pu, T, spk, dur = synthetic.batch(32, 128, 0, "const7")
    pad_to = np.full(32, 896, np.int32)
    out = ctx.synthesize(ph, pu, T, spk, dur, pad_to)
    assert (out["mel_len"] == 896).all() and out["wav"].shape == (32, 229376)
    assert np.isfinite(out["wav"]).all() and np.abs(out["wav"]).max() <= 1.0 and out["wav"].std() > 1e-3
    # batch invariance: utterance 5 synthesised alone is bit-identical (utterances never interact)
    solo = ctx.synthesize(ph[5:6], pu[5:6], T[5:6], spk[5:6], dur[5:6], pad_to[5:6])
    assert np.array_equal(solo["wav"][0], out["wav"][5]) and np.array_equal(solo["mel"][0], out["mel"][5])
    # permutation equivariance
    perm = np.array([3, 1, 2, 0])
    sub = ctx.synthesize(ph[perm], pu[perm], T[perm], spk[perm], dur[perm], pad_to[perm])
    for i, p in enumerate(perm):
        assert np.array_equal(sub["wav"][i], out["wav"][p])
    # once >= ~9 trailing zero frames exist, more padding leaves the audio unchanged (SURVEY.md §7 probe of model.py:331-335)
    m1 = ctx.synthesize(ph[:1], pu[:1], T[:1], spk[:1], dur[:1], np.array([960], np.int32))
    m2 = ctx.synthesize(ph[:1], pu[:1], T[:1], spk[:1], dur[:1], np.array([1100], np.int32))
    assert np.abs(m1["wav"][0] - m2["wav"][0]).max() < 1e-6
    assert np.abs(m1["wav"][0, :200000] - out["wav"][0, :200000]).max() < 1e-6       # far from the tail: identical


@pytest.mark.parametrize("kind", ["styletts", "fastspeech2"])
def test_config2_headline_utterance_against_oracle(kind):
    """BASELINE configs[1] at its own size in the benchmarked 16-bit mode: the 32 x 128-phoneme batch (durations 7 -> 896 frames,
    pad_to 896) is synthesised as ONE call and utterance 7 of it is compared with the oracle's batch-1 `inference_ex` -- mel within
    SURVEY 8c's 2e-2 / 0.4 %, waveform within 1e-2 / 2e-3, per-phoneme predictions within the f32 tolerance, mel_len exact."""
    cfg, sd = tts_sd(kind)
    h, hsd = voc_sd("v1")
    ctx = ctx_for(kind, "v1", "bf16")
    ph, pu, T, spk, dur = synthetic.batch(32, 128, 0, "const7")
    pad_to = np.full(32, 896, np.int32)
    out = ctx.synthesize(ph, pu, T, spk, dur, pad_to)
    b = 7
    ref = O.inference_ex(sd, hsd, cfg, h, ph[b], pu[b], spk[b], duration=dur[b], pad_to=896)
    assert ref["mel_len"] == 896 == int(out["mel_len"][b])
    check_f32(out["log_duration"][b], ref["log_duration"], "log_duration")
    check_mel(out["mel"][b, :896], ref["mel"].T, "bf16", f"headline mel [{kind}]", kind)
    check_wav(out["wav"][b, :896 * 256], ref["wav"], "bf16", f"headline wav [{kind}]")


def test_headline_shape_discrete_decisions_equal_the_exact_f32_encoder_up_to_f32_noise():
    """32 x 128 phonemes with PREDICTED durations in the default 16-bit mode (split products on IEEE-half planes) against the
    same context with the encoder on the exact-f32 MFMA: every pitch / energy bucket id and duration whose exact-f32 value sits
    more than 1e-3 from its rounding boundary is identical (the fp32 ambiguity margin the f32 mode is held to against the oracle),
    and the per-phoneme predictions agree to 2e-5 (measured ~1e-6: summation-order noise of two f32-class paths)."""
    ctx = ctx_for("styletts", "tiny", "bf16")
    ph, pu, T, spk, _ = synthetic.batch(32, 128, 0, None)
    res = {}
    try:
        for mode in (0, 2):
            ctx.set_int("enc_split", mode)
            mel_len, logd, pitch, energy = ctx.encode(ph, pu, T, spk)
            res[mode] = dict(mel_len=mel_len, logd=logd, pitch=pitch, energy=energy, pidx=ctx.fetch("pitch_idx", (32, 128)),
                             eidx=ctx.fetch("energy_idx", (32, 128)), dur=ctx.fetch("duration", (32, 128)))
    finally:
        ctx.set_int("enc_split", 2)
    a, b = res[0], res[2]
    same_p = a["pidx"] == b["pidx"]
    # the energy predictor (two k = 3 convolutions) sees the pitch-embedded rows t-2 .. t+2: where a boundary-sitting pitch bucket
    # moved, a whole 528-wide embedding row differs and the energies around it are not comparable
    near = np.ones_like(same_p)
    for sh in (-2, -1, 0, 1, 2):
        near &= np.roll(same_p, sh, 1)
    for k in ("logd", "pitch", "energy"):
        d = np.abs(a[k] - b[k])
        d = float((d[near] if k == "energy" else d).max())
        _errlog("enc-split", k, d)
        assert d <= 2e-5, (k, d)
    safe_p = np.abs((a["pitch"] * 255) % 1 - 0.5) > 1e-3
    assert same_p[safe_p].all()
    safe_e = (np.abs((a["energy"] * 255) % 1 - 0.5) > 1e-3) & near
    assert (a["eidx"] == b["eidx"])[safe_e].all()
    safe_d = np.abs((np.exp(a["logd"]) - 1) % 1 - 0.5) > 1e-3
    assert (a["dur"] == b["dur"])[safe_d].all()
    moved = int((~same_p).sum()) + int(((a["eidx"] != b["eidx"]) & near).sum()) + int((a["dur"] != b["dur"]).sum())
    _errlog("enc-split", "decisions moved of 12288 (pitch, energy away from a moved pitch row, durations)", moved)
    assert moved <= 6, moved      # measured 2 pitch buckets that sit within 3e-6 x 255 of a boundary: what two exact f32 summation orders also move


def test_fs2_half_decoder_after_the_exact_f32_encoder_on_ragged_batches_stays_finite():
    """the FS2 decoder in IEEE half projects V into the FFN buffer it shares with the encoder; with the encoder
    on the exact-f32 MFMA (`enc_split 0`) that buffer holds f32 bit patterns (1/32 of their low halves read as half Inf / NaN).
    Rows past an utterance's length are not written by the V projection, so the 16-bit transpose must not carry them into V^T
    (P = 0 there, but 0 * NaN = NaN).  Ragged batch with mel lengths not multiples of 8, two consecutive calls, against the oracle."""
    cfg, sd = tts_sd("fastspeech2")
    h, hsd = voc_sd("tiny2")
    ctx = ctx_for("fastspeech2", "tiny2", "bf16")
    Ts = [13, 5, 9]                                                        # x 7 frames: 91, 35, 63 -- none a multiple of 8
    B, Tmax = len(Ts), max(Ts)
    ph = np.zeros((B, Tmax), np.int32); pu = np.zeros((B, Tmax), np.int32); dur = np.zeros((B, Tmax), np.int32)
    spk = np.zeros((B, 528), np.float32)
    for b, T in enumerate(Ts):
        p, q, s, d = synthetic.utterance(T, 90 + b, "const7")
        ph[b, :T], pu[b, :T], dur[b, :T], spk[b] = p, q, d, s
    try:
        ctx.set_int("enc_split", 0)
        for call in range(2):
            out = ctx.synthesize(ph, pu, np.array(Ts, np.int32), spk, dur, None)
            assert np.isfinite(out["mel"]).all() and np.isfinite(out["wav"]).all(), f"call {call}"
        for b, T in enumerate(Ts):
            ref = O.inference_ex(sd, hsd, cfg, h, ph[b, :T], pu[b, :T], spk[b], duration=dur[b, :T], pad_to=0)
            ml = ref["mel_len"]
            assert int(out["mel_len"][b]) == ml
            check_mel(out["mel"][b, :ml], ref["mel"].T, "bf16", f"mel[{b}]", "fastspeech2")
            check_wav(out["wav"][b, : ml * 256], ref["wav"], "bf16", f"wav[{b}]")
    finally:
        ctx.set_int("enc_split", 2)


@pytest.mark.parametrize("kind", ["styletts", "fastspeech2"])
def test_batch_flattened_decoder_convolutions_equal_per_utterance_launches(kind):
    """The StyleTTS decoder's buffers keep one padding row per utterance so that its k = 3 / 1x1 convolutions run over the whole
    batch as ONE row axis (GemmArgs::bflat: row tiles cross utterance boundaries, rows past an utterance's length are staged as
    zeros).  Against the per-utterance launches (`dec_flat 0`): bit-identical mel and waveform for ragged batches whose utterances
    end inside, at and just past tile boundaries, in half and in bf16; an utterance alone (never flattened) equals its batch row."""
    ctx = ctx_for(kind, "tiny", "bf16")                  # (the FS2 / SCLN decoder pads every utterance by the k = 9 halo and flattens the GEMMs of its FFT blocks)
    rng = np.random.default_rng(31)
    try:
        for f16 in (1, 0):
            ctx.set_int("dec_f16", f16)
            for Ls in ([255, 256, 257, 130, 511], [896, 129, 640, 385], [300] * 7):
                B, Lmax = len(Ls), max(Ls)
                feats = np.zeros((B, Lmax, 528), np.float32)
                for b, L in enumerate(Ls):
                    feats[b, :L] = rng.standard_normal((L, 528)).astype(np.float32)
                spk = rng.standard_normal((B, 528)).astype(np.float32); spk /= np.linalg.norm(spk, axis=1, keepdims=True)
                Ln = np.array(Ls, np.int32)
                ctx.set_int("dec_flat", 0); ref = ctx.decode_features(feats, Ln, spk).copy()
                ctx.set_int("dec_flat", 1); got = ctx.decode_features(feats, Ln, spk).copy()
                assert np.isfinite(got).all() and np.array_equal(got, ref), (f16, Ls)
                solo = ctx.decode_features(feats[1:2, :Ls[1]], Ln[1:2], spk[1:2])
                assert np.array_equal(solo[0], got[1, :Ls[1]]), (f16, Ls)
                # the 1x1 shortcut of a residual block inside its last k = 3 convolution (a second source of the K loop) against the two
                # launches: the fused form keeps conv2's result in the f32 accumulator instead of rounding it to 16 bits in between
                if kind != "styletts":
                    continue
                ctx.set_int("dec_sc_fuse", 0); two = ctx.decode_features(feats, Ln, spk).copy(); ctx.set_int("dec_sc_fuse", 1)
                d = max(float(np.abs(two[b, :L] - got[b, :L]).max()) for b, L in enumerate(Ls))
                _errlog("sc-fuse", f"f16={f16}", d)
                assert d <= (1.5e-2 if f16 else 6e-2), (f16, Ls, d)          # measured 6.5e-3 (half): one 11-bit rounding of an O(10) tensor, carried through three more blocks
    finally:
        ctx.set_int("dec_flat", 1); ctx.set_int("dec_f16", 1); ctx.set_int("dec_sc_fuse", 1)


@pytest.mark.parametrize("kind", ["styletts", "fastspeech2"])
def test_decoder_padding_rows_never_reach_a_result(kind):
    """batch-flattened decoder launches write every row of the flattened axis (padding rows and rows past an utterance's
    length receive junk) and later launches read them; correctness rests on every consumer masking by select, never multiplying by
    zero.  zvx_set_int("poison_pads", 1) fills every decoder work buffer with NaN bit patterns before each decode: a ragged batch must
    come out bit-identical to the clean run (and finite), twice in a row, in the 16-bit mode and in f32."""
    for prec in ("bf16", "f32"):
        ctx = ctx_for(kind, "tiny", prec)
        rng = np.random.default_rng(23)
        L = np.array([255, 256, 257, 130, 31], np.int32)
        feats = np.zeros((len(L), int(L.max()), 528), np.float32)
        for b in range(len(L)):
            feats[b, :L[b]] = rng.standard_normal((L[b], 528)).astype(np.float32)
        spk = rng.standard_normal((len(L), 528)).astype(np.float32); spk /= np.linalg.norm(spk, axis=1, keepdims=True)
        clean = ctx.decode_features(feats, L, spk)
        try:
            ctx.set_int("poison_pads", 1)
            p1 = ctx.decode_features(feats, L, spk)
            p2 = ctx.decode_features(feats, L, spk)
        finally:
            ctx.set_int("poison_pads", 0)
        for b in range(len(L)):
            assert np.isfinite(p1[b, :L[b]]).all(), f"{kind} {prec}: NaN reached utterance {b}"
            assert np.array_equal(p1[b, :L[b]], clean[b, :L[b]]) and np.array_equal(p2[b, :L[b]], clean[b, :L[b]]), f"{kind} {prec}: utterance {b} depends on padding rows"


def test_streaming_pair_kernel_on_a_ragged_batch_against_the_oracle():
    """pairstream.hip forced for every job size (`pairstream 3`) on a ragged HiFi-GAN V1 batch, DIRECTLY against the oracle's
    batch-1 generator calls (the bit-equality test below ties it to the two-launch path; this one does not lean on that chain)."""
    h, hsd = voc_sd("v1")
    ctx = ctx_for("styletts", "v1", "bf16")
    P = np.array([37, 9, 21, 30], np.int32)
    rng = np.random.default_rng(23)
    mel = np.zeros((4, int(P.max()), 80), np.float32)
    for b in range(4):
        mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
    try:
        ctx.set_int("pairstream", 3)
        wav = ctx.vocode_mel(mel, P)
    finally:
        ctx.set_int("pairstream", 1)
    for b in range(4):
        ref = O.hifigan_generator(mel[b, :P[b]].T, hsd, h)
        check_wav(wav[b, :P[b] * 256], ref, "bf16", f"utt {b}", e2e=False)
        assert not wav[b, P[b] * 256:].any()


def test_error_behaviour():
    ctx = ctx_for("styletts", "tiny", "f32")
    ph, pu, T, spk, dur = synthetic.batch(1, 8, 0, "uniform")
    bad = ph.copy(); bad[0, 3] = 29                                      # nn.Embedding IndexError in the reference
    with pytest.raises(_lib.ZvxError) as e:
        ctx.encode(bad, pu, T, spk, dur)
    assert e.value.code == _lib.ZVX_E_INVALID and "phoneme id" in str(e.value)
    with pytest.raises(_lib.ZvxError) as e:
        ctx.decode(1, 4)                                                 # state was invalidated by the failed encode
    assert e.value.code == _lib.ZVX_E_STATE
    cfg, sd = tts_sd("styletts")
    h, hsd = voc_sd("tiny")
    man, blob = pack.pack_model(cfg, sd, h, hsd, "f32")
    with pytest.raises(_lib.ZvxError) as e:
        _lib.Context(man.replace("cfg dec_kind styletts", "cfg dec_kind tacotron"), blob, 0)
    assert e.value.code == _lib.ZVX_E_MANIFEST and "unknown decoder kind" in str(e.value)     # model.py:244
    with pytest.raises(_lib.ZvxError):
        _lib.Context(man.replace("tensor enc.emb ", "tensor enc.embx "), blob, 0).encode(ph, pu, T, spk, dur)


def test_host_api_tts_ex_roundtrip():
    """ZeroVoxTTS mirror (synthesize.py:213-243): forced durations, stateful _min_mel_len, sentinel."""
    from zerovox_amd.synthesize import ZeroVoxTTS
    modelcfg, synth = ZeroVoxTTS.load_model("synthetic:styletts", "synthetic:tiny", infer_device="cuda:0", precision="f32")
    cfg, sd = tts_sd("styletts")
    h, hsd = voc_sd("tiny")
    spk = synthetic.utterance(4, 9)[2]
    text = "hello, world"
    ids, pus = synth.transcript2phonemids(text)
    dur = [3] * len(ids)
    wav, phoneme, length, mel = synth.tts_ex(text, spk[None, None], duration=dur)
    assert phoneme.shape == (1, len(ids)) and length == 3 * len(ids) and wav.shape == (length * 256,) and mel.shape == (80, length)
    ref = O.inference_ex(sd, hsd, cfg, h, np.array(ids), np.array(pus), spk, duration=np.array(dur), pad_to=689)
    check_f32(wav, ref["wav"], "tts_ex wav")
    check_f32(mel, ref["mel"], "tts_ex mel")
    assert synth.model._min_mel_len == 689
    w0, p0, l0 = synth.tts("?!", spk[None, None])
    assert l0 == 0 and w0.shape == (1, 1)


# ------------------------------------------------------------------------------------------------
# edge cases of the reference path
# ------------------------------------------------------------------------------------------------
def test_long_sequences_use_recomputed_position_tables():
    """T > max_txt_len (512) and L > max_mel_len (1750): the reference recomputes the sinusoid table on the fly
    (fs2.py:383-388, 287-294)."""
    ctx = ctx_for("fastspeech2", "tiny", "f32")
    cfg, sd = tts_sd("fastspeech2")
    T = 520
    ph, pu, spk, _ = synthetic.utterance(T, 90, None)
    dur = np.full(T, 4, np.int32)                                       # L = 2080 > 1750
    mel_len, logd, _, _ = ctx.encode(ph[None], pu[None], np.array([T], np.int32), spk[None], dur[None])
    assert int(mel_len[0]) == 2080
    ref = O.fs2_encoder(ph, pu, spk, sd, cfg, dur)
    check_f32(ctx.fetch("encoder_out", (1, T, 528))[0], ref["encoder_out"], "encoder_out T>512")
    mel = ctx.decode(1, 2080)
    check_f32(mel[0], O.fs2_decoder(ref["features"], spk, sd, cfg), "mel L>1750", 5e-4)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_tiny_utterances(prec):
    """1-phoneme / 2-frame utterances next to a normal one (the shortest lengths the conv stacks accept).
    FS2 decoder only: with the StyleTTS decoder a 2..5-frame utterance consists of repeated identical rows, its
    InstanceNorm variance is ~0 and the reference's own fp32 result is noise there (the oracle in fp32 vs fp64 differs by
    2.0 at L=2 and 4e-3 at L=5, but 1.6e-5 at L=11) -- nothing to be exact against."""
    ctx = ctx_for("fastspeech2", "tiny", prec)
    cfg, sd = tts_sd("fastspeech2")
    h, hsd = voc_sd("tiny")
    Ts = [1, 9]
    ph = np.zeros((2, 9), np.int32); pu = np.zeros((2, 9), np.int32); dur = np.zeros((2, 9), np.int32); spk = np.zeros((2, 528), np.float32)
    for b, T in enumerate(Ts):
        p, q, s, _ = synthetic.utterance(T, 95 + b, None)
        ph[b, :T], pu[b, :T], spk[b] = p, q, s
    dur[0, 0] = 2
    dur[1, :9] = [0, 3, 0, 0, 5, 1, 0, 2, 0]                          # dropped phonemes (duration 0) inside and at both ends
    out = ctx.synthesize(ph, pu, np.array(Ts, np.int32), spk, dur, np.array([4, 11], np.int32))
    for b, T in enumerate(Ts):
        ref = O.inference_ex(sd, hsd, cfg, h, ph[b, :T], pu[b, :T], spk[b], duration=dur[b, :T], pad_to=[4, 11][b])
        ml = ref["mel_len"]
        assert int(out["mel_len"][b]) == ml == int(dur[b, :T].sum())
        check_mel(out["mel"][b, :ml], ref["mel"].T, prec, f"mel[{b}]", "fastspeech2")
        check_wav(out["wav"][b, : ml * 256], ref["wav"], prec, f"wav[{b}]")


def test_predicted_length_overflow_reports_buffer_error():
    ctx = ctx_for("styletts", "tiny", "f32")
    ph, pu, T, spk, _ = synthetic.batch(1, 12, 0, None)
    with pytest.raises(_lib.ZvxError) as e:
        ctx.synthesize(ph, pu, T, spk, None, None, Lmax_cap=3)          # predicted ~6 frames per phoneme
    assert e.value.code == _lib.ZVX_E_BUFFER
    out = ctx.synthesize(ph, pu, T, spk, None, np.array([64], np.int32), Lmax_cap=400)
    assert 12 <= int(out["mel_len"][0]) <= 400


def test_speaker_encoder_short_and_odd_lengths():
    ctx = ctx_for("styletts", "tiny", "f32")
    cfg, sd = tts_sd("styletts")
    r = np.random.default_rng(11)
    lens = np.array([9, 17, 31, 2], np.int32)                            # stride-2 levels shrink these to 2 / 3 / 4 / 1 columns
    mels = r.standard_normal((4, 31, 80)).astype(np.float32)
    e = ctx.spkemb(mels, lens)
    for b in range(4):
        check_f32(e[b], O.resnet_se34v2(mels[b, :lens[b]], sd, cfg), f"embed[{b}]", 5e-5)


def test_speaker_encoder_persistent_convolutions_and_fused_se_pool():
    """Round 5: the C = 32 / 64 levels' 3 x 3 convolutions run as persistent workgroups (conv2d_persist_kernel) whose second
    convolution of a block leaves the squeeze-excite pool's partial sums itself.  A ragged batch with widths either side of the tile
    and staging-step boundaries: every clip against the oracle, and the three launch sets (default / pool as its own pass / the
    per-tile kernels of rounds 1-4) against each other -- they differ by the order of f32 additions in the pool and nothing else."""
    ctx = ctx_for("styletts", "tiny", "bf16")
    cfg, sd = tts_sd("styletts")
    r = np.random.default_rng(23)
    lens = np.array([258, 200, 97, 64, 131], np.int32)
    mels = r.standard_normal((5, 258, 80)).astype(np.float32)
    try:
        e_new = ctx.spkemb(mels, lens)
        ctx.set_int("spk_pool_fuse", 0)
        e_pass = ctx.spkemb(mels, lens)
        ctx.set_int("slab_small", 2 | 32)
        e_old = ctx.spkemb(mels, lens)
        ctx.set_int("slab_small", 2)
        ctx.set_int("spk_s2_fuse", 0)                                       # level transition as two launches of the gathered-row GEMM
        e_s2 = ctx.spkemb(mels, lens)
    finally:
        ctx.set_int("spk_pool_fuse", 1)
        ctx.set_int("slab_small", 2)
        ctx.set_int("spk_s2_fuse", 1)
    for b in range(5):
        ref = O.resnet_se34v2(mels[b, :lens[b]], sd, cfg)
        check_embed16(e_new[b], ref, f"persistent convolutions + fused pool, clip {b} ({lens[b]} frames)")
        check_embed16(e_pass[b], ref, f"persistent convolutions + pool pass, clip {b}")
    # The persistent kernels run the per-tile kernels' matrix steps and arithmetic: bit-identical (measured: 0).  The fused pool sums
    # the f32 results BEFORE their bf16 rounding (the pass sums the rounded tensor): the gates move by bf16-rounding noise, the
    # embedding by <= 6.3e-4 (measured), the same size as the bf16 path's distance to the f32 oracle (<= 6.7e-4 on these clips).
    d1, d2 = float(np.abs(e_new - e_pass).max()), float(np.abs(e_pass - e_old).max())
    _errlog("embed-variants", "fused pool vs pool pass / persistent vs per-tile kernels", d1, d2)
    assert d1 <= 1.5e-3 and d2 <= 1e-6, f"launch sets disagree: fused pool vs pass {d1:.3e}, persistent vs per-tile {d2:.3e}"
    # the fused level transition (conv2d_s2_kernel: stride-2 conv1 + shortcut in one launch) against the two gathered-row launches:
    # same products, another order of f32 additions
    d3 = float(np.abs(e_s2 - e_pass).max())
    _errlog("embed-variants", "fused level transition vs two gathered-row launches", d3)
    assert d3 <= 1.5e-3, f"fused level transition disagrees with the gathered-row launches: {d3:.3e}"
    # the same call twice: bit-identical (the pool's partial sums are folded in a fixed order, no atomics)
    assert np.array_equal(e_new, ctx.spkemb(mels, lens))


def test_speaker_encoder_long_reference_clips_take_the_fallback_kernels():
    """Reference clips longer than the persistent / flattened kernels cover (a 3 x 3 convolution over a flattened map needs
    2 (width + 1) halo rows in LDS: widths up to 271; the stride-2 kernel's parity planes up to 135 output columns): an 8 s clip
    (700 frames) beside a short one -- level 0 runs on the gathered-row GEMM, level 1 (351 columns) too, levels 2-3 on the
    flattened kernels; every clip against the oracle."""
    ctx = ctx_for("styletts", "tiny", "bf16")
    cfg, sd = tts_sd("styletts")
    r = np.random.default_rng(29)
    lens = np.array([700, 300, 541], np.int32)
    mels = r.standard_normal((3, 700, 80)).astype(np.float32)
    e = ctx.spkemb(mels, lens)
    assert np.abs(np.linalg.norm(e, axis=1) - 1.0).max() < 1e-4
    for b in range(3):
        check_embed16(e[b], O.resnet_se34v2(mels[b, :lens[b]], sd, cfg), f"long clip {b} ({lens[b]} frames)")


def test_stage_times_and_kernel_stats_are_reported():
    ctx = ctx_for("styletts", "tiny", "bf16")
    ph, pu, T, spk, dur = synthetic.batch(2, 16, 0, "uniform")
    ctx.set_int("profile", 2)
    ctx.reset_stats()
    ctx.synthesize(ph, pu, T, spk, dur, np.array([32, 32], np.int32))
    st, ks = ctx.stage_times(), ctx.kernel_stats()
    ctx.set_int("profile", 0)
    assert st["encoder"] > 0 and st["decoder"] > 0 and st["vocoder"] > 0
    assert ks and all(k["launches"] > 0 and k["ms"] > 0 and k["flops"] > 0 for k in ks)


def test_mel_frontend_matches_oracle():
    """zvx_melspec (reflect pad + DFT GEMM + mel GEMM + log-clip on the device) against the independent CPU oracle of
    get_mel_from_wav (oracle/mel_oracle.py, pinned to librosa's documented values in tests/test_mel_oracle.py) on a ragged batch."""
    from oracle.mel_oracle import get_mel_from_wav
    ctx = ctx_for("styletts", "tiny", "bf16")
    rng = np.random.default_rng(3)
    sr = 22050
    wavs = []
    for n in (22050, 9000, 513, 40000):
        t = np.arange(n) / sr
        wavs.append((0.3 * np.sin(2 * np.pi * 220 * t) + 0.1 * np.sin(2 * np.pi * 3100 * t) + 0.05 * rng.standard_normal(n)).astype(np.float32))
    ctx.melspec([np.ones(60000, np.float32)])                             # a longer call first: the rows handed back below must not carry it
    mel, frames = ctx.melspec(wavs)
    for b, w in enumerate(wavs):
        ref, _ = get_mel_from_wav(w, sr, 1024, 256, 1024, 80, 0, 8000)
        assert int(frames[b]) == ref.shape[1]
        check_f32(mel[b, :frames[b]], ref.T, f"mel {b}", tol=5e-4)
        assert not mel[b, frames[b]:].any()
    with pytest.raises(_lib.ZvxError):
        ctx.melspec([np.zeros(100, np.float32)])                          # shorter than the reflect padding


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_speaker_embed_end_to_end_from_raw_audio(prec):
    """ZeroVoxTTS.speaker_embed (synthesize.py:123-143): trim -> log-mel -> ResNetSE34V2, raw waveform in, [1,1,528] out,
    against the oracle chain (mel_oracle.trim / get_mel_from_wav + zvx_oracle.resnet_se34v2)."""
    from oracle import mel_oracle as MO
    from zerovox_amd.synthesize import ZeroVoxTTS
    _, synth = ZeroVoxTTS.load_model("synthetic:styletts", "synthetic:tiny", infer_device="cuda:0", precision=prec)
    cfg, sd = tts_sd("styletts")
    rng = np.random.default_rng(17)
    n = 22050 * 2
    t = np.arange(n) / 22050.0
    voiced = (0.25 * np.sin(2 * np.pi * 140 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.02 * rng.standard_normal(n)).astype(np.float32)
    wav = np.concatenate([np.zeros(6000, np.float32), voiced, 1e-4 * rng.standard_normal(9000).astype(np.float32)])
    e = synth.speaker_embed(wav)
    assert e.shape == (1, 1, 528) and abs(np.linalg.norm(e) - 1.0) < 1e-4
    trimmed = MO.trim(wav, top_db=40)
    assert 0 < len(trimmed) < len(wav)
    mel, _ = MO.get_mel_from_wav(trimmed, 22050, 1024, 256, 1024, 80, 0, 8000)
    ref = O.resnet_se34v2(mel.T, sd, cfg)
    if prec == "f32":
        mx, _, _, bm = stats(e[0, 0], ref)
        assert mx <= 5e-5, f"speaker_embed err {mx:.3e} (ref max {bm:.3g})"
    else:
        check_embed16(e[0, 0], ref, "speaker_embed from raw audio")


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_streamed_vocoding_equals_whole_utterance(prec):
    """Chunked vocoding with a 16-frame halo reproduces the whole-utterance waveform: bit-exact in bf16 mode
    (every output sample sees the same inputs in the same order), to rounding in f32 mode; a halo that is too small does not."""
    from zerovox_amd.model import ZeroVox
    ctx = ctx_for("styletts", "v1", prec)
    zv = ZeroVox.__new__(ZeroVox)
    zv._ctx, zv._hop_length = ctx, 256
    rng = np.random.default_rng(21)
    mel = rng.standard_normal((150, 80)).astype(np.float32)
    whole = ctx.vocode_mel(mel[None], np.array([150], np.int32))[0]
    h, hsd = voc_sd("v1")
    ref = O.hifigan_generator(mel.T, hsd, h)                                 # the oracle's whole-utterance waveform
    check_wav(whole, ref, prec, "whole-utterance wav", e2e=False)
    for cpc in (1, 3):
        parts = list(zv.vocode_stream(mel, chunk_frames=40, chunks_per_call=cpc))
        assert [len(p) for p in parts] == [40 * 256, 40 * 256, 40 * 256, 30 * 256]
        got = np.concatenate(parts)
        check_wav(got, ref, prec, f"streamed wav ({cpc} chunks per call) against the oracle", e2e=False)
        if prec == "bf16":
            assert np.array_equal(got, whole)
        else:
            check_f32(got, whole, "streamed wav", tol=1e-5)
    short = np.concatenate(list(zv.vocode_stream(mel, chunk_frames=40, halo=2)))
    assert np.abs(short - whole).max() > 1e-4                              # the halo is what makes it exact


@pytest.mark.parametrize("voc", ["v1", "tiny"])
def test_vocoder_batch_invariance_random_ragged(voc):
    """Random ragged batches through the vocoder: every utterance's waveform is bit-identical to the one it gets alone
    (tile walk of the persistent fused kernel, row tiles past an utterance's end, length tables, register-ring slab kernel)."""
    ctx = ctx_for("styletts", voc, "bf16")
    rng = np.random.default_rng(77)
    for trial in range(3):
        B = int(rng.integers(2, 12))
        P = rng.integers(1, 41, B).astype(np.int32)
        mel = np.zeros((B, int(P.max()), 80), np.float32)
        for b in range(B):
            mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
        wav = ctx.vocode_mel(mel, P)
        for b in rng.choice(B, size=min(B, 3), replace=False):
            alone = ctx.vocode_mel(mel[b:b + 1, :P[b]], P[b:b + 1])
            assert np.array_equal(wav[b, :P[b] * 256], alone[0, :P[b] * 256]), (trial, int(b), int(P[b]))
            assert not wav[b, P[b] * 256:].any()


@pytest.mark.parametrize("voc", ["v1", "tiny"])
def test_outputs_do_not_depend_on_context_history(voc):
    """The bytes handed back never depend on what earlier calls left in the context's reusable device buffers
    (include/zvx.h: wav row b = mel_len[b]*hop samples, then zeros up to the batch maximum; mel rows >= mel_len[b] zero).
    A LARGE batch runs first, then a ragged one on the same context, for the stand-alone vocoder and for synthesize()."""
    ctx = ctx_for("styletts", voc, "bf16")
    rng = np.random.default_rng(123)
    big = rng.standard_normal((12, 48, 80)).astype(np.float32)
    ctx.vocode_mel(big, np.full(12, 48, np.int32))                            # fills the reusable buffers with live samples
    P = np.array([30, 7, 19, 1], np.int32)
    mel = np.zeros((4, 30, 80), np.float32)
    for b in range(4):
        mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
    wav = ctx.vocode_mel(mel, P)
    for b in range(4):
        alone = ctx_for("styletts", voc, "bf16").vocode_mel(mel[b:b + 1, :P[b]], P[b:b + 1])
        assert np.array_equal(wav[b, :P[b] * 256], alone[0, :P[b] * 256])
        assert not wav[b, P[b] * 256:].any(), f"utt {b}: stale samples past mel_len*hop"
    # end-to-end: long utterances first, then a ragged batch; wav AND mel_out tails
    ph, pu, T, spk, dur = synthetic.batch(6, 24, 300, "uniform")
    ctx.synthesize(ph, pu, T, spk, dur, np.full(6, 200, np.int32))
    Ts = np.array([20, 3, 11], np.int32)
    ph2, pu2, _, spk2, dur2 = synthetic.batch(3, 20, 310, "uniform")
    for b in range(3):
        ph2[b, Ts[b]:] = 0; pu2[b, Ts[b]:] = 0; dur2[b, Ts[b]:] = 0
    out = ctx.synthesize(ph2, pu2, Ts, spk2, dur2, np.full(3, 64, np.int32))
    for b in range(3):
        ml = int(out["mel_len"][b])
        assert ml == int(dur2[b, :Ts[b]].sum())
        assert not out["wav"][b, ml * 256:].any(), f"utt {b}: stale wav tail"
        assert not out["mel"][b, ml:].any(), f"utt {b}: stale mel rows"
        solo = ctx.synthesize(ph2[b:b + 1, :Ts[b]], pu2[b:b + 1, :Ts[b]], Ts[b:b + 1], spk2[b:b + 1], dur2[b:b + 1, :Ts[b]], np.array([64], np.int32))
        assert np.array_equal(solo["wav"][0, :ml * 256], out["wav"][b, :ml * 256])
        assert np.array_equal(solo["mel"][0, :ml], out["mel"][b, :ml])
    # staged API after a bigger call: zvx_decode's mel_out tail
    mel_len, _, _, _ = ctx.encode(ph2, pu2, Ts, spk2, dur2)
    m = ctx.decode(3, int(mel_len.max()))
    for b in range(3):
        assert not m[b, int(mel_len[b]):].any()


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_pcm16_output_equals_scaled_float_output(prec):
    """ZVX_PCM16: the int16 row is (float_sample * 32760).astype(int16) of the float row, bit for bit (demo.py:29-35)."""
    ctx = ctx_for("styletts", "tiny", prec)
    rng = np.random.default_rng(9)
    P = np.array([17, 5], np.int32)
    mel = np.zeros((2, 17, 80), np.float32)
    for b in range(2):
        mel[b, :P[b]] = 2.0 * rng.standard_normal((P[b], 80)).astype(np.float32)
    f = ctx.vocode_mel(mel, P)
    i = ctx.vocode_mel(mel, P, pcm16=True)
    assert i.dtype == np.int16 and i.shape == f.shape
    assert np.array_equal(i, (f * np.float32(32760)).astype(np.int16))
    assert np.abs(i).max() > 100


def test_unfusable_resblock_shapes_fall_back_to_two_launches():
    """A HiFi-GAN config.json the reference accepts but the fused ResBlock kernels do not cover (dilation 9 with k = 11:
    conv1 halo 45 > the fused kernels' 32 rows; k = 5) must still run -- through the two-launch path -- and match the oracle."""
    h = {"resblock": "1", "upsample_rates": [8, 8, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4],
         "upsample_initial_channel": 128, "resblock_kernel_sizes": [5, 11], "resblock_dilation_sizes": [[1, 2], [1, 9]]}
    hsd = zw.hifigan_state_dict(h, 2)
    cfg, sd = tts_sd("styletts")
    man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
    ctx = _lib.Context(man, blob, 0)
    rng = np.random.default_rng(4)
    mel = rng.standard_normal((1, 14, 80)).astype(np.float32)
    wav = ctx.vocode_mel(mel, np.array([14], np.int32))
    check_wav(wav[0], O.hifigan_generator(mel[0].T, hsd, h), "bf16", "unfused shapes", e2e=False)
    ctx.close()


@pytest.mark.parametrize("voc", ["v1", "v2"])
def test_streaming_resblock_kernels_equal_per_pair_launches(voc):
    """resstream.hip (a whole ResBlock as one LDS-ring pipeline, C = 32 / 64) against the per-pair launches it replaces:
    bit-identical waveforms -- ragged batches, utterances of 1 frame, utterances long enough to be cut into several
    segments per workgroup (halo recomputation at segment starts), more segments than CUs."""
    ctx = ctx_for("styletts", voc, "bf16")
    rng = np.random.default_rng(31)
    try:
        for B, Pmax in ((3, 23), (2, 300), (1, 1), (40, 33), (1, 1100)):
            P = rng.integers(1, Pmax + 1, B).astype(np.int32); P[0] = Pmax
            mel = np.zeros((B, Pmax, 80), np.float32)
            for b in range(B):
                mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
            ctx.set_int("resstream", 0); ref = ctx.vocode_mel(mel, P)
            ctx.set_int("resstream", 1); got = ctx.vocode_mel(mel, P)
            assert np.isfinite(got).all() and np.array_equal(got, ref), (voc, B, Pmax)
    finally:
        ctx.set_int("resstream", 1)


def test_fused_f32_attention_of_the_encoder_matches_the_unfused_products():
    """attention.hip: attn_f32_kernel (one Q|K|V GEMM + one launch per layer) against V^T / score / P.V GEMMs + softmax: f32
    round-off apart (different summation order), for ragged batches, one key, T just over one / four 128-key tiles; and an
    utterance's encoder output does not depend on the batch it travels in (fixed key tiles masked by its own length)."""
    for prec in ("bf16", "f32"):
        ctx = ctx_for("styletts", "tiny", prec)
        rng = np.random.default_rng(3)
        try:
            for (B, Tmax) in ((1, 1), (3, 33), (4, 128), (3, 129), (1, 530)):
                ph, pu, T, spk, dur = synthetic.batch(B, Tmax, 11, "const7")
                T = rng.integers(1, Tmax + 1, B).astype(np.int32); T[0] = Tmax
                for b in range(B):
                    ph[b, T[b]:] = 0; pu[b, T[b]:] = 0; dur[b, T[b]:] = 0
                outs = {}
                for mode in (0, 1):
                    ctx.set_int("attn_f32", mode)
                    ctx.encode(ph, pu, T, spk, dur)
                    outs[mode] = ctx.fetch("encoder_out", (B, Tmax, 528)).copy()
                assert np.isfinite(outs[1]).all()
                for b in range(B):
                    assert np.abs(outs[0][b, :T[b]] - outs[1][b, :T[b]]).max() <= 2e-4 * max(1.0, np.abs(outs[0]).max()), (prec, B, Tmax, b)
                ctx.encode(ph[:1], pu[:1], T[:1], spk[:1], dur[:1])
                assert np.array_equal(ctx.fetch("encoder_out", (1, Tmax, 528))[0, :T[0]], outs[1][0, :T[0]]), (prec, B, Tmax)
        finally:
            ctx.set_int("attn_f32", 1)


def test_single_request_sizing_does_not_change_a_bit():
    """A single request takes shorter streaming-ResBlock segments, 64- / 128-row conv-slab tiles and 32-channel tiles for its
    one-row-tile GEMMs (zvx_set_int rs_seg_min / slab_small; the pair kernel declines it): launch geometry only -- the utterance
    is bit-identical to itself computed with the batch geometry, alone and inside a batch of 5."""
    ctx = ctx_for("styletts", "v1", "bf16")
    ph, pu, T, spk, dur = synthetic.batch(5, 64, 7, "const7")
    pad = np.full(5, 448, np.int32)
    try:
        ctx.set_int("rs_seg_min", -1); ctx.set_int("slab_small", 0)
        ref1 = ctx.synthesize(ph[:1], pu[:1], T[:1], spk[:1], dur[:1], pad[:1], want_mel=True)
        ref5 = ctx.synthesize(ph, pu, T, spk, dur, pad, want_mel=True)
        assert np.array_equal(ref1["wav"][0], ref5["wav"][0])
        for seg, small in ((0, 1), (0, 2), (512, 2)):
            ctx.set_int("rs_seg_min", seg); ctx.set_int("slab_small", small)
            one = ctx.synthesize(ph[:1], pu[:1], T[:1], spk[:1], dur[:1], pad[:1], want_mel=True)
            five = ctx.synthesize(ph, pu, T, spk, dur, pad, want_mel=True)
            assert np.array_equal(one["mel"], ref1["mel"]) and np.array_equal(one["wav"], ref1["wav"]), (seg, small)
            assert np.array_equal(five["mel"], ref5["mel"]) and np.array_equal(five["wav"], ref5["wav"]), (seg, small)
    finally:
        ctx.set_int("rs_seg_min", 0); ctx.set_int("slab_small", 2)


@pytest.mark.parametrize("voc,prec", [("v1", "bf16"), ("v1", "f32"), ("v2", "bf16"), ("v3", "bf16")])
def test_single_request_overlap_does_not_change_a_bit(voc, prec):
    """Small batches fuse InstanceNorm into one launch (norm_fuse_maxb), run the duration predictor on a second stream beside the
    pitch predictor (va_overlap_maxb) and run the non-final conv pairs of a vocoder stage's 2nd / 3rd ResBlock on streams of their
    own (voc_overlap_maxb).  Scheduling only: the FIRST call of a fresh context (buffers just allocated) and later calls equal the
    serial schedule bit for bit, for 1, 2 and 4 utterances, ragged."""
    from zerovox_amd import _lib
    cfg, sd = tts_sd("styletts"); h, hsd = voc_sd(voc)
    man, blob = pack.pack_model(cfg, sd, h, hsd, prec)
    ph, pu, T, spk, dur = synthetic.batch(4, 48, 31, "const7")
    T = np.array([48, 17, 33, 5], np.int32)
    for b in range(4): ph[b, T[b]:] = 0; pu[b, T[b]:] = 0; dur[b, T[b]:] = 0
    ser = _lib.Context(man, blob, 0)
    for k in ("norm_fuse_maxb", "va_overlap_maxb", "voc_overlap_maxb"): ser.set_int(k, 0)
    for B in (1, 2, 4):
        ref = ser.synthesize(ph[:B], pu[:B], T[:B], spk[:B], dur[:B], None, want_mel=True)
        fresh = _lib.Context(man, blob, 0)            # defaults: all three on
        for rep in range(3):
            got = fresh.synthesize(ph[:B], pu[:B], T[:B], spk[:B], dur[:B], None, want_mel=True)
            assert np.array_equal(got["mel"], ref["mel"]), (B, rep)
            assert all(np.array_equal(g, r) for g, r in zip(got["wav"], ref["wav"])), (B, rep)
        fresh.close()
    ser.close()


def test_batches_in_flight_on_two_contexts_equal_the_sequential_calls():
    """ZeroVox.synthesize_batches: batches alternate over two contexts from worker threads (batch i + 1's front end under batch
    i's vocoder); results come back in order and bit-identical to synthesize_batch, for ragged batches of different shapes."""
    from zerovox_amd.model import ZeroVox
    cfg, sd = tts_sd("styletts")
    h, hsd = voc_sd("v1")
    m = ZeroVox(cfg, sd, h, hsd, infer_device="cuda:0", precision="bf16")
    rng = np.random.default_rng(12)
    batches = []
    for i, (B, T) in enumerate(((4, 40), (1, 64), (7, 23), (3, 90), (5, 31))):
        ph, pu, Tl, spk, dur = synthetic.batch(B, T, 20 * i, "const7")
        Tl = rng.integers(1, T + 1, B).astype(np.int32); Tl[0] = T
        for b in range(B):
            ph[b, Tl[b]:] = 0; pu[b, Tl[b]:] = 0; dur[b, Tl[b]:] = 0
        batches.append(dict(phoneme=ph, puncts=pu, T=Tl, style_embed=spk, duration=dur))
    seq = [m.synthesize_batch(b["phoneme"], b["puncts"], b["T"], b["style_embed"], b["duration"], want_mel=False) for b in batches]
    for n in (2, 3):
        out = list(m.synthesize_batches(iter(batches), in_flight=n))
        assert len(out) == len(seq)
        for a, r in zip(out, seq):
            assert np.array_equal(a["mel_len"], r["mel_len"]) and np.array_equal(a["wav"], r["wav"])
    for c in m._more_ctx:
        c.close()
    m.ctx.close()


def test_benchmark_batch_is_bit_reproducible_over_many_runs():
    """The hand-scheduled kernels count their outstanding loads (s_waitcnt vmcnt(n)): an off-by-one is a RARE wrong block, not a
    wrong result every time (round 3: the k = 3 pair kernel let one LDS-DMA piece outlive the step barrier, a wrong 128-row
    block in ~1 % of runs).  60 runs of the benchmark vocoder batch and 20 of the whole step must agree bit for bit; the
    encoder's fused f32 attention likewise (same result for an utterance alone and in a batch)."""
    ctx = ctx_for("styletts", "v1", "bf16")
    ph, pu, T, spk, dur = synthetic.batch(32, 128, 192, "const7")
    pad_to = np.full(32, 896, np.int32)
    r0 = ctx.synthesize(ph, pu, T, spk, dur, pad_to, want_mel=True)
    for i in range(20):
        r = ctx.synthesize(ph, pu, T, spk, dur, pad_to, want_mel=True)
        assert np.array_equal(r["mel"], r0["mel"]) and np.array_equal(r["wav"], r0["wav"]), f"step run {i}"
    w0 = ctx.vocode_mel(r0["mel"], r0["mel_len"])
    for i in range(60):
        assert np.array_equal(ctx.vocode_mel(r0["mel"], r0["mel_len"]), w0), f"vocoder run {i}"
    one = ctx.synthesize(ph[5:6], pu[5:6], T[5:6], spk[5:6], dur[5:6], pad_to[5:6], want_mel=True)
    assert np.array_equal(one["mel"][0], r0["mel"][5])


@pytest.mark.parametrize("pcm16", [False, True])
def test_vocode_mel_with_input_stride_larger_than_longest_utterance(pcm16):
    """zvx_vocode_mel with Pmax (the input mel's row stride) > max(P): the zero tail of a row ends at max_b(P) * hop -- what
    include/zvx.h promises and the caller's wav_stride covers -- not at Pmax * hop.  Host output and device output (guard words
    behind every row and behind the buffer must survive)."""
    ctx = ctx_for("styletts", "tiny", "bf16")
    rng = np.random.default_rng(77)
    B, Pmax = 3, 40
    P = np.array([9, 17, 4], np.int32)
    mel = np.zeros((B, Pmax, 80), np.float32)
    for b in range(B):
        mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
    tight = ctx.vocode_mel(mel[:, :17], P, pcm16=pcm16)                      # reference: stride = longest utterance
    wide = ctx.vocode_mel(mel, P, pcm16=pcm16)                               # stride 40 > 17
    n = 17 * 256
    assert wide.shape == (B, Pmax * 256) and np.array_equal(wide[:, :n], tight[:, :n]) and not wide[:, n:].any()
    for b in range(B):
        assert np.abs(wide[b, :P[b] * 256].astype(np.float32)).max() > 0 and not wide[b, P[b] * 256:].any()
    # device output with a row stride of exactly max(P) * hop + 8 guard elements
    ss, dt = (2, np.int16) if pcm16 else (4, np.float32)
    stride = n + 8
    guard = np.full((B + 1, stride), 12345, dt)
    wav_d, mel_d = ctx.dev_alloc(guard.nbytes), ctx.dev_alloc(mel.nbytes)
    ctx.dev_from_host(wav_d, guard); ctx.dev_from_host(mel_d, mel)
    ctx.vocode_mel_device(mel_d, P, Pmax, wav_d, stride, pcm16=pcm16)
    got = ctx.dev_to_host(wav_d, (B + 1, stride), dt)
    assert np.array_equal(got[:B, :n], tight[:, :n]), "rows"
    assert (got[:B, n:] == 12345).all() and (got[B] == 12345).all(), "writes past max(P) * hop"
    ctx.dev_free(wav_d); ctx.dev_free(mel_d)


def test_streaming_pair_kernel_equals_two_conv_slab_launches():
    """pairstream.hip (conv1 + conv2 of a C = 128 ResBlock pair in one launch: T in an LDS ring, weights through register rings,
    x by LDS-DMA into an XOR-swizzled ring) against the two conv-slab launches it replaces: bit-identical waveforms for every
    kernel size (k = 3 / 7 / 11, all three running-sum modes) -- ragged batches, one-frame utterances, utterances cut into
    several segments per workgroup (segment seams: halo rows of T and x), more segments than CUs."""
    ctx = ctx_for("styletts", "v1", "bf16")
    rng = np.random.default_rng(37)
    try:
        for B, Pmax in ((3, 23), (2, 300), (1, 1), (40, 33), (1, 1100), (5, 70), (9, 420)):
            P = rng.integers(1, Pmax + 1, B).astype(np.int32); P[0] = Pmax
            mel = np.zeros((B, Pmax, 80), np.float32)
            for b in range(B):
                mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
            ctx.set_int("pairstream", -1); ref = ctx.vocode_mel(mel, P)       # no fused kernel at all for C = 128
            ctx.set_int("pairstream", 3); got = ctx.vocode_mel(mel, P)        # every pair of the C = 128 stage streamed, whatever the job size
            assert np.isfinite(got).all() and np.array_equal(got, ref), (B, Pmax)
            ctx.set_int("pairstream", 1); got = ctx.vocode_mel(mel, P)        # default: streamed from ~200 k rows up ((9, 420) is), two launches below
            assert np.isfinite(got).all() and np.array_equal(got, ref), (B, Pmax)
    finally:
        ctx.set_int("pairstream", 1)


def test_rccl_gather_path_on_a_one_rank_communicator():
    """The multi-GPU gather inside libzvx (dlopen of librccl, ncclCommInitRank, grouped ncclSend/ncclRecv on the communication
    stream, the device-side fence that orders a later synthesis behind the gather) exercised on ONE GPU through a real
    one-rank communicator; the 8-GPU run itself belongs to the driver."""
    cfg, sd = tts_sd("styletts")
    h, hsd = voc_sd("tiny")
    man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
    ctx = _lib.Context(man, blob, 0)
    try:
        cid = _lib.Context.comm_unique_id()
        assert len(cid) == 128 and any(cid)
        ctx.comm_init(cid, 0, 1)
        ph, pu, T, spk, dur = synthetic.batch(3, 10, 0, "uniform")
        L = int(dur.sum(axis=1).max()); N = L * 256
        ref = ctx.synthesize(ph, pu, T, spk, dur, np.full(3, 64, np.int32), want_mel=False)["wav"]
        wav_d, recv_d = ctx.dev_alloc(3 * N * 4), ctx.dev_alloc(3 * N * 4)
        for _ in range(3):                                                   # repeated: the next synthesis into wav_d queues behind the gather
            ctx.synthesize(ph, pu, T, spk, dur, np.full(3, 64, np.int32), want_mel=False, wav_device_ptr=wav_d, wav_stride=N, no_sync=True)
            ctx.comm_gather(wav_d, 3 * N * 4, recv_d, root=0, no_sync=True)
        ctx.comm_barrier()
        got = ctx.dev_to_host(recv_d, (3, N), np.float32)
        assert np.array_equal(got, ref)
        assert ctx.comm_max(1.25) == 1.25
        info = ctx.comm_info()                                               # what bench.py puts into the N > 1 line ("rccl": {...})
        assert info["world"] == 1 and info["comm_count"] == 1 and info["ranks_seen"] == 1 and info["version_code"] > 20000, info
        assert info["device_pci"][0] is not None and ":" in info["device_pci"][0], info
        ctx.dev_free(wav_d); ctx.dev_free(recv_d)
    finally:
        ctx.close()


def test_queued_calls_own_their_inputs_and_keep_their_order():
    """zvx_synthesize with forced durations, a device output and ZVX_NO_SYNC only queues work (include/zvx.h): the host inputs are
    copied into the context's pinned staging before the call returns.  Eight calls with different inputs are queued back to back,
    each call's host arrays are overwritten with garbage as soon as it returns, and only then is the stream drained: every
    waveform equals the synchronous call's, and mel_len is filled on return."""
    ctx = ctx_for("styletts", "v1", "bf16")
    n, B, T = 8, 3, 24
    cases, refs = [], []
    for i in range(n):
        ph, pu, Tl, spk, dur = synthetic.batch(B, T, 50 + 7 * i, "uniform")
        Tl = np.array([T, 5 + i, 13], np.int32)
        for b in range(B): ph[b, Tl[b]:] = 0; pu[b, Tl[b]:] = 0; dur[b, Tl[b]:] = 0
        cases.append((ph, pu, Tl, spk, dur))
        refs.append(ctx.synthesize(ph, pu, Tl, spk, dur, None, want_mel=False))
    N = max(int(r["mel_len"].max()) for r in refs) * 256
    bufs = [ctx.dev_alloc(B * N * 4) for _ in range(n)]
    try:
        for i, (ph, pu, Tl, spk, dur) in enumerate(cases):
            a = [x.copy() for x in (ph, pu, Tl, spk, dur)]
            r = ctx.synthesize(a[0], a[1], a[2], a[3], a[4], None, want_mel=False, wav_device_ptr=bufs[i], wav_stride=N, no_sync=True)
            assert np.array_equal(r["mel_len"], refs[i]["mel_len"]) and r["log_duration"] is None
            a[0][:] = 1; a[1][:] = 1; a[2][:] = 1; a[3][:] = 1e9; a[4][:] = 30            # the caller's arrays are free again
        ctx.sync()
        for i in range(n):
            got = ctx.dev_to_host(bufs[i], (B, N), np.float32)
            for b in range(B):
                k = int(refs[i]["mel_len"][b]) * 256
                assert np.array_equal(got[b, :k], refs[i]["wav"][b][:k]), (i, b)
    finally:
        for p in bufs: ctx.dev_free(p)


@pytest.mark.parametrize("kind", ["styletts", "fastspeech2"])
def test_front_end_under_the_previous_vocoder_equals_the_serial_schedule(kind):
    """zvx_synthesize issues encoder / variance adaptor / mel decoder on the context's front stream and the vocoder on the main
    stream (`front_overlap`, default 1): queued calls run call i+1's front end under call i's vocoder.  Ten queued calls of
    changing shape (so that work buffers are re-cut and re-grown), with staged-API calls (zvx_encode / zvx_decode / zvx_vocode on
    the main stream) and a zvx_vocode_mel in between, must equal the serial schedule (`front_overlap 0`, every call waited for)
    bit for bit -- waveform, mel and mel_len."""
    ctx = ctx_for(kind, "v1", "bf16")
    shapes = [(6, 64), (3, 24), (8, 48), (6, 64), (1, 17), (5, 64), (6, 64), (2, 90), (6, 33), (6, 64)]
    cases, refs = [], []
    rng = np.random.default_rng(77)
    try:
        ctx.set_int("front_overlap", 0)
        for i, (B, T) in enumerate(shapes):
            ph, pu, Tl, spk, dur = synthetic.batch(B, T, 300 + 11 * i, "uniform")
            Tl = rng.integers(max(1, T // 2), T + 1, B).astype(np.int32); Tl[0] = T
            for b in range(B): ph[b, Tl[b]:] = 0; pu[b, Tl[b]:] = 0; dur[b, Tl[b]:] = 0
            cases.append((ph, pu, Tl, spk, dur))
            refs.append(ctx.synthesize(ph, pu, Tl, spk, dur, None, want_mel=True))
        staged_in = synthetic.batch(2, 20, 900, "uniform")
        staged_ml, _, _, _ = ctx.encode(staged_in[0], staged_in[1], staged_in[2], staged_in[3], staged_in[4])
        staged_mel = ctx.decode(2, int(staged_ml.max())).copy()
        staged_wav = ctx.vocode(2, staged_ml, None).copy()
        vm_mel = rng.standard_normal((2, 30, 80)).astype(np.float32); vm_P = np.array([30, 11], np.int32)
        vm_ref = ctx.vocode_mel(vm_mel, vm_P).copy()
        # the two-stream schedule forced on calls that wait for their result: same bits again
        ctx.set_int("front_overlap", 2)
        for i in (0, 4, 7):
            r2 = ctx.synthesize(*cases[i], None, want_mel=True)
            assert np.array_equal(r2["wav"], refs[i]["wav"]) and np.array_equal(r2["mel"], refs[i]["mel"]), i
    finally:
        ctx.set_int("front_overlap", 1)
    Ns = [int(r["mel_len"].max()) * 256 for r in refs]
    Ls = [int(r["mel_len"].max()) for r in refs]
    bufs = [ctx.dev_alloc(shapes[i][0] * Ns[i] * 4) for i in range(len(shapes))]
    mbufs = [ctx.dev_alloc(shapes[i][0] * Ls[i] * 80 * 4) for i in range(len(shapes))]
    try:
        for i, (ph, pu, Tl, spk, dur) in enumerate(cases):
            r = ctx.synthesize(ph, pu, Tl, spk, dur, None, want_mel=True, wav_device_ptr=bufs[i], wav_stride=Ns[i], mel_device_ptr=mbufs[i], no_sync=True)
            assert np.array_equal(r["mel_len"], refs[i]["mel_len"])
            if i == 3:                                   # staged API on the main stream between two queued calls
                ml, _, _, _ = ctx.encode(staged_in[0], staged_in[1], staged_in[2], staged_in[3], staged_in[4])
                assert np.array_equal(ml, staged_ml)
                assert np.array_equal(ctx.decode(2, int(ml.max())), staged_mel)
                assert np.array_equal(ctx.vocode(2, ml, None), staged_wav)
            if i == 6:
                assert np.array_equal(ctx.vocode_mel(vm_mel, vm_P), vm_ref)
        ctx.sync()
        for i, (B, T) in enumerate(shapes):
            got = ctx.dev_to_host(bufs[i], (B, Ns[i]), np.float32)
            gmel = ctx.dev_to_host(mbufs[i], (B, Ls[i], 80), np.float32)
            for b in range(B):
                k = int(refs[i]["mel_len"][b])
                assert np.array_equal(got[b, :k * 256], refs[i]["wav"][b][:k * 256]), (i, b)
                assert np.array_equal(gmel[b, :k], refs[i]["mel"][b][:k]), (i, b)
    finally:
        for p_ in bufs + mbufs: ctx.dev_free(p_)


# ------------------------------------------------------------------------------------------------
# BASELINE.json configs at their full sizes
# ------------------------------------------------------------------------------------------------
def test_config4_hifigan_v1_alone_1024_frames():
    """configs[3]: HiFi-GAN V1 alone on a [80, 1024] N(0,1) mel (seed 7): B = 1 against the oracle (f32 exact-class and bf16),
    B = 32 through size-independent properties (every row of a batch of identical mels equals the B = 1 waveform bit for bit;
    distinct rows stay finite, bounded and different)."""
    h, hsd = voc_sd("v1")
    mel = np.random.default_rng(7).standard_normal((1024, 80)).astype(np.float32)
    ref = O.hifigan_generator(mel.T, hsd, h)
    assert ref.shape == (262144,)
    for prec in ("f32", "bf16"):
        ctx = ctx_for("styletts", "v1", prec)
        wav = ctx.vocode_mel(mel[None], np.array([1024], np.int32))
        check_wav(wav[0], ref, prec, f"config4 B=1 {prec}", e2e=False)
    ctx = ctx_for("styletts", "v1", "bf16")
    one = ctx.vocode_mel(mel[None], np.array([1024], np.int32))[0]
    rng = np.random.default_rng(70)
    batch = rng.standard_normal((32, 1024, 80)).astype(np.float32)
    batch[5] = mel; batch[31] = mel
    out = ctx.vocode_mel(batch, np.full(32, 1024, np.int32))
    assert out.shape == (32, 262144) and np.isfinite(out).all() and np.abs(out).max() <= 1.0
    assert np.array_equal(out[5], one) and np.array_equal(out[31], one)
    assert out[0].std() > 1e-3 and not np.array_equal(out[0], out[1])


def test_vocoder_in_ieee_half_against_bf16_and_the_oracle():
    """The generator's 16-bit arithmetic per stage: voc_f16 = 0 runs the bf16 kernels of rounds 1-4 everywhere, voc_f16_stages = 31 IEEE half
    everywhere (round 5), the default (voc_f16_stages -1) half everywhere but V1's 128-channel stage (round 6).  On a 256-frame N(0,1) mel all
    three meet the oracle within their own limits; all-half is at least 3x closer than all-bf16 (measured ~8x: 11 significand bits against
    8), the default sits between the two and at least 1.5x closer than all-bf16."""
    h, hsd = voc_sd("v1")
    ctx = ctx_for("styletts", "v1", "bf16")
    mel = np.random.default_rng(21).standard_normal((256, 80)).astype(np.float32)
    ref = O.hifigan_generator(mel.T, hsd, h)
    P1 = np.array([256], np.int32)
    try:
        ctx.set_int("voc_f16", 0); wb = ctx.vocode_mel(mel[None], P1)[0]
        ctx.set_int("voc_f16", 1); ctx.set_int("voc_f16_stages", 31); wh = ctx.vocode_mel(mel[None], P1)[0]
        ctx.set_int("voc_f16_stages", 0b11011); wm = ctx.vocode_mel(mel[None], P1)[0]
        ctx.set_int("voc_f16_stages", -1); wd = ctx.vocode_mel(mel[None], P1)[0]
    finally:
        ctx.set_int("voc_f16", 1); ctx.set_int("voc_f16_stages", -1)
    assert np.array_equal(wd, wm)                        # the default IS "all but stage 2" on this generator
    eb, eh, em = stats(wb, ref), stats(wh, ref), stats(wd, ref)
    _errlog("wav-voc", "voc_f16 A/B bf16", eb[0], eb[1]); _errlog("wav-voc", "voc_f16 A/B half", eh[0], eh[1]); _errlog("wav-voc", "voc_f16 A/B default", em[0], em[1])
    assert eb[0] <= 1e-2 and eb[1] <= 2e-3, f"bf16 vocoder: {eb[:2]}"
    check_wav(wh, ref, "bf16", "half vocoder", e2e=False, voc="all-half")
    check_wav(wd, ref, "bf16", "default vocoder", e2e=False)
    assert eh[1] * 3 <= eb[1], f"half rms {eh[1]:.3e} is not 3x below bf16 rms {eb[1]:.3e}"
    assert eh[1] <= em[1] and em[1] * 1.5 <= eb[1], f"default rms {em[1]:.3e} (half {eh[1]:.3e}, bf16 {eb[1]:.3e})"


@pytest.mark.parametrize("voc", ["v1", "v2", "v3"])
def test_half_vocoder_saturates_instead_of_overflowing(voc):
    """IEEE half tops out at 65504.  Mels scaled by 16 and by 256 drive the activations of the wide stages far past that; every 16-bit
    store of the vocoder saturates (MODE.FP16_OVFL in the kernels, v_med3 in the run-time epilogues), so the waveform stays finite and
    inside [-1, 1] -- no Inf, no NaN from Inf - Inf -- and at the nominal scale the result is untouched by the clamps (oracle check)."""
    h, hsd = voc_sd(voc)
    ctx = ctx_for("styletts", voc, "bf16")
    P = np.array([40, 33], np.int32)
    rng = np.random.default_rng(31)
    mel = np.zeros((2, 40, 80), np.float32)
    for b in range(2):
        mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
    w1 = ctx.vocode_mel(mel, P)
    check_wav(w1[0, :P[0] * 256], O.hifigan_generator(mel[0, :P[0]].T, hsd, h), "bf16", f"{voc} x1", e2e=False)
    for scale in (16.0, 256.0, 4096.0):
        w = ctx.vocode_mel(mel * scale, P)
        assert np.isfinite(w).all() and np.abs(w).max() <= 1.0, f"{voc} x{scale}: non-finite or out-of-range samples"
        assert np.abs(w[0, :P[0] * 256]).max() > 0.0
        assert not w[1, P[1] * 256:].any()


@pytest.mark.parametrize("voc", ["v1", "v2", "v3"])
def test_half_mode_saturation_audit_counts_clamped_stores(voc):
    """Round 6: the half mode's clamp at +-65504 has a telltale.  zvx_set_int("f16_sat_check", 1) runs every convolution of the
    vocoder and the mel decoders as its own launch (no LDS-resident intermediate) and counts the clamped values in every 16-bit tensor it
    writes; zvx_get_int("f16_sat_events") reads the count.  At the nominal scale it is 0 -- on the vocoder alone and on a whole synthesis
    call -- and the audit's waveform meets the oracle like the default path's; a mel scaled by 2^18 (values up to ~1e6: past half's range
    already in the generator's padded input) clamps and is counted, the waveform stays finite; re-arming zeroes the counter; the bf16
    kernels (voc_f16 0) never count."""
    h, hsd = voc_sd(voc)
    ctx = ctx_for("styletts", voc, "bf16")
    P = np.array([40, 33], np.int32)
    rng = np.random.default_rng(31)
    mel = np.zeros((2, 40, 80), np.float32)
    for b in range(2):
        mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
    assert ctx.get_int("f16_sat_check") == 0
    try:
        ctx.set_int("f16_sat_check", 1)
        assert ctx.get_int("f16_sat_events") == 0
        w1 = ctx.vocode_mel(mel, P)
        assert ctx.get_int("f16_sat_events") == 0, "clamped stores at the nominal scale"
        check_wav(w1[0, :P[0] * 256], O.hifigan_generator(mel[0, :P[0]].T, hsd, h), "bf16", f"{voc} audit path x1", e2e=False)
        ph, pu, Tl, spk, dur = synthetic.batch(2, 16, 5, "uniform")
        ctx.synthesize(ph, pu, Tl, spk, dur, None)
        assert ctx.get_int("f16_sat_events") == 0, "clamped stores in a whole synthesis call at the nominal scale"
        w = ctx.vocode_mel(mel * 4096.0, P)
        assert ctx.get_int("f16_sat_events") >= 0 and np.isfinite(w).all()      # (seeded weights are variance-preserving: x 4096 need not clamp)
        w = ctx.vocode_mel(mel * 2.0 ** 18, P)
        nbig = ctx.get_int("f16_sat_events")
        assert nbig > 0 and np.isfinite(w).all(), nbig
        ctx.set_int("f16_sat_check", 1)                      # re-armed: counter back to zero
        assert ctx.get_int("f16_sat_events") == 0
        ctx.set_int("voc_f16", 0)
        ctx.vocode_mel(mel * 2.0 ** 18, P)
        assert ctx.get_int("f16_sat_events") == 0             # bf16 tensors are not half: nothing to clamp, nothing counted
    finally:
        ctx.set_int("voc_f16", 1)
        ctx.set_int("f16_sat_check", 0)
    assert np.array_equal(ctx.vocode_mel(mel, P), ctx.vocode_mel(mel, P))


def test_asynchronous_host_delivery_equals_the_synchronous_call():
    """Round 6 (synthesize.py:233-239 hands the caller host memory): zvx_synthesize with ZVX_HOST_ASYNC only queues work -- the
    waveform rows travel to one of the context's two pinned host slots on its copy stream -- and zvx_wait_host hands them out.  Seven calls
    with different inputs and shapes are queued, each call's host arrays are overwritten as soon as it returns, the host takes delivery of
    call i - 1 after queueing call i (the pipelined pattern of `bench.py --host-out`): every waveform equals the synchronous call's bit
    for bit; slots alternate; int16 PCM rows too; a host mel output is refused."""
    ctx = ctx_for("styletts", "v1", "bf16")
    shapes = [(3, 24), (2, 40), (3, 24), (1, 9), (4, 31), (3, 24), (2, 17)]
    cases, refs = [], []
    for i, (B, T) in enumerate(shapes):
        ph, pu, Tl, spk, dur = synthetic.batch(B, T, 70 + 5 * i, "uniform")
        Tl = np.array([T] + [max(1, T - 3 * (b + i % 3)) for b in range(1, B)], np.int32)
        for b in range(B): ph[b, Tl[b]:] = 0; pu[b, Tl[b]:] = 0; dur[b, Tl[b]:] = 0
        cases.append((ph, pu, Tl, spk, dur))
        refs.append(ctx.synthesize(ph, pu, Tl, spk, dur, None, want_mel=False, pcm16=(i == 4)))
    got, prev = {}, None
    for i, (ph, pu, Tl, spk, dur) in enumerate(cases):
        a = [x.copy() for x in (ph, pu, Tl, spk, dur)]
        r = ctx.synthesize(a[0], a[1], a[2], a[3], a[4], None, want_mel=False, host_async=True, pcm16=(i == 4))
        assert r["slot"] == i % 2 and r["wav"] is None and np.array_equal(r["mel_len"], refs[i]["mel_len"])
        a[0][:] = 1; a[1][:] = 1; a[2][:] = 1; a[3][:] = 1e9; a[4][:] = 30
        if prev is not None:
            got[prev[0]] = ctx.wait_host(prev[1], pcm16=(prev[0] == 4)).copy()
        prev = (i, r["slot"])
    got[prev[0]] = ctx.wait_host(prev[1]).copy()
    for i, (B, T) in enumerate(shapes):
        n = int(refs[i]["mel_len"].max()) * 256
        assert got[i].shape == (B, n) and got[i].dtype == refs[i]["wav"].dtype, (i, got[i].shape, got[i].dtype)
        assert np.array_equal(got[i], refs[i]["wav"][:, :n]), i
    with pytest.raises(_lib.ZvxError):
        ctx.synthesize(*cases[0], None, want_mel=True, host_async=True)
    # predicted durations (the call waits once, for the mel lengths; the waveform still travels asynchronously) and the stand-alone vocoder
    ph, pu, Tl, spk, dur = cases[0]
    ref_p = ctx.synthesize(ph, pu, Tl, spk, None, None, want_mel=False, Lmax_cap=400)
    rp = ctx.synthesize(ph, pu, Tl, spk, None, None, want_mel=False, Lmax_cap=400, host_async=True)
    assert np.array_equal(rp["mel_len"], ref_p["mel_len"])
    n_p = int(ref_p["mel_len"].max()) * 256
    assert np.array_equal(ctx.wait_host(rp["slot"]), ref_p["wav"][:, :n_p])
    rng = np.random.default_rng(5)
    vm = rng.standard_normal((2, 30, 80)).astype(np.float32); vP = np.array([30, 11], np.int32)
    ref_v = ctx.vocode_mel(vm, vP)
    assert np.array_equal(ctx.wait_host(ctx.vocode_mel(vm, vP, host_async=True)), ref_v)
    # the staged API and a synchronous call between two queued ones
    r = ctx.synthesize(*cases[1], None, want_mel=False, host_async=True)
    mid = ctx.synthesize(*cases[2], None, want_mel=False)
    assert np.array_equal(mid["wav"], refs[2]["wav"])
    assert np.array_equal(ctx.wait_host(r["slot"]), refs[1]["wav"][:, :int(refs[1]["mel_len"].max()) * 256])


def test_vocoder_v2_full_size_ragged_batch_against_the_oracle():
    """HiFi-GAN V2 (the reference's default vocoder, model.py:84) at the benchmark's utterance length on a ragged batch, every utterance
    against its own batch-1 oracle call.  Its last two stages (C = 16 / 8) run as ONE launch per stage (narrowstage.hip: tiles of 384 /
    512 rows with 64 halo rows, utterance ends inside / at / just past tile boundaries here); zvx_set_int("stagefuse", 0) -- the per-pair
    kernels of rounds 1-4 -- must agree with it to rounding (the stage kernel keeps the running sum in f32 registers instead of a
    16-bit tensor: not bit-identical by design) and lands no closer to the oracle."""
    h, hsd = voc_sd("v2")
    ctx = ctx_for("styletts", "v2", "bf16")
    P = np.array([896, 513, 384 // 2, 3 * 512 // 4 + 1, 57], np.int32)
    rng = np.random.default_rng(61)
    mel = np.zeros((len(P), int(P.max()), 80), np.float32)
    for b in range(len(P)):
        mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
    wav = ctx.vocode_mel(mel, P)
    try:
        ctx.set_int("stagefuse", 0); pairs = ctx.vocode_mel(mel, P)
    finally:
        ctx.set_int("stagefuse", 1)
    for b in range(len(P)):
        ref = O.hifigan_generator(mel[b, :P[b]].T, hsd, h)
        check_wav(wav[b, :P[b] * 256], ref, "bf16", f"v2 utt {b} (stage kernel)", e2e=False)
        check_wav(pairs[b, :P[b] * 256], ref, "bf16", f"v2 utt {b} (pair kernels)", e2e=False)
        assert not wav[b, P[b] * 256:].any()
        es, ep = stats(wav[b, :P[b] * 256], ref), stats(pairs[b, :P[b] * 256], ref)
        assert es[1] <= ep[1] * 1.05, f"utt {b}: stage kernel rms {es[1]:.3e} vs pair kernels {ep[1]:.3e}"
    alone = ctx.vocode_mel(mel[1:2, :P[1]], P[1:2])
    assert np.array_equal(alone[0], wav[1, :P[1] * 256])                 # an utterance alone == inside the batch, bit for bit


def test_non_finite_mel_values_stay_inside_their_utterance():
    """zvx_vocode_mel with a NaN / Inf in ONE utterance of a batch (gemm.hip, pairstream.hip and resstream.hip are compiled with
    -fno-honor-nans: min / max of the leaky-relu forms do not propagate NaN the IEEE way).  Defined behaviour (include/zvx.h): no fault,
    the call succeeds, the samples of the poisoned utterance are unspecified (finite or not), every OTHER utterance of the batch is bit
    for bit what it is without the poison, and a later clean call on the same context is unaffected."""
    ctx = ctx_for("styletts", "v1", "bf16")
    P = np.array([48, 48, 31], np.int32)
    rng = np.random.default_rng(41)
    mel = np.zeros((3, 48, 80), np.float32)
    for b in range(3):
        mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
    clean = ctx.vocode_mel(mel, P)
    bad = mel.copy()
    bad[1, 10, 3] = np.nan; bad[1, 20, 7] = np.inf; bad[1, 30, 11] = -np.inf
    out = ctx.vocode_mel(bad, P)
    assert np.array_equal(out[0], clean[0]) and np.array_equal(out[2], clean[2])
    assert not out[2, P[2] * 256:].any()
    again = ctx.vocode_mel(mel, P)
    assert np.array_equal(again, clean)


def test_non_finite_mel_values_stay_inside_their_utterance_in_the_narrow_stage_kernel():
    """The same contract on HiFi-GAN V2, whose last two stages run on narrowstage.hip with PERSISTENT workgroups: 24 utterances
    of 48 frames are 384 / 576 tiles on 256 / 512 workgroups, so a workgroup moves from a tile of utterance b to a tile of utterance
    b + 16 (b + 21) with its LDS streams still holding the first one's rows -- a NaN / Inf mel in utterances 1 and 2 must not reach any
    other utterance through the zero-weight padded tap slots of a convolution's last matrix step (0 x NaN)."""
    ctx = ctx_for("styletts", "v2", "bf16")
    B = 24
    P = np.full(B, 48, np.int32); P[5] = 31; P[20] = 40
    rng = np.random.default_rng(43)
    mel = np.zeros((B, 48, 80), np.float32)
    for b in range(B):
        mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
    clean = ctx.vocode_mel(mel, P)
    bad = mel.copy()
    bad[1, :, :] = np.nan; bad[2, 7:40, 3::5] = np.inf; bad[2, 11:30, 1::7] = -np.inf
    out = ctx.vocode_mel(bad, P)
    for b in range(B):
        if b not in (1, 2):
            assert np.array_equal(out[b], clean[b]), f"utterance {b} changed by a non-finite mel in utterances 1 / 2"
    assert np.array_equal(ctx.vocode_mel(mel, P), clean)


def test_config5_speaker_encoder_1000_clips():
    """configs[4]: 1000 x [258, 80] N(0,1) reference mels in batches of 50: unit norm everywhere, 8 sampled clips against the
    oracle, and batch invariance (clip i in its batch == clip i alone, to f32 round-off)."""
    ctx = ctx_for("styletts", "tiny", "bf16")
    cfg, sd = tts_sd("styletts")
    rng = np.random.default_rng(8)
    mels = rng.standard_normal((1000, 258, 80)).astype(np.float32)
    lens = np.full(50, 258, np.int32)
    emb = np.concatenate([ctx.spkemb(mels[i:i + 50], lens) for i in range(0, 1000, 50)])
    assert emb.shape == (1000, 528) and np.isfinite(emb).all()
    assert np.abs(np.linalg.norm(emb, axis=1) - 1.0).max() < 1e-4
    for i in (0, 49, 50, 333, 512, 777, 950, 999):
        check_embed16(emb[i], O.resnet_se34v2(mels[i], sd, cfg), f"clip {i}")
    solo = ctx.spkemb(mels[333:334], np.array([258], np.int32))
    assert np.abs(solo[0] - emb[333]).max() < 1e-6                      # tile shapes follow the batch size: equal to f32 round-off, not bit for bit


def test_config3_workload_on_one_gpu_256_utterances_equal_eight_shards():
    """configs[2] shards 256 utterances as 8 x 32 over the GPUs of a node.  On one GPU: the 256-utterance batch in ONE call
    is bit-identical to the eight 32-utterance shard calls (what each rank computes) -- utterances never interact, so the
    gathered result of the 8-GPU job is exactly this tensor."""
    ctx = ctx_for("styletts", "v1", "bf16")
    ph, pu, T, spk, dur = synthetic.batch(256, 128, 0, "const7")
    pad_to = np.full(256, 896, np.int32)
    whole = ctx.synthesize(ph, pu, T, spk, dur, pad_to, want_mel=False)["wav"]
    assert whole.shape == (256, 229376) and np.isfinite(whole).all()
    for r in range(8):
        sl = slice(32 * r, 32 * r + 32)
        shard = ctx.synthesize(ph[sl], pu[sl], T[sl], spk[sl], dur[sl], pad_to[sl], want_mel=False)["wav"]
        assert np.array_equal(shard, whole[sl]), f"shard {r}"


@pytest.mark.parametrize("kind", ["styletts", "fastspeech2"])
def test_config2_secondary_workload_variable_lengths_full_size(kind):
    """configs[1], secondary input of SURVEY 8(d): 32 x 128 phonemes with forced durations ~ U{3..10} (each utterance its own
    length, 770-890 frames).  At this size the oracle is out of reach; the size-independent property is that an utterance does
    not see its batch: sampled utterances equal their own batch-1 calls bit for bit (waveform, mel, mel_len), tails are zero."""
    ctx = ctx_for(kind, "v1", "bf16")
    ph, pu, T, spk, dur = synthetic.batch(32, 128, 0, "uniform")
    L = dur.sum(1).astype(np.int32)
    assert L.min() >= 600 and L.max() <= 1280 and len(set(L.tolist())) > 16
    pad_to = np.maximum(689, L).astype(np.int32)                          # the fresh-model value of model.py:331-335 per utterance
    out = ctx.synthesize(ph, pu, T, spk, dur, pad_to, want_mel=True)
    wav, mel, ml = out["wav"], out["mel"], out["mel_len"]
    assert np.array_equal(ml, pad_to) and np.isfinite(wav).all()
    for b in range(32):
        assert not wav[b, int(ml[b]) * 256:].any() and not mel[b, int(ml[b]):].any()
        assert np.abs(wav[b, :int(L[b]) * 256]).max() > 1e-3
    for b in (0, 7, 19, 31):
        solo = ctx.synthesize(ph[b:b + 1], pu[b:b + 1], T[b:b + 1], spk[b:b + 1], dur[b:b + 1], pad_to[b:b + 1], want_mel=True)
        n = int(ml[b])
        assert int(solo["mel_len"][0]) == n
        assert np.array_equal(solo["wav"][0, :n * 256], wav[b, :n * 256]), f"utt {b}: waveform depends on the batch"
        assert np.array_equal(solo["mel"][0, :n], mel[b, :n]), f"utt {b}: mel depends on the batch"


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_reference_written_checkpoint_reproduces_the_reference_output(prec, tmp_path):
    """SURVEY 8 f-2 end to end on a checkpoint the REFERENCE wrote (tests/golden/gen_ref_checkpoint.py, reduced-width model: hidden 32,
    ~2 MB): the model directory as the reference lays it out (modelcfg.yaml + checkpoints/*.ckpt with pickled hyper_parameters holding
    the real `Symbols` object, a generator baked in under `_meldec.` after remove_weight_norm()) -> ZeroVoxTTS.load_model, directly and
    through tools/convert_checkpoint.py -> inference_ex on the GPU == the reference's own inference_ex output stored next to it, with
    forced and with predicted durations; the baked-in generator overrides the external one (other seed)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import convert_checkpoint as cc
    from zerovox_amd.synthesize import ZeroVoxTTS
    d = os.path.join(GOLDEN, "refckpt")
    g = np.load(os.path.join(GOLDEN, "refckpt_expected.npz"))
    ck = os.path.join(d, "checkpoints", os.listdir(os.path.join(d, "checkpoints"))[0])
    cc.convert_tts(ck, os.path.join(d, "modelcfg.yaml"), str(tmp_path / "tts"))
    x = {"phoneme": g["phoneme"][None], "puncts": g["puncts"][None], "duration": g["duration"][None]}
    outs = []
    for path in (d, str(tmp_path / "tts")):
        _, synth = ZeroVoxTTS.load_model(path, "synthetic:tiny3:5", infer_device="cuda:0", precision=prec)
        for tag, forced in (("forced", True), ("pred", False)):
            synth._model._min_mel_len = 689
            wav, ml, logd, mel = synth._model.inference_ex(x, g["spk"][None, None], force_duration=forced)
            assert ml == int(g[tag + "_mel_len"]), f"{tag}: mel_len {ml} != {int(g[tag + '_mel_len'])}"
            check_f32(logd[0], g[tag + "_log_duration"], f"{tag}: log_duration")
            check_mel(mel.T, g[tag + "_mel"].T, prec, f"{tag}: mel", "styletts")
            check_wav(wav, g[tag + "_wav"], prec, f"{tag}: wav", voc="tiny3")
            outs.append(wav)
        synth._model.close()
    assert np.array_equal(outs[0], outs[2]) and np.array_equal(outs[1], outs[3])             # converted directory == reference layout read directly


def test_demo_cli_prints_the_reference_rtf_lines(capsys, monkeypatch, tmp_path):
    """zerovox_amd.demo mirrors demo.py --iter (demo.py:99-138): per-iteration line, warm-up of 10 discarded, mean RTF."""
    import re
    import sys
    import wave
    from zerovox_amd import demo
    out_wav = tmp_path / "demo.wav"
    monkeypatch.setattr(sys, "argv", ["demo", "--model", "synthetic:styletts", "--meldec-model", "synthetic:tiny", "--precision", "bf16",
                                      "--iter", "13", "--wav-filename", str(out_wav), "hello world, this is a test."])
    demo.main()
    lines = capsys.readouterr().out.splitlines()
    its = [l for l in lines if re.match(r"^\[\d+/13\] Synth time: \d+\.\d\d sec, voice length: \d+\.\d\d sec, rtf: \d+\.\d\d$", l)]
    assert len(its) == 13 and lines[0] == "computing speaker embedding..." and re.match(r"^Average RTF: \d+\.\d\d$", lines[-1])
    with wave.open(str(out_wav), "rb") as w:
        assert w.getframerate() == 22050 and w.getnchannels() == 1 and w.getsampwidth() == 2 and w.getnframes() > 0


def test_fused_attention_of_the_fs2_decoder():
    """attention.hip (softmax(Q K^T / sqrt(d)) V in one launch, online softmax, no [L][L] tensor) against the score-GEMM +
    softmax + PV-GEMM path it replaces (both bf16: equal to bf16 rounding) and against the oracle; ragged lengths down to 2 frames,
    lengths that are not multiples of the 32-key tile or the 128-query tile, and more than one query tile."""
    ctx = ctx_for("fastspeech2", "tiny", "bf16")
    cfg, sd = tts_sd("fastspeech2")
    rng = np.random.default_rng(41)
    L = np.array([50, 2, 130, 33, 257], np.int32)
    feats = np.zeros((5, 257, 528), np.float32)
    spk = rng.standard_normal((5, 528)).astype(np.float32); spk /= np.linalg.norm(spk, axis=1, keepdims=True)
    for b in range(5):
        feats[b, :L[b]] = rng.standard_normal((L[b], 528)).astype(np.float32)
    try:
        ctx.set_int("flash", 0); unfused = ctx.decode_features(feats, L, spk)
        ctx.set_int("flash", 1); fused = ctx.decode_features(feats, L, spk)
    finally:
        ctx.set_int("flash", 1)
    for b in range(5):
        ref = O.fs2_decoder(feats[b, :L[b]], spk[b], sd, cfg)
        check_mel(fused[b, :L[b]], ref, "bf16", f"fused attention utt {b}")
        check_mel(unfused[b, :L[b]], ref, "bf16", f"unfused attention utt {b}", "bf16")       # flash 0: the block falls back to bf16
        # the two paths against each other: no looser than the sum of their own stated limits against the oracle (check_mel: 2e-2 half, 4e-2 bf16)
        assert np.abs(fused[b, :L[b]] - unfused[b, :L[b]]).max() <= 2e-2 + 4e-2
        assert not fused[b, L[b]:].any()


def test_speaker_encoder_sap_pooling_against_reference_golden():
    """encoder_type 'SAP' against the fixture the REFERENCE produced with that option (tests/golden/gen_golden.py)."""
    import copy
    g = np.load(os.path.join(GOLDEN, "spkemb_sap_T96.npz"))
    cfg = copy.deepcopy(zcfg.medium_modelcfg("styletts"))
    cfg["model"]["resnet"]["encoder_type"] = "SAP"
    sd = zw.tts_state_dict(cfg, 0)
    h, hsd = voc_sd("tiny")
    for prec in ("f32", "bf16"):
        man, blob = pack.pack_model(cfg, sd, h, hsd, prec)
        ctx = _lib.Context(man, blob, 0)
        try:
            e = ctx.spkemb(g["ref_mel"][None], np.array([g["ref_mel"].shape[0]], np.int32))[0]
            assert abs(np.linalg.norm(e) - 1.0) < 1e-3
            if prec == "f32":
                check_f32(e, g["embed"], "SAP embed", 5e-5)
            else:
                check_embed16(e, g["embed"], "SAP embed (16-bit mode)")
        finally:
            ctx.close()


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_vocoder_v3_against_reference_golden(prec):
    """HiFi-GAN V3 at its published width (ResBlock2 with dilations up to 12) against the reference's own waveform."""
    g = np.load(os.path.join(GOLDEN, "blocks_hifigan_v3.npz"))
    ctx = ctx_for("styletts", "v3", prec)
    mel = g["mel"].T[None]
    wav = ctx.vocode_mel(mel, np.array([mel.shape[1]], np.int32))[0]
    check_wav(wav, g["wav"], prec, "V3 wav", e2e=False)


def test_speaker_encoder_sap_pooling():
    """encoder_type 'SAP' (ResNetSE34V2.py:135-143, 199-200: attention-weighted mean only, fc over 2560 inputs)."""
    import copy
    cfg = copy.deepcopy(zcfg.medium_modelcfg("styletts"))
    cfg["model"]["resnet"]["encoder_type"] = "SAP"
    sd = zw.tts_state_dict(cfg, 0)
    h, hsd = voc_sd("tiny")
    man, blob = pack.pack_model(cfg, sd, h, hsd, "f32")
    ctx = _lib.Context(man, blob, 0)
    try:
        r = np.random.default_rng(13)
        lens = np.array([40, 23], np.int32)
        mels = r.standard_normal((2, 40, 80)).astype(np.float32)
        e = ctx.spkemb(mels, lens)
        for b in range(2):
            check_f32(e[b], O.resnet_se34v2(mels[b, :lens[b]], sd, cfg), f"SAP embed[{b}]", 5e-5)
    finally:
        ctx.close()


def test_rccl_gather_path_in_a_torch_free_process():
    """bench.py's configuration: no torch in the process, the SYSTEM librccl + HIP runtime (the pytest process itself runs on
    torch's bundled ROCm copies).  tools/rccl_selftest.py: one-rank communicator, send/recv to self, barrier, result check."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_selftest.py")], capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0 and "torch-free RCCL self-test: OK" in p.stdout, (p.stdout[-500:], p.stderr[-1500:])
