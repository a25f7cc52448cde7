"""StyleTTS decoder in IEEE half (default in the 16-bit mode) against bf16 (zvx_set_int("dec_f16", 0)) and the f32 oracle: mel and
waveform error on the golden utterance and on ragged batches, decoder stage time at the benchmark shape.   python tools/ab_dec_f16.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
from oracle import zvx_oracle as O
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "e2e_styletts_v1_T64.npz"))
def stats(a, b):
    d = np.abs(np.asarray(a, np.float64) - b); return d.max(), np.sqrt((d ** 2).mean())
feat, spk, L = g["features"], g["spk"], g["features"].shape[0]
for mode in (0, 1):
    ctx.set_int("dec_f16", mode)
    mel = ctx.decode_features(feat[None], np.array([L], np.int32), spk[None])[0]
    wav = ctx.vocode_mel(mel[None], np.array([L], np.int32))[0]
    mx, rms = stats(mel[:L], g["mel"].T); wx, wr = stats(wav[:len(g["wav"])], g["wav"])
    print(f"dec_f16={mode}: golden T=64: mel max {mx:.3e} rms {rms:.3e} | wav (decoder + vocoder) max {wx:.3e} rms {wr:.3e}  finite={np.isfinite(mel).all()}", flush=True)
ph, pu, Tl, spk32, dur = synthetic.batch(32, 128, 0, "const7"); pad = np.full(32, 896, np.int32)
res = {}
for mode in (0, 1, 0, 1):
    ctx.set_int("dec_f16", mode)
    for _ in range(2): r = ctx.synthesize(ph, pu, Tl, spk32, dur, pad, want_mel=True)
    ctx.set_int("profile", 1); ts = []
    for _ in range(5): r = ctx.synthesize(ph, pu, Tl, spk32, dur, pad, want_mel=True); ts.append(ctx.stage_times()["decoder"])
    ctx.set_int("profile", 0); res[mode] = r
    print(f"dec_f16={mode}: decoder stage {np.mean(ts):.3f} ms (B = 32 x 896 frames)  mel finite={np.isfinite(r['mel']).all()} max|mel| {np.abs(r['mel']).max():.2f}")
d = np.abs(res[0]["mel"] - res[1]["mel"]); print(f"bf16 vs f16 decoder on the benchmark batch: mel max diff {d.max():.3e} rms {np.sqrt((d**2).mean()):.3e}")
# one utterance of the batch against the oracle (f32)
b = 3
ref = O.inference_ex(sd, hsd, cfg, h, ph[b], pu[b], spk32[b], duration=dur[b], pad_to=896)
for mode in (0, 1):
    mx, rms = stats(res[mode]["mel"][b][:ref["mel_len"]], np.asarray(ref["mel"]).T if np.asarray(ref["mel"]).shape[0] == 80 else ref["mel"]); wx, wr = stats(res[mode]["wav"][b][:len(ref["wav"])], ref["wav"])
    print(f"dec_f16={mode}: utterance {b} of the batch vs the f32 oracle: mel max {mx:.3e} rms {rms:.3e} | wav max {wx:.3e} rms {wr:.3e}")
