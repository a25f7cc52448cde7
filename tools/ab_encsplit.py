"""A/B of the 3-plane bf16 split GEMMs in the phoneme encoder / variance adaptor against the exact-f32 MFMA path."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("tiny"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
B, T = 32, 128
ph, pu, Tl, spk, _ = synthetic.batch(B, T, 0, None)
res = {}
for mode in (0, 1):
    ctx.set_int("enc_split", mode)
    for _ in range(2): ctx.encode(ph, pu, Tl, spk)
    ctx.set_int("profile", 2); ctx.reset_stats()
    mel_len, logd, pitch, energy = ctx.encode(ph, pu, Tl, spk)
    st = ctx.stage_times(); ks = ctx.kernel_stats(); ctx.set_int("profile", 0)
    res[mode] = dict(mel_len=mel_len, logd=logd, pitch=pitch, energy=energy, enc=ctx.fetch("encoder_out", (B, T, 528)),
                     pidx=ctx.fetch("pitch_idx", (B, T)), eidx=ctx.fetch("energy_idx", (B, T)), dur=ctx.fetch("duration", (B, T)))
    print(f"enc_split={mode}: encoder {st['encoder']:.3f} ms  variance {st['variance']:.3f} ms")
    for k in sorted(ks, key=lambda k: -k['ms']):
        print(f"   {k['name']:24s} {k['launches']:4d} launches {k['ms']:8.3f} ms {k['flops']/k['ms']/1e9:8.1f} TF/s(alg)")
a, b = res[0], res[1]
for k in ("enc", "logd", "pitch", "energy"):
    d = np.abs(a[k].astype(np.float64) - b[k]); print(f"{k:8s}: max abs diff {d.max():.3e}  rms {np.sqrt((d**2).mean()):.3e}  (ref max {np.abs(a[k]).max():.3g})")
for k in ("pidx", "eidx", "dur"):
    print(f"{k:8s}: {int((a[k] != b[k]).sum())} of {a[k].size} decisions differ")
print("mel_len equal:", np.array_equal(a["mel_len"], b["mel_len"]), a["mel_len"][:6], b["mel_len"][:6])
ctx.set_int("enc_split", 1); ctx.set_int("profile", 2); ctx.set_int("shape_log", 1); ctx.reset_stats()
ctx.encode(ph, pu, Tl, spk); ctx.stage_times()
