"""A/B of the phoneme encoder's arithmetic in the 16-bit mode at the benchmark shape (32 x 128 phonemes, predicted durations):
enc_split 0 = exact-f32 MFMA, 1 = 3-plane split products on bf16 planes (rounds 2-3), 2 = on IEEE-half planes (default).
Prints stage times, float differences and the number of discrete decisions (pitch / energy buckets, durations) that differ
between the modes -- and, with --oracle N, against the NumPy oracle for the first N utterances (the f32 noise floor: two exact
f32 implementations with different summation orders also move a few decisions that sit on a rounding boundary)."""
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
for mode in (0, 1, 2):
    ctx.set_int("enc_split", mode)
    for _ in range(2): ctx.encode(ph, pu, Tl, spk)
    ctx.set_int("profile", 2); ctx.reset_stats()
    mel_len, logd, pitch, energy = ctx.encode(ph, pu, Tl, spk)
    st = ctx.stage_times(); ks = ctx.kernel_stats(); ctx.set_int("profile", 0)
    res[mode] = dict(mel_len=mel_len, logd=logd, pitch=pitch, energy=energy, enc=ctx.fetch("encoder_out", (B, T, 528)),
                     pidx=ctx.fetch("pitch_idx", (B, T)), eidx=ctx.fetch("energy_idx", (B, T)), dur=ctx.fetch("duration", (B, T)))
    print(f"enc_split={mode}: encoder {st['encoder']:.3f} ms  variance {st['variance']:.3f} ms")
    for k in sorted(ks, key=lambda k: -k['ms'])[:4]:
        print(f"   {k['name']:24s} {k['launches']:4d} launches {k['ms']:8.3f} ms {k['flops']/k['ms']/1e9:8.1f} TF/s(alg)")
a = res[0]
for m in (1, 2):
    b = res[m]
    print(f"--- enc_split {m} against the exact-f32 MFMA path")
    for k in ("enc", "logd", "pitch", "energy"):
        d = np.abs(a[k].astype(np.float64) - b[k]); print(f"{k:8s}: max abs diff {d.max():.3e}  rms {np.sqrt((d**2).mean()):.3e}  (ref max {np.abs(a[k]).max():.3g})")
    for k in ("pidx", "eidx", "dur"):
        print(f"{k:8s}: {int((a[k] != b[k]).sum())} of {a[k].size} decisions differ")
    print("mel_len equal:", np.array_equal(a["mel_len"], b["mel_len"]))
if "--oracle" in sys.argv:
    n = int(sys.argv[sys.argv.index("--oracle") + 1])
    from oracle import zvx_oracle as O
    cnt = {m: [0, 0, 0] for m in res}
    err = {m: 0.0 for m in res}
    for u in range(n):
        ref = O.fs2_encoder(ph[u], pu[u], spk[u], sd, cfg)
        for m, r in res.items():
            cnt[m][0] += int((r["pidx"][u] != ref["pitch_idx"]).sum()); cnt[m][1] += int((r["eidx"][u] != ref["energy_idx"]).sum())
            cnt[m][2] += int((r["dur"][u] != ref["duration"]).sum())
            err[m] = max(err[m], float(np.abs(r["pitch"][u] - ref["pitch"]).max()))
    for m in res:
        print(f"enc_split {m} against the NumPy oracle, {n} utterances ({n * T} phonemes): pitch / energy / duration decisions that differ: {cnt[m]}, max |pitch err| {err[m]:.2e}")
