"""A/B of the encoder's fused exact-f32 attention (attention.hip: attn_f32_kernel, one Q|K|V GEMM + one launch) against the
unfused path (V^T / score / P.V GEMMs + softmax): difference of the encoder output on ragged batches (incl. T > 128 and
T > 512), batch invariance, and the encoder stage time at the benchmark shape.   python tools/ab_attn_f32.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
ok = True
for prec in ("bf16", "f32"):
    man, blob = pack.pack_model(cfg, sd, h, hsd, prec)
    ctx = _lib.Context(man, blob, 0)
    for split in ((1, 0) if prec == "bf16" else (0,)):
        ctx.set_int("enc_split", split)
        rng = np.random.default_rng(3)
        for (B, Tmax) in ((1, 1), (1, 7), (3, 33), (4, 128), (3, 129), (2, 300), (1, 530)):
            ph, pu, T, spk, dur = synthetic.batch(B, Tmax, first_utt=11, dur_mode="const7")
            T = rng.integers(1, Tmax + 1, B).astype(np.int32); T[0] = Tmax
            for b in range(B): ph[b, T[b]:] = 0; pu[b, T[b]:] = 0; dur[b, T[b]:] = 0
            outs = {}
            for mode in (0, 1):
                ctx.set_int("attn_f32", mode)
                ml, logd, pitch, energy = ctx.encode(ph, pu, T, spk, dur)
                outs[mode] = (ctx.fetch("encoder_out", (B, Tmax, 528)).copy(), logd.copy(), pitch.copy(), energy.copy())
            e = [float(np.abs(outs[0][i] - outs[1][i]).max()) for i in range(4)]
            scale = float(np.abs(outs[0][0]).max())
            # batch invariance of the fused path: utterance 0 alone == utterance 0 in the batch
            ml1, logd1, _, _ = ctx.encode(ph[:1], pu[:1], T[:1], spk[:1], dur[:1])
            alone = ctx.fetch("encoder_out", (1, Tmax, 528))[0]
            inv = np.array_equal(alone[:T[0]], outs[1][0][0][:T[0]])
            fin = bool(np.isfinite(outs[1][0]).all())
            good = e[0] <= 2e-4 * max(1.0, scale) and inv and fin
            ok = ok and good
            print(f"{prec} split={split} B={B} Tmax={Tmax} T={T.tolist()}: max|fused - unfused| enc_out {e[0]:.2e} (scale {scale:.2f}) logd {e[1]:.2e} pitch {e[2]:.2e} energy {e[3]:.2e}  batch-invariant={inv} finite={fin} {'ok' if good else 'BAD'}", flush=True)
    if prec == "bf16":
        ctx.set_int("enc_split", 1)
        ph, pu, T, spk, dur = synthetic.batch(32, 128, first_utt=0, dur_mode="const7")
        for mode in (0, 1, 0, 1):
            ctx.set_int("attn_f32", mode)
            for _ in range(3): ctx.encode(ph, pu, T, spk, dur)
            ctx.set_int("profile", 1); ts = []
            for _ in range(5):
                ctx.encode(ph, pu, T, spk, dur); ts.append(ctx.stage_times()["encoder"])
            ctx.set_int("profile", 0)
            print(f"attn_f32={mode}: encoder stage {np.mean(ts):.3f} ms (B = 32 x 128 phonemes, bf16 mode, split products)")
        for (B1, T1) in ((1, 64), (1, 128)):
            ph, pu, T, spk, dur = synthetic.batch(B1, T1, first_utt=0, dur_mode="const7")
            for mode in (0, 1):
                ctx.set_int("attn_f32", mode)
                for _ in range(3): ctx.encode(ph, pu, T, spk, dur)
                ctx.set_int("profile", 1); ts = []
                for _ in range(5):
                    ctx.encode(ph, pu, T, spk, dur); ts.append(ctx.stage_times()["encoder"])
                ctx.set_int("profile", 0)
                print(f"attn_f32={mode}: encoder stage {np.mean(ts):.3f} ms (B = {B1} x {T1})")
print("ALL OK" if ok else "FAILURES")
