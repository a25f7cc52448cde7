"""ResBlocks of a vocoder stage side by side (voc_overlap_maxb) by batch size and length: vocoder stage time per call.
   python tools/ab_voc_overlap.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
rng = np.random.default_rng(2)
for P in (224, 448, 896):
    for B in (1, 2, 4, 6, 8, 12, 16, 32):
        mel = rng.standard_normal((B, P, 80)).astype(np.float32); L = np.full(B, P, np.int32)
        row = []; ref = None
        for maxb in (0, 1 << 20):
            ctx.set_int("voc_overlap_maxb", maxb)
            for _ in range(3): w = ctx.vocode_mel(mel, L)
            ctx.set_int("profile", 1); ts = []
            for _ in range(10): w = ctx.vocode_mel(mel, L); ts.append(ctx.stage_times()["vocoder"])
            ctx.set_int("profile", 0)
            if ref is None: ref = w
            row.append((np.median(ts), np.array_equal(w, ref)))
        print(f"P={P:4d} B={B:3d}: serial {row[0][0]:.3f} ms, side by side {row[1][0]:.3f} ms ({100 * (row[1][0] / row[0][0] - 1):+.1f} %), bit-equal {row[1][1]}", flush=True)
