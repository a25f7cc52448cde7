"""Duration predictor beside the pitch predictor (va_overlap_maxb) against the serial schedule by batch size: variance stage time (events) per call.
   python tools/ab_norm_fuse.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
for T in (64, 128):
    for B in (1, 2, 4, 8, 16, 32):
        ph, pu, Tl, spk, dur = synthetic.batch(B, T, 0, "const7")
        row = []
        ref = None
        for maxb in (0, 1 << 20):
            ctx.set_int("va_overlap_maxb", maxb)
            for _ in range(3): r = ctx.synthesize(ph, pu, Tl, spk, dur, None, want_mel=True)
            ctx.set_int("profile", 1); ts = []
            for _ in range(12): ctx.synthesize(ph, pu, Tl, spk, dur, None, want_mel=False); ts.append(ctx.stage_times()["variance"])
            ctx.set_int("profile", 0)
            if ref is None: ref = r
            row.append((np.median(ts), np.array_equal(r["mel"], ref["mel"])))
        print(f"T={T:4d} B={B:3d}: variance adaptor serial {row[0][0]:.3f} ms, side by side {row[1][0]:.3f} ms, mel bit-equal {row[1][1]}", flush=True)
