"""A/B of the multi-chunk slab fill of the conv-slab kernel (G K-chunks staged per global round trip; plain GEMMs / 1 x 1
convolutions over a deep K): bit-equality of the whole step and stage times, batch 32 and batch 1.   python tools/ab_slab_multi.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
for kind in ("styletts", "fastspeech2"):
    cfg = zcfg.medium_modelcfg(kind); sd = zw.tts_state_dict(cfg, 0)
    h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
    man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
    ctx = _lib.Context(man, blob, 0)
    for (B, T) in ((32, 128), (1, 64), (3, 37)):
        ph, pu, Tl, spk, dur = synthetic.batch(B, T, 0, "const7")
        pad = np.full(B, 7 * T, np.int32)
        res = {}
        for mode in (0, 1, 0, 1):
            ctx.set_int("slab_multi", mode)
            for _ in range(2): r = ctx.synthesize(ph, pu, Tl, spk, dur, pad, want_mel=True)
            ctx.set_int("profile", 1); ts = []
            for _ in range(5):
                r = ctx.synthesize(ph, pu, Tl, spk, dur, pad, want_mel=True); ts.append(ctx.stage_times())
            ctx.set_int("profile", 0)
            st = {k: float(np.mean([t[k] for t in ts])) for k in ("encoder", "variance", "decoder", "vocoder")}
            if mode in res:
                same = np.array_equal(res[mode]["mel"], r["mel"]) and np.array_equal(res[mode]["wav"], r["wav"])
            res[mode] = r
            print(f"{kind} B={B} T={T} slab_multi={mode}: " + " ".join(f"{k} {v:.3f}" for k, v in st.items()) + f"  sum {sum(st.values()):.3f} ms", flush=True)
        print(f"   multi == single: mel {np.array_equal(res[0]['mel'], res[1]['mel'])} wav {np.array_equal(res[0]['wav'], res[1]['wav'])}", flush=True)
