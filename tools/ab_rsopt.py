"""A/B of StreamArgs.opt (zvx_set_int("rs_opt", v)) of the streaming ResBlock kernels: bit-equality and per-launch timing of the
narrow stages at the benchmark shape, alternating the settings so that box drift cancels.
   python tools/ab_rsopt.py [v ...]      (default: 0 1)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib
opts = [int(v) for v in sys.argv[1:]] or [0, 1]
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
rng = np.random.default_rng(5)
ok = True
for (B, Pmax) in ((3, 23), (5, 70), (2, 300), (32, 40)):
    P = rng.integers(1, Pmax + 1, B).astype(np.int32); P[0] = Pmax
    mel = np.zeros((B, Pmax, 80), np.float32)
    for b in range(B): mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
    ctx.set_int("rs_opt", opts[0]); w0 = ctx.vocode_mel(mel, P)
    for o in opts[1:]:
        ctx.set_int("rs_opt", o); w1 = ctx.vocode_mel(mel, P)
        same = np.array_equal(w0, w1); ok = ok and same
        print(f"B={B} Pmax={Pmax} rs_opt {opts[0]} vs {o}: bit-equal={same} maxdiff={np.abs(w0 - w1).max():.3e}", flush=True)
print("ALL BIT-EQUAL" if ok else "MISMATCH")
B, Pn = 32, 896
mel = rng.standard_normal((B, Pn, 80)).astype(np.float32); P = np.full(B, Pn, np.int32)
acc = {o: {} for o in opts}
voc = {o: [] for o in opts}
for rep in range(3):
    for o in opts:
        ctx.set_int("rs_opt", o)
        for _ in range(2): ctx.vocode_mel(mel, P)
        ctx.set_int("profile", 2); ctx.set_int("shape_log", 1 if rep == 2 else 0); ctx.reset_stats()
        n = 3
        for _ in range(n): ctx.vocode_mel(mel, P)
        st = ctx.stage_times(); ks = ctx.kernel_stats(); ctx.set_int("profile", 0); ctx.set_int("shape_log", 0)
        voc[o].append(st["vocoder"])
        for k in ks:
            if k["launches"] and "resstream" in k["name"]:
                acc[o].setdefault(k["name"], []).append((k["ms"] / n, k["flops"] / k["ms"] / 1e9))
for o in opts:
    print(f"rs_opt={o}: vocoder " + " ".join(f"{v:.2f}" for v in voc[o]) + " ms")
    for name, v in acc[o].items():
        print(f"   {name:24s} " + " ".join(f"{ms:7.3f} ms ({tf:6.1f} TF/s)" for ms, tf in v))
