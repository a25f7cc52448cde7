"""Race screen for the streaming pair kernel (pairstream.hip) at every job size: random ragged batches, the pair kernel forced
(modes 3 and 4: 1024- / 256-row segment floors) against the two conv-slab launches per pair (mode -1), each variant run 4 times.
Any mismatch is printed with the first differing (utterance, sample).     python tools/stress_pairstream.py [iterations=40]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
rng = np.random.default_rng(77)
bad = 0
for it in range(n_iter):
    B = int(rng.integers(1, 48)); Pmax = int(rng.integers(1, 900 if B < 6 else (200 if B < 20 else 64)))
    P = rng.integers(1, Pmax + 1, B).astype(np.int32); P[int(rng.integers(0, B))] = Pmax
    mel = np.zeros((B, Pmax, 80), np.float32)
    for b in range(B): mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
    ctx.set_int("pairstream", -1); ref = ctx.vocode_mel(mel, P)
    for mode in (3, 4, 1):
        ctx.set_int("pairstream", mode)
        for rep in range(4):
            w = ctx.vocode_mel(mel, P)
            if not np.array_equal(w, ref):
                bad += 1
                d = np.abs(w - ref); u = np.argwhere(d.max(1) > 0)[:, 0]
                print(f"MISMATCH it={it} B={B} Pmax={Pmax} mode={mode} rep={rep}: utterances {u[:6].tolist()} first samples {[int(np.argmax(d[b] > 0)) for b in u[:4]]} max {d.max():.3e}", flush=True)
ctx.set_int("pairstream", 1)
print(f"{n_iter} shapes x 3 modes x 4 runs: {bad} mismatches")
