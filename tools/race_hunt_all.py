"""Race screen over the other paths: every result of N repeated calls compared bit for bit with the first -- FS2 / SCLN decoder
(fused bf16 attention), HiFi-GAN V2 / V3, f32 mode, the speaker encoder (ASP and SAP), ragged batches, batch 1.
   python tools/race_hunt_all.py [N=60]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad_total = 0
def screen(name, fn, n=N):
    global bad_total
    ref = fn(); bad = 0
    for i in range(n):
        out = fn()
        if not all(np.array_equal(a, b) for a, b in zip(out, ref)): bad += 1
    bad_total += bad
    print(f"{name}: {bad} / {n} mismatches", flush=True)
for kind, voc, prec in (("fastspeech2", "v1", "bf16"), ("styletts", "v2", "bf16"), ("styletts", "v3", "bf16"), ("styletts", "v1", "f32"), ("fastspeech2", "v2", "f32")):
    cfg = zcfg.medium_modelcfg(kind); sd = zw.tts_state_dict(cfg, 0)
    h = zcfg.hifigan_config(voc); hsd = zw.hifigan_state_dict(h, 0)
    man, blob = pack.pack_model(cfg, sd, h, hsd, prec)
    ctx = _lib.Context(man, blob, 0)
    rng = np.random.default_rng(1)
    for (B, T) in (((32, 128), (5, 70), (1, 64)) if prec == "bf16" else ((6, 100), (1, 64))):
        ph, pu, Tl, spk, dur = synthetic.batch(B, T, 40, "const7")
        Tl = rng.integers(max(1, T // 2), T + 1, B).astype(np.int32); Tl[0] = T
        for b in range(B): ph[b, Tl[b]:] = 0; pu[b, Tl[b]:] = 0; dur[b, Tl[b]:] = 0
        pad = (dur.sum(1)).astype(np.int32)
        def f():
            r = ctx.synthesize(ph, pu, Tl, spk, dur, pad, want_mel=True)
            return r["mel"], r["wav"]
        screen(f"{kind}/{voc}/{prec} B={B} T={T} ragged", f, N if B > 1 else 2 * N)
    if voc == "v1" and prec == "bf16":
        for (B, L) in ((250, 258), (7, 301), (1, 96)):
            mels = rng.standard_normal((B, L, 80)).astype(np.float32)
            lens = rng.integers(max(20, L // 2), L + 1, B).astype(np.int32); lens[0] = L
            screen(f"speaker encoder B={B} L={L}", lambda: (ctx.spkemb(mels, lens),), N)
    ctx.close()
print("ALL REPRODUCIBLE" if bad_total == 0 else f"{bad_total} MISMATCHES")
