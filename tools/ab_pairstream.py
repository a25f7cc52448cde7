"""A/B of the streaming pair kernel (pairstream.hip, C = 128) against the two conv-slab launches it replaces:
bit-equality on ragged batches + per-kernel timing of the vocoder at the benchmark shape.
   python tools/ab_pairstream.py [v1] [quick]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib
voc = sys.argv[1] if len(sys.argv) > 1 else "v1"
quick = "quick" in sys.argv
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config(voc); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
rng = np.random.default_rng(5)
ok = True
for (B, Pmax) in ((3, 23), (5, 70), (2, 300), (1, 9), (1, 1), (32, 40), (1, 1100), (40, 33)):
    P = rng.integers(1, Pmax + 1, B).astype(np.int32); P[0] = Pmax
    mel = np.zeros((B, Pmax, 80), np.float32)
    for b in range(B): mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
    ctx.set_int("pairstream", -1); w0 = ctx.vocode_mel(mel, P)
    for mode in (3, 4):
        ctx.set_int("pairstream", mode); w1 = ctx.vocode_mel(mel, P)
        same = np.array_equal(w0, w1)
        d = np.abs(w0 - w1)
        print(f"B={B} Pmax={Pmax} P={P[:6]} mode={mode}: bit-equal={same} maxdiff={d.max():.3e} nonfinite={np.sum(~np.isfinite(w1))}", flush=True)
        if not same:
            ok = False
            bad = np.argwhere(d > 0)
            print("   first diffs (utt, sample):", bad[:5].tolist(), " samples with diffs per utt:", [int((d[b] > 0).sum()) for b in range(B)][:8])
    # determinism of the new path
    w2 = ctx.vocode_mel(mel, P)
    if not np.array_equal(w1, w2): ok = False; print("   NON-DETERMINISTIC")
print("ALL BIT-EQUAL" if ok else "MISMATCH", flush=True)
B, Pn = 32, 896
mel = rng.standard_normal((B, Pn, 80)).astype(np.float32); P = np.full(B, Pn, np.int32)
for mode in ((-1, 1, 3) if not quick else (-1, 1)):
    ctx.set_int("pairstream", mode)
    for _ in range(3): ctx.vocode_mel(mel, P)
    ctx.set_int("profile", 2); ctx.reset_stats()
    n = 3
    for _ in range(n): ctx.vocode_mel(mel, P)
    st = ctx.stage_times(); ks = ctx.kernel_stats(); ctx.set_int("profile", 0)
    print(f"pairstream={mode}: vocoder {st['vocoder']:.2f} ms")
    for k in sorted(ks, key=lambda k: -k['ms']):
        if k['launches']:
            print(f"   {k['name']:24s} {k['launches']//n:4d} launches {k['ms']/n:8.3f} ms {k['flops']/k['ms']/1e9:8.1f} TF/s {k['bytes']/k['ms']/1e6:8.1f} GB/s(alg)")
for mode in (-1, 1):
    print(f"---- shape log, pairstream={mode}")
    ctx.set_int("pairstream", mode); ctx.set_int("profile", 2); ctx.set_int("shape_log", 1); ctx.reset_stats()
    ctx.vocode_mel(mel, P); ctx.stage_times(); ctx.set_int("shape_log", 0)
