"""Race screen for the barrier / counted-wait protocols of the hand-scheduled kernels: many random ragged shapes through the
streaming ResBlock kernels (bit-equal to the per-pair path), the fused attention (equal to the unfused path to bf16 rounding)
and repeated identical calls (bit-identical run to run)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = zcfg.medium_modelcfg("fastspeech2"); sd = zw.tts_state_dict(cfg, 0)
rng = np.random.default_rng(2024)
bad = 0
for voc in ("v1", "v2"):
    h = zcfg.hifigan_config(voc); hsd = zw.hifigan_state_dict(h, 0)
    man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
    ctx = _lib.Context(man, blob, 0)
    for it in range(n_iter):
        B = int(rng.integers(1, 40)); Pmax = int(rng.integers(1, 400 if B < 8 else 60))
        P = rng.integers(1, Pmax + 1, B).astype(np.int32); P[int(rng.integers(0, B))] = Pmax
        mel = np.zeros((B, Pmax, 80), np.float32)
        for b in range(B): mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
        ctx.set_int("resstream", 0); ref = ctx.vocode_mel(mel, P)
        ctx.set_int("resstream", 1); a = ctx.vocode_mel(mel, P); b2 = ctx.vocode_mel(mel, P)
        ok = np.array_equal(a, ref) and np.array_equal(a, b2) and np.isfinite(a).all()
        if not ok:
            bad += 1; print(f"MISMATCH voc={voc} it={it} B={B} Pmax={Pmax} P={P[:8]} stream-vs-pair={np.array_equal(a, ref)} rerun={np.array_equal(a, b2)}", flush=True)
    if voc == "v1":
        for it in range(n_iter):
            B = int(rng.integers(1, 12)); Lmax = int(rng.integers(2, 700 if B < 4 else 200))
            L = rng.integers(2, Lmax + 1, B).astype(np.int32); L[int(rng.integers(0, B))] = Lmax
            feats = np.zeros((B, Lmax, 528), np.float32); spk = rng.standard_normal((B, 528)).astype(np.float32); spk /= np.linalg.norm(spk, axis=1, keepdims=True)
            for b in range(B): feats[b, :L[b]] = rng.standard_normal((L[b], 528)).astype(np.float32)
            ctx.set_int("flash", 0); u = ctx.decode_features(feats, L, spk)
            ctx.set_int("flash", 1); f1 = ctx.decode_features(feats, L, spk); f2 = ctx.decode_features(feats, L, spk)
            d = max(np.abs(f1[b, :L[b]] - u[b, :L[b]]).max() / max(1.0, np.abs(u[b, :L[b]]).max()) for b in range(B))
            ok = np.array_equal(f1, f2) and np.isfinite(f1).all() and d < 0.05
            if not ok:
                bad += 1; print(f"MISMATCH attention it={it} B={B} Lmax={Lmax} L={L[:8]} rerun={np.array_equal(f1, f2)} maxdiff={d:.3e}", flush=True)
    ctx.close()
print("stress:", "OK" if bad == 0 else f"{bad} FAILURES")
