"""A/B of the fused attention kernel in the FS2 / SCLN decoder against score GEMM + softmax + PV GEMM."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib
from oracle import zvx_oracle as O
cfg = zcfg.medium_modelcfg("fastspeech2"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("tiny"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
rng = np.random.default_rng(3)
# small ragged case against the oracle
L = np.array([50, 7, 130, 33], np.int32)
feats = np.zeros((4, 130, 528), np.float32); spk = rng.standard_normal((4, 528)).astype(np.float32); spk /= np.linalg.norm(spk, axis=1, keepdims=True)
for b in range(4): feats[b, :L[b]] = rng.standard_normal((L[b], 528)).astype(np.float32)
outs = {}
for mode in (0, 1):
    ctx.set_int("flash", mode)
    outs[mode] = ctx.decode_features(feats, L, spk)
for b in range(4):
    ref = O.fs2_decoder(feats[b, :L[b]], spk[b], sd, cfg)
    e0 = np.abs(outs[0][b, :L[b]] - ref); e1 = np.abs(outs[1][b, :L[b]] - ref); d = np.abs(outs[0][b, :L[b]] - outs[1][b, :L[b]])
    print(f"utt {b} L={L[b]:3d}: unfused vs oracle max {e0.max():.3e} rms {np.sqrt((e0**2).mean()):.3e} | flash vs oracle max {e1.max():.3e} rms {np.sqrt((e1**2).mean()):.3e} | flash vs unfused max {d.max():.3e}  (ref rms {np.sqrt((ref**2).mean()):.3f})")
# bench shape
B, Lb = 32, 896
feats = rng.standard_normal((B, Lb, 528)).astype(np.float32); spk = rng.standard_normal((B, 528)).astype(np.float32); Ln = np.full(B, Lb, np.int32)
for mode in (0, 1):
    ctx.set_int("flash", mode)
    for _ in range(2): ctx.decode_features(feats, Ln, spk)
    ctx.set_int("profile", 2); ctx.reset_stats()
    ctx.decode_features(feats, Ln, spk)
    st = ctx.stage_times(); ts = ctx.tag_stats(); ks = ctx.kernel_stats(); ctx.set_int("profile", 0)
    print(f"flash={mode}: decoder {st['decoder']:.3f} ms; " + "; ".join(f"{k['name']} {k['launches']}x {k['ms']:.3f}" for k in sorted(ks, key=lambda k: -k['ms'])) + f"; tagged total {sum(t['ms'] for t in ts):.3f}")
