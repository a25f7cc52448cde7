"""Vocoder time against the ResBlock sub-batch size (Infinity-Cache residency of the stage tensors)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
rng = np.random.default_rng(5)
mel = rng.standard_normal((32, 896, 80)).astype(np.float32); P = np.full(32, 896, np.int32)
ref = None
for ch in (-1, 16, 8, 4, 2, 1, 0):
    ctx.set_int("voc_chunk", ch)
    for _ in range(2): w = ctx.vocode_mel(mel, P)
    if ref is None: ref = w
    ctx.set_int("profile", 2); ctx.reset_stats()
    n = 3
    for _ in range(n): ctx.vocode_mel(mel, P)
    st = ctx.stage_times(); ts = ctx.tag_stats(); ctx.set_int("profile", 0)
    print(f"voc_chunk={ch:3d}: vocoder {st['vocoder']:.2f} ms  bit-equal={np.array_equal(w, ref)}  " + "  ".join(f"{t['name']}={t['ms']/n:.2f}" for t in sorted(ts, key=lambda t: t['name']) if t['name'].startswith('voc.res')))
