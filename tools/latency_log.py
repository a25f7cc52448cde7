"""Per-launch timeline of ONE short utterance (default 64 phonemes -> 448 frames), phoneme -> waveform: python tools/latency_log.py [T]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
for kv in sys.argv[2:]:
    k, v = kv.split("="); ctx.set_int(k, int(v))
ph, pu, Tl, spk, dur = synthetic.batch(1, T, 0, "const7")
pad = np.full(1, 7 * T, np.int32)
for _ in range(3): ctx.synthesize(ph, pu, Tl, spk, dur, pad, want_mel=False)
import time
t0 = time.time()
for _ in range(20): ctx.synthesize(ph, pu, Tl, spk, dur, pad, want_mel=False)
print(f"host wall per call (profile off): {(time.time() - t0) / 20 * 1e3:.3f} ms")
ctx.set_int("profile", 2); ctx.set_int("shape_log", 1); ctx.reset_stats()
ctx.synthesize(ph, pu, Tl, spk, dur, pad, want_mel=False)
print({k: round(v, 3) for k, v in ctx.stage_times().items()})
