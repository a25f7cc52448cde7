"""Per-launch shape log of one bench step (dev aid): python tools/shape_log.py [decoder] [vocoder] -> stderr lines from libzvx."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
kind = sys.argv[1] if len(sys.argv) > 1 else "styletts"
voc = sys.argv[2] if len(sys.argv) > 2 else "v1"
cfg = zcfg.medium_modelcfg(kind); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config(voc); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
ph, pu, T, spk, dur = synthetic.batch(32, 128, 0, "const7")
pad = np.full(32, 896, np.int32)
for _ in range(2): ctx.synthesize(ph, pu, T, spk, dur, pad, want_mel=False)
ctx.set_int("profile", 2); ctx.set_int("shape_log", 1); ctx.reset_stats()
ctx.synthesize(ph, pu, T, spk, dur, pad, want_mel=False)
print({k: round(v, 3) for k, v in ctx.stage_times().items()})
