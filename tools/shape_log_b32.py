import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
ph, pu, Tl, spk, dur = synthetic.batch(32, 128, 0, "const7")
for _ in range(3): ctx.synthesize(ph, pu, Tl, spk, dur, None, want_mel=False)
ctx.set_int("profile", 2); ctx.set_int("shape_log", 1); ctx.reset_stats()
ctx.synthesize(ph, pu, Tl, spk, dur, None, want_mel=False)
