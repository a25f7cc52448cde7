"""Per-launch timeline of the encoder + variance adaptor for B utterances of T phonemes (dev aid): python tools/enc_log.py B T"""
import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
B, T = int(sys.argv[1]), int(sys.argv[2])
ph, pu, Tl, spk, dur = synthetic.batch(B, T, first_utt=0, dur_mode="const7")
for _ in range(3): ctx.encode(ph, pu, Tl, spk, dur)
ctx.set_int("profile", 2); ctx.set_int("shape_log", 1); ctx.reset_stats()
ctx.encode(ph, pu, Tl, spk, dur); st = ctx.stage_times(); print(st)
