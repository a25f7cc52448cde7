"""Per-launch log of one speaker-encoder batch (dev aid): ZVX_SHAPE_LOG=1 python tools/spk_log.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("tiny"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
mels = np.random.default_rng(8).standard_normal((50, 258, 80)).astype(np.float32); lens = np.full(50, 258, np.int32)
ctx.spkemb(mels, lens)
ctx.set_int("profile", 2); ctx.reset_stats()
ctx.spkemb(mels, lens)
print(ctx.stage_times())
for k in sorted(ctx.kernel_stats(), key=lambda k: -k["ms"]): print(k["name"], k["launches"], round(k["ms"], 3), round(k["flops"] / k["ms"] / 1e9, 1), "TF/s")
