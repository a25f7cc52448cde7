"""Per-launch log (kernel variant, shape, ms, TF/s) of one benchmark-shape step: `python tools/shape_log_b32.py [key=value ...]`
(zvx_set_int switches, e.g. dec_flat=0)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
DEC = os.environ.get("ZVX_DECODER", "styletts")
cfg = zcfg.medium_modelcfg(DEC); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config(os.environ.get("ZVX_VOCODER", "v1")); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_int(k, int(v))
ph, pu, Tl, spk, dur = synthetic.batch(32, 128, 0, "const7")
for _ in range(3): ctx.synthesize(ph, pu, Tl, spk, dur, None, want_mel=False)
ctx.set_int("profile", 2); ctx.set_int("shape_log", 1); ctx.reset_stats()
ctx.synthesize(ph, pu, Tl, spk, dur, None, want_mel=False)
