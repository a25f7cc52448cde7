"""Development aid (round 5): which ResBlock2 shape breaks in IEEE half?  Custom generator configs, f16 vs bf16 vs oracle."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import zvx_oracle as O
from zerovox_amd import _lib, config as zcfg, pack, weights as zw

cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
base = zcfg.hifigan_config("v3")
cases = {"v3": {}, "k3": {"resblock_kernel_sizes": [3], "resblock_dilation_sizes": [[1, 2]]}, "k5": {"resblock_kernel_sizes": [5], "resblock_dilation_sizes": [[2, 6]]},
         "k7d3": {"resblock_kernel_sizes": [7], "resblock_dilation_sizes": [[3, 3]]}, "k7d12": {"resblock_kernel_sizes": [7], "resblock_dilation_sizes": [[3, 12]]},
         "c64": {"upsample_initial_channel": 128}}
P = np.array([19, 7], np.int32)
rng = np.random.default_rng(13)
mel = np.zeros((2, 19, 80), np.float32)
for b in range(2): mel[b, :P[b]] = rng.standard_normal((P[b], 80)).astype(np.float32)
for name, ov in cases.items():
    h = dict(base); h.update(ov)
    hsd = zw.hifigan_state_dict(h, 0)
    man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
    ctx = _lib.Context(man, blob, 0)
    ref = O.hifigan_generator(mel[0, :P[0]].T, hsd, h)
    for f16 in (0, 1):
        ctx.set_int("voc_f16", f16)
        w = ctx.vocode_mel(mel, P)
        e = np.abs(w[0, :P[0] * 256] - ref)
        print(f"{name:6s} voc_f16={f16}: max err {e.max():.3e} rms {np.sqrt((e**2).mean()):.3e} finite={np.isfinite(w).all()}", flush=True)
    ctx.close()
