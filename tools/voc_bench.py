"""Vocoder-only timing (BASELINE config #4 style): B x P frames random mel -> waveform."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
P = int(sys.argv[2]) if len(sys.argv) > 2 else 896
voc = sys.argv[3] if len(sys.argv) > 3 else "v1"
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config(voc); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
mel = np.random.default_rng(7).standard_normal((B, P, 80)).astype(np.float32)
Pn = np.full(B, P, np.int32)
for _ in range(2): ctx.vocode_mel(mel, Pn)
ctx.set_int("profile", 2); ctx.reset_stats()
n = 5
for _ in range(n): ctx.vocode_mel(mel, Pn)
st = ctx.stage_times(); ks = ctx.kernel_stats()
print(f"vocoder stage {st['vocoder']:.2f} ms  ({B*P*256/st['vocoder']/1e3:.1f} M samples/s)")
for k in sorted(ks, key=lambda k: -k['ms']):
    print(f"   {k['name']:24s} {k['launches']//n:4d} launches/step {k['ms']/n:8.3f} ms/step {k['flops']/k['ms']/1e9:8.1f} TF/s")
