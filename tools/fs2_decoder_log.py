import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
cfg = zcfg.medium_modelcfg("fastspeech2"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("tiny"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
ph, pu, T, spk, dur = synthetic.batch(32, 128, 0, "const7")
ctx.encode(ph, pu, T, spk, dur); ctx.decode(32, 896)
ctx.set_int("profile", 2); ctx.reset_stats()
ctx.encode(ph, pu, T, spk, dur); ctx.decode(32, 896)
print(ctx.stage_times())
for k in sorted(ctx.kernel_stats(), key=lambda k: -k["ms"]): print(k["name"], k["launches"], round(k["ms"], 3), round(k["flops"] / k["ms"] / 1e9, 1), "TF/s")
