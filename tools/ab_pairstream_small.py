import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
for B, P in ((1, 1024), (1, 448), (2, 896), (4, 896), (8, 896), (16, 896)):
    mel = np.random.default_rng(7).standard_normal((B, P, 80)).astype(np.float32)
    mel_d = ctx.dev_alloc(mel.nbytes); ctx.dev_from_host(mel_d, mel); wav_d = ctx.dev_alloc(B * P * 256 * 4)
    Pn = np.full(B, P, np.int32)
    for mode in (-1, 1):
        ctx.set_int("pairstream", mode)
        for _ in range(5): ctx.vocode_mel_device(mel_d, Pn, P, wav_d, P * 256, no_sync=True)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(30): ctx.vocode_mel_device(mel_d, Pn, P, wav_d, P * 256, no_sync=True)
        ctx.sync(); print(f"B={B} P={P} pairstream={mode}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms", flush=True)
