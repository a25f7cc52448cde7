"""Per-wave cycle breakdown of the streaming ResBlock kernels (library built with ZVX_RS_PROFILE_BUILD=1 python -m zerovox_amd.build --force)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
rng = np.random.default_rng(5)
mel = rng.standard_normal((32, 896, 80)).astype(np.float32); P = np.full(32, 896, np.int32)
ctx.set_int("rs_prof", 1)
if len(sys.argv) > 1: ctx.set_int("rs_opt", int(sys.argv[1]))
for _ in range(2): ctx.vocode_mel(mel, P)
names = ["stage 3 k3", "k7a", "k7b", "k11a", "k11b", "stage 4 k3", "k7", "k11a", "k11b"]
for rbi, t0s in ((6, [0]), (7, [0, 2]), (8, [0, 2]), (9, [0]), (10, [0]), (11, [0])):
    for t0 in t0s:
        a = ctx.fetch(f"buf:rs.prof.voc.rb{rbi}.{t0}", (512,)).view(np.int64).reshape(16, 16)
        print(f"rb{rbi} t0={t0}  [per wave: role sub | store_phase(tail), mma, epilogue, dma_wait, lgkm0, barrier] (kcycles)")
        for w in range(12):
            if a[w, :7].sum() == 0: continue
            print(f"   w{w:2d} role {a[w,14]} sub {a[w,15]} | " + " ".join(f"{x/1e3:8.1f}" for x in a[w, 1:7]) + f" | total {a[w,1:10].sum()/1e3:8.1f}" + (" | store phase: lds %.1f vmcnt0 %.1f rest %.1f" % tuple(a[w, 7:10] / 1e3) if a[w, 7:10].sum() else ""))
