"""Development aid (round 5): reduced-width models of several hidden sizes in the 16-bit mode against the NumPy oracle, shortcut fusion on / off."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import zvx_oracle as O
from zerovox_amd import _lib, config as zcfg, pack, weights as zw
h = zcfg.hifigan_config("tiny3"); hsd = zw.hifigan_state_dict(h, 3)
for H in [int(a) for a in sys.argv[1:]] or [32, 48, 64, 96, 128]:
    cfg = zcfg.reduced_modelcfg("styletts"); cfg["model"]["emb_dim"] = H - 16
    sd = zw.tts_state_dict(cfg, 3)
    r = np.random.default_rng(5); T = 20
    ph = r.integers(1, 29, size=T).astype(np.int32); pu = r.integers(1, 11, size=T).astype(np.int32)
    spk = r.standard_normal(H); spk = (spk / np.linalg.norm(spk)).astype(np.float32)
    dur = r.integers(2, 7, size=T).astype(np.int32)
    ref = O.inference_ex(sd, hsd, cfg, h, ph, pu, spk, duration=dur, pad_to=689)
    man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
    ctx = _lib.Context(man, blob, 0)
    for fuse in (1, 0, 2 + 1, 4 + 1, 8 + 1, 16 + 1):
        ctx.set_int("dec_sc_fuse", fuse)
        out = ctx.synthesize(ph[None], pu[None], np.array([T], np.int32), spk[None], dur[None], np.array([689], np.int32))
        ml = int(out["mel_len"][0])
        em = np.abs(out["mel"][0, :ml] - ref["mel"].T if ref["mel"].shape[0] == 80 else out["mel"][0, :ml] - ref["mel"])
        print(f"H={H} fuse={fuse}: mel_len {ml}/{ref['mel_len']} mel max err {em.max():.3e} rms {np.sqrt((em**2).mean()):.3e}", flush=True)
    ctx.close()
