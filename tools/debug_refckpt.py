"""Development aid (round 5): the reduced-width reference checkpoint in the 16-bit mode -- which decoder switch makes the mel wrong?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zerovox_amd.synthesize import ZeroVoxTTS
G = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(G, "refckpt_expected.npz"))
x = {"phoneme": g["phoneme"][None], "puncts": g["puncts"][None], "duration": g["duration"][None]}
for prec in ("f32", "bf16"):
    _, synth = ZeroVoxTTS.load_model(os.path.join(G, "refckpt"), "synthetic:tiny3:5", infer_device="cuda:0", precision=prec)
    ctx = synth._model.ctx
    combos = [{}] if prec == "f32" else [{}, {"dec_f16": 0}, {"dec_flat": 0}, {"dec_sc_fuse": 0}, {"norm_fuse_maxb": 0}, {"slab_small": 0}, {"dec_f16": 0, "dec_flat": 0, "dec_sc_fuse": 0, "norm_fuse_maxb": 0}, {"voc_f16": 0}]
    for sw in combos:
        for k, v in sw.items(): ctx.set_int(k, v)
        synth._model._min_mel_len = 689
        wav, ml, logd, mel = synth._model.inference_ex(x, g["spk"][None, None], force_duration=True)
        em = np.abs(mel - g["forced_mel"]); ew = np.abs(wav - g["forced_wav"])
        print(f"{prec} {sw}: mel max {em.max():.3e} rms {np.sqrt((em**2).mean()):.3e} | wav max {ew.max():.3e}  finite={np.isfinite(mel).all()}", flush=True)
        for k, v in sw.items(): ctx.set_int(k, {"norm_fuse_maxb": 1 << 20, "slab_small": 2}.get(k, 1))
    synth._model.close()
