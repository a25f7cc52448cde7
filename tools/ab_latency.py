"""Single-request sizing A/B: vocoder alone (B = 1, 448 / 1024 frames) and the whole 64-phoneme utterance under
   rs_seg_min (streaming-ResBlock segment floor) and pairstream modes 1 (decline small jobs) / 3 (1024-row segments) / 4 (256-row segments);
   bit-equality of every variant against the default.   python tools/ab_latency.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
rng = np.random.default_rng(2)
variants = (("round-2 sizing", {"rs_seg_min": -1, "pairstream": 1, "slab_small": 0, "norm_fuse_maxb": 0, "va_overlap_maxb": 0, "voc_overlap_maxb": 0}), ("short resstream segments", {"rs_seg_min": 0, "pairstream": 1, "slab_small": 0}),
            ("+ small conv-slab tiles", {"rs_seg_min": 0, "pairstream": 1, "slab_small": 1}), ("+ 32-channel tiles for one-row-tile launches", {"rs_seg_min": 0, "pairstream": 1, "slab_small": 2}),
            ("+ one-launch InstanceNorm", {"rs_seg_min": 0, "pairstream": 1, "slab_small": 2, "norm_fuse_maxb": 1 << 20}),
            ("+ duration beside pitch predictor", {"rs_seg_min": 0, "pairstream": 1, "slab_small": 2, "norm_fuse_maxb": 1 << 20, "va_overlap_maxb": 1 << 20}),
            ("+ ResBlocks side by side, B <= 2", {"rs_seg_min": 0, "pairstream": 1, "slab_small": 2, "norm_fuse_maxb": 1 << 20, "va_overlap_maxb": 1 << 20, "voc_overlap_maxb": 2}),
            ("+ ResBlocks side by side (default)", {"rs_seg_min": 0, "pairstream": 1, "slab_small": 2, "norm_fuse_maxb": 1 << 20, "va_overlap_maxb": 1 << 20, "voc_overlap_maxb": 1 << 20}),
            ("+ pair kernel, 256-row segments", {"rs_seg_min": 0, "pairstream": 4, "slab_small": 2, "norm_fuse_maxb": 1 << 20, "va_overlap_maxb": 1 << 20, "voc_overlap_maxb": 1 << 20}))
for (B, P) in ((1, 448), (1, 1024), (2, 448), (4, 448)):
    mel = rng.standard_normal((B, P, 80)).astype(np.float32); L = np.full(B, P, np.int32)
    ref = None
    for name, sets in variants:
        for k, v in sets.items(): ctx.set_int(k, v)
        for _ in range(3): w = ctx.vocode_mel(mel, L)
        ctx.set_int("profile", 1); ts = []
        for _ in range(10): w = ctx.vocode_mel(mel, L); ts.append(ctx.stage_times()["vocoder"])
        ctx.set_int("profile", 0)
        if ref is None: ref = w
        print(f"vocoder B={B} P={P} {name:42s}: {np.median(ts):.3f} ms  bit-equal to the first: {np.array_equal(w, ref)}", flush=True)
ph, pu, Tl, spk, dur = synthetic.batch(1, 64, 0, "const7"); pad = np.full(1, 448, np.int32)
ref = None
for name, sets in variants:
    for k, v in sets.items(): ctx.set_int(k, v)
    for _ in range(3): r = ctx.synthesize(ph, pu, Tl, spk, dur, pad, want_mel=True)
    if ref is None: ref = r
    print(f"   mel / wav bit-equal to the first variant: {np.array_equal(r['mel'], ref['mel'])} / {np.array_equal(r['wav'], ref['wav'])}")
    t0 = time.time()
    for _ in range(30): ctx.synthesize(ph, pu, Tl, spk, dur, pad, want_mel=False)
    print(f"64-phoneme utterance, {name:42s}: {(time.time() - t0) / 30 * 1e3:.3f} ms per call (host wall, waveform to host)")
