"""Per-stage arithmetic of the generator (zvx_set_int("voc_f16_stages", mask): bit k = domain k in IEEE half, else bf16; domain 0 = mel / conv_pre,
domain i = upsampling stage i): waveform error of the headline utterance against the f32 oracle and step time, per mask.
    python tools/ab_voc_stages.py [masks ...]        (GPU box; ~1 min: one oracle call + a few synthesis calls per mask)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import zvx_oracle as O                       # checker only
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic

masks = [int(m, 0) for m in sys.argv[1:]] or [31, 27, 25, 29, 17, 0]
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
ph, pu, T, spk, dur = synthetic.batch(32, 128, 0, "const7")
pad_to = np.full(32, 896, np.int32)
b = 7
ref = O.inference_ex(sd, hsd, cfg, h, ph[b], pu[b], spk[b], duration=dur[b], pad_to=896)
N = 896 * 256
buf = ctx.dev_alloc(32 * N * 4)
for m in masks:
    ctx.set_int("voc_f16", 1 if m else 0); ctx.set_int("voc_f16_stages", m if m else 31)
    out = ctx.synthesize(ph, pu, T, spk, dur, pad_to, want_mel=False)
    e = out["wav"][b, :N].astype(np.float64) - ref["wav"]
    # vocoder alone on the oracle's mel: the generator's own error
    wv = ctx.vocode_mel(ref["mel"].T[None].astype(np.float32), np.array([896], np.int32))[0][:N]
    rv = O.hifigan_generator(ref["mel"], hsd, h)
    ev = wv.astype(np.float64) - rv
    for _ in range(5): ctx.synthesize(ph, pu, T, spk, dur, pad_to, want_mel=False, wav_device_ptr=buf, wav_stride=N, no_sync=True)
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(60): ctx.synthesize(ph, pu, T, spk, dur, pad_to, want_mel=False, wav_device_ptr=buf, wav_stride=N, no_sync=True)
    ctx.sync(); ms = (time.perf_counter() - t0) / 60 * 1e3
    print(f"mask {m:#07b}: step {ms:7.3f} ms | e2e wav err max {np.abs(e).max():.3e} rms {np.sqrt((e ** 2).mean()):.3e} | vocoder alone max {np.abs(ev).max():.3e} rms {np.sqrt((ev ** 2).mean()):.3e}", flush=True)
