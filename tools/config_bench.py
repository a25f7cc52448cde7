"""Secondary BASELINE configs (not the headline bench line): #4 HiFi-GAN alone, #5 speaker encoder, FS2 decoder,
variable-length batch, V2 vocoder.  Prints one line per config."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic

def ctx_for(kind, voc, prec="bf16"):
    cfg = zcfg.medium_modelcfg(kind); sd = zw.tts_state_dict(cfg, 0)
    h = zcfg.hifigan_config(voc); hsd = zw.hifigan_state_dict(h, 0)
    man, blob = pack.pack_model(cfg, sd, h, hsd, prec)
    return _lib.Context(man, blob, 0)

def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    t = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t) / n

which = sys.argv[1:] or ["voc", "spk", "fs2", "var", "v2"]
if "voc" in which:      # config #4: 1024-frame random mel -> waveform, B in {1, 32}
    ctx = ctx_for("styletts", "v1")
    for B in (1, 32):
        mel = np.random.default_rng(7).standard_normal((B, 1024, 80)).astype(np.float32); P = np.full(B, 1024, np.int32)
        dt = timeit(lambda: ctx.vocode_mel(mel, P))
        print(f"config4 HiFi-GAN V1 alone B={B} x 1024 frames: {dt*1e3:.2f} ms/call (incl. H2D mel + D2H wav), {B*262144/dt/1e6:.1f} M samples/s, "
              f"{B*628.84e9/dt/1e12:.1f} TFLOP/s algorithmic", flush=True)
    ctx.close()
if "spk" in which:      # config #5: 1000 x 3 s reference mels, batches of 50
    ctx = ctx_for("styletts", "tiny")
    mels = np.random.default_rng(8).standard_normal((50, 258, 80)).astype(np.float32); lens = np.full(50, 258, np.int32)
    dt = timeit(lambda: ctx.spkemb(mels, lens), n=20)
    e = ctx.spkemb(mels, lens)
    print(f"config5 ResNetSE34V2: 50 clips x 258 frames: {dt*1e3:.2f} ms/call -> {50/dt:.0f} embeds/s ({1000/(50/dt):.2f} s per 1000 clips), "
          f"{50*11.81e9/dt/1e12:.1f} TFLOP/s algorithmic, |e|-1 max {np.abs(np.linalg.norm(e,axis=1)-1).max():.1e}", flush=True)
    ctx.close()
for tag, kind, voc, mode in (("fs2", "fastspeech2", "v1", "const7"), ("var", "styletts", "v1", "uniform"), ("v2", "styletts", "v2", "const7")):
    if tag not in which: continue
    ctx = ctx_for(kind, voc)
    ph, pu, T, spk, dur = synthetic.batch(32, 128, 0, mode)
    L = dur.sum(axis=1); pad = np.maximum(689, L).astype(np.int32)
    ctx.set_int("profile", 1)
    dt = timeit(lambda: ctx.synthesize(ph, pu, T, spk, dur, pad, want_mel=False))
    st = ctx.stage_times()
    print(f"{tag}: decoder={kind} vocoder={voc} durations={mode} (frames {L.min()}..{L.max()}): {dt*1e3:.2f} ms/step (host wav copy incl.), "
          f"{L.sum()*256/dt/1e6:.1f} M samples/s; stages ms: " + ", ".join(f"{k}={v:.2f}" for k, v in st.items() if v > 0), flush=True)
    ctx.close()
