"""Throughput of the benchmark step with one context (one stream, steps back to back) and with two contexts on two streams whose
steps alternate (batch i+1's encoder / decoder -- latency-paced, 5-28 % of the matrix roof -- under batch i's vocoder).
   python tools/two_contexts.py [steps=60]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
B, T = 32, 128
ph, pu, Tl, spk, dur = synthetic.batch(B, T, 0, "const7")
N = int(dur[0].sum()) * 256
pad = np.full(B, 896, np.int32)
for nctx in (1, 2, 3, 1, 2):
    ctxs = [_lib.Context(man, blob, 0) for _ in range(nctx)]
    wavs = [c.dev_alloc(B * N * 4) for c in ctxs]
    def step(i):
        c = ctxs[i % nctx]
        c.synthesize(ph, pu, Tl, spk, dur, pad, want_mel=False, wav_device_ptr=wavs[i % nctx], wav_stride=N, no_sync=True)
    for i in range(6): step(i)
    for c in ctxs: c.sync()
    t0 = time.time()
    for i in range(K): step(i)
    for c in ctxs: c.sync()
    dt = time.time() - t0
    print(f"{nctx} context(s): {1e3 * dt / K:.3f} ms per step = {B * N * K / dt / 1e6:.1f} M samples/s", flush=True)
    for c in ctxs: c.close()

# the same through the host API (waveforms delivered to host memory, worker threads): ZeroVox.synthesize_batches
from zerovox_amd.model import ZeroVox
m = ZeroVox(cfg, sd, h, hsd, infer_device="cuda:0", precision="bf16")
batch = dict(phoneme=ph, puncts=pu, T=Tl, style_embed=spk, duration=dur, pad_to=pad)
for _ in range(3): m.synthesize_batch(ph, pu, Tl, spk, dur, pad, want_mel=False)
t0 = time.time()
for _ in range(K): m.synthesize_batch(ph, pu, Tl, spk, dur, pad, want_mel=False)
dt1 = time.time() - t0
for n in (2, 3):
    list(m.synthesize_batches((batch for _ in range(4)), in_flight=n))
    t0 = time.time()
    for _r in m.synthesize_batches((batch for _ in range(K)), in_flight=n): pass
    dt = time.time() - t0
    print(f"host API, waveforms to host: sequential {1e3 * dt1 / K:.3f} ms per batch, {n} in flight {1e3 * dt / K:.3f} ms per batch ({B * N * K / dt / 1e6:.1f} M samples/s)", flush=True)
