"""First-audio latency of the chunked vocoder: one 896-frame mel, HiFi-GAN V1, batch 1.
   python tools/stream_latency.py [out.json]   -> per chunk size: time to the first waveform chunk, time for all chunks, and
   the whole-utterance call for comparison (host mel in, host waveform out: the PCIe copies are inside every figure)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw
from zerovox_amd.model import ZeroVox

cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
model = ZeroVox(cfg, sd, h, hsd, "cuda:0", "bf16")
ctx = model.ctx
STREAM_HALO = ZeroVox.STREAM_HALO
rng = np.random.default_rng(3)
L = 896
mel = rng.standard_normal((L, 80)).astype(np.float32)
whole = ctx.vocode_mel(mel[None], np.array([L], np.int32))[0]


def best(f, n=20):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


res = {"workload": f"one {L}-frame mel ({L * 256 / 22050:.2f} s of audio), HiFi-GAN V1 bf16, batch 1, halo {STREAM_HALO} frames per side",
       "whole_utterance_ms": best(lambda: ctx.vocode_mel(mel[None], np.array([L], np.int32))), "chunks": []}
for cf in (16, 32, 64, 128, 256):
    got = np.concatenate(list(model.vocode_stream(mel, chunk_frames=cf)))
    first = best(lambda: next(iter(model.vocode_stream(mel, chunk_frames=cf))))
    total = best(lambda: list(model.vocode_stream(mel, chunk_frames=cf)), n=5)
    res["chunks"].append({"chunk_frames": cf, "chunk_audio_ms": round(cf * 256 / 22.05, 1), "first_chunk_ms": round(first, 3), "all_chunks_ms": round(total, 3),
                          "bit_equal_to_whole": bool(np.array_equal(got, whole)), "max_abs_diff": float(np.abs(got - whole).max())})
    print(res["chunks"][-1], flush=True)
print(json.dumps(res))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
