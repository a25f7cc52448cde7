"""Torch-free RCCL self-test on one GPU: dlopen of the system librccl, a real one-rank communicator, send/recv to self, barrier."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
assert "torch" not in sys.modules
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("tiny"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
ctx.comm_init(_lib.Context.comm_unique_id(), 0, 1)
ph, pu, T, spk, dur = synthetic.batch(2, 10, 0, "uniform")
N = int(dur.sum(axis=1).max()) * 256
ref = ctx.synthesize(ph, pu, T, spk, dur, np.full(2, 64, np.int32), want_mel=False)["wav"]
a, b = ctx.dev_alloc(2 * N * 4), ctx.dev_alloc(2 * N * 4)
ctx.synthesize(ph, pu, T, spk, dur, np.full(2, 64, np.int32), want_mel=False, wav_device_ptr=a, wav_stride=N, no_sync=True)
ctx.comm_gather(a, 2 * N * 4, b, root=0, no_sync=True)
ctx.comm_barrier()
print("torch-free RCCL self-test:", "OK" if np.array_equal(ctx.dev_to_host(b, (2, N), np.float32), ref) and "torch" not in sys.modules else "FAILED")
