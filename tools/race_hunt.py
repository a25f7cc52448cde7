"""Race screen: the benchmark batch (32 x 128 phonemes) synthesised N times, every result compared bit for bit with the first --
encoder + decoder (mel) and the vocoder alone under several kernel selections.  A rare mismatch (a wrong 128-row block in ~1 % of
runs) is how the NT = 3 DMA-retire race of pairstream.hip showed up.     python tools/race_hunt.py [N=100]"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ph, pu, T, spk, dur = synthetic.batch(32, 128, 192, "const7")
pad_to = np.full(32, 896, np.int32)
r = ctx.synthesize(ph, pu, T, spk, dur, pad_to, want_mel=True)
mel0, P = r["mel"], r["mel_len"]
defaults = (("attn_f32", 1), ("resstream", 1), ("pairstream", 1), ("rs_opt", 3))
def reset(sets):
    for k, v in defaults: ctx.set_int(k, v)
    for k, v in sets.items(): ctx.set_int(k, v)
# 1. encoder + decoder: mel determinism
reset({})
bad = 0
for i in range(N):
    m = ctx.synthesize(ph, pu, T, spk, dur, pad_to, want_mel=True)["mel"]
    if not np.array_equal(m, mel0): bad += 1
print(f"mel (encoder + decoder) mismatches: {bad} / {N}", flush=True)
# 2. vocoder alone
for name, sets in (("default", {}), ("pairstream=3", {"pairstream": 3}), ("resstream=0", {"resstream": 0}), ("default", {})):
    reset(sets)
    outs = [ctx.vocode_mel(mel0, P) for _ in range(3)]
    ref = outs[0] if np.array_equal(outs[0], outs[1]) else outs[2]
    bad = []
    for i in range(N):
        w = ctx.vocode_mel(mel0, P)
        if not np.array_equal(w, ref):
            d = np.abs(w - ref); u = np.argwhere(d.max(1) > 0)[:, 0]
            bad.append([(int(b), int(np.argmax(d[b] > 0)), int(len(d[b]) - np.argmax(d[b][::-1] > 0)), float(d[b].max())) for b in u[:3]])
    print(f"vocoder {name}: mismatches {len(bad)} / {N}: {bad[:6]}", flush=True)
