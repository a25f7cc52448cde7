"""Run the headline configuration several times and compare the waveform batch bit for bit (races in the persistent /
register-ring kernels would show up as run-to-run differences)."""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
digests = []
for mode in ("const7", "uniform"):
    ph, pu, T, spk, dur = synthetic.batch(32, 128, 0, mode)
    pad = np.maximum(689, dur.sum(axis=1)).astype(np.int32)
    for i in range(n):
        out = ctx.synthesize(ph, pu, T, spk, dur, pad, want_mel=True)
        d = hashlib.sha256(out["wav"].tobytes() + out["mel"].tobytes()).hexdigest()[:16]
        digests.append((mode, d))
    ds = {d for m, d in digests if m == mode}
    print(mode, "runs", n, "distinct digests", len(ds), sorted(ds))
    assert len(ds) == 1, "non-deterministic output"
print("deterministic")
