import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import zvx_oracle as O
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic
kind = sys.argv[1] if len(sys.argv) > 1 else "styletts"
cfg = zcfg.medium_modelcfg(kind); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("tiny"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "f32")
ctx = _lib.Context(man, blob, 0)
def err(a, b): return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())
for Ts, durs in (([1, 9], [[2], [0, 3, 0, 0, 5, 1, 0, 2, 0]]), ([1], [[2]]), ([1], [[5]]), ([2], [[1, 1]]), ([3], [[1, 1, 1]]), ([9], [[0, 3, 0, 0, 5, 1, 0, 2, 0]])):
    B, Tm = len(Ts), max(Ts)
    ph = np.zeros((B, Tm), np.int32); pu = np.zeros((B, Tm), np.int32); dur = np.zeros((B, Tm), np.int32); spk = np.zeros((B, 528), np.float32)
    for b, T in enumerate(Ts):
        p, q, s, _ = synthetic.utterance(T, 95 + b, None)
        ph[b, :T], pu[b, :T], spk[b] = p, q, s; dur[b, :T] = durs[b]
    ml, logd, pitch, en = ctx.encode(ph, pu, np.array(Ts, np.int32), spk, dur)
    L = int(ml.max())
    eo = ctx.fetch("encoder_out", (B, Tm, 528)); fe = ctx.fetch("features", (B, L, 528)); mel = ctx.decode(B, L)
    for b, T in enumerate(Ts):
        ref = O.fs2_encoder(ph[b, :T], pu[b, :T], spk[b], sd, cfg, dur[b, :T]); m = ref["mel_len"]
        rm = O.mel_decoder(ref["features"], spk[b], sd, cfg)
        print(Ts, "utt", b, "mel_len", ml[b], m, "enc", err(eo[b, :T], ref["encoder_out"]), "logd", err(logd[b, :T], ref["log_duration"]),
              "feat", err(fe[b, :m], ref["features"]), "mel", err(mel[b, :m], rm), flush=True)
