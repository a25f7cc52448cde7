"""bf16-mode error budget on the reference's own fixture e2e_styletts_v1_T64 (tests/golden): where the waveform error of the
benchmarked precision mode comes from.  Stages are isolated by composing an f32 context and a bf16 context:
    mel error of the bf16 decoder alone;  waveform error of the bf16 vocoder fed the EXACT (golden) mel;
    waveform error of the f32 vocoder fed the bf16 decoder's mel;  end-to-end bf16.
Prints max / rms errors; the tolerances of tests/test_gpu_parity.py are derived from these."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", (sys.argv[1] if len(sys.argv) > 1 else "e2e_styletts_v1_T64") + ".npz"))
kind, voc = str(g["decoder_kind"]), str(g["vocoder"])
cfg = zcfg.medium_modelcfg(kind); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config(voc); hsd = zw.hifigan_state_dict(h, 0)
ctx = {}
for prec in ("f32", "bf16"):
    man, blob = pack.pack_model(cfg, sd, h, hsd, prec)
    ctx[prec] = _lib.Context(man, blob, 0)
T = len(g["phoneme"]); ml = int(g["mel_len"]); pad = int(g["pad_to"])
dur = g["duration"][None] if bool(g["forced"]) else None

def err(a, b):
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    return f"max {np.abs(d).max():.3e}  rms {np.sqrt((d ** 2).mean()):.3e}  (ref rms {np.sqrt((np.asarray(b, np.float64) ** 2).mean()):.3e})"

mels = {}
for prec in ("f32", "bf16"):
    c = ctx[prec]
    mel_len, *_ = c.encode(g["phoneme"][None], g["puncts"][None], np.array([T], np.int32), g["spk"][None], dur)
    assert int(mel_len[0]) == ml
    mels[prec] = c.decode(1, ml)[0, :ml]
    print(f"decoder {prec:4s}: mel  {err(mels[prec], g['mel'].T)}")
P = max(pad, ml)
def voc_on(prec, mel):
    m = np.zeros((1, P, 80), np.float32); m[0, :ml] = mel
    return ctx[prec].vocode_mel(m, np.array([P], np.int32))[0, :ml * 256]
ref = g["wav"]
print(f"vocoder f32  on golden mel : wav  {err(voc_on('f32', g['mel'].T), ref)}")
print(f"vocoder bf16 on golden mel : wav  {err(voc_on('bf16', g['mel'].T), ref)}     <- vocoder storage rounding alone")
print(f"vocoder f32  on bf16 mel   : wav  {err(voc_on('f32', mels['bf16']), ref)}     <- decoder error propagated through an exact vocoder")
print(f"vocoder bf16 on bf16 mel   : wav  {err(voc_on('bf16', mels['bf16']), ref)}     <- end to end (bf16 mode)")
# sensitivity of the vocoder: a mel perturbed by bf16 rounding of the mel itself
mq = (g["mel"].T.astype(np.float32).view(np.uint32) + 0x8000 & 0xffff0000).view(np.float32)
print(f"vocoder f32  on bf16-ROUNDED golden mel: wav {err(voc_on('f32', mq), ref)}     <- what rounding the mel to 8 bits alone costs")
