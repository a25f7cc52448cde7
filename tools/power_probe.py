"""Board power and clocks while the headline workload runs (rocm-smi sampled from a side thread; dev aid for DESIGN.md section 4).
   python tools/power_probe.py [seconds] [zeros]    'zeros': the same launches with all-zero activations (zero mel, zero biases; weights unchanged)."""
import json, os, subprocess, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
zeros = len(sys.argv) > 2 and sys.argv[2] == "zeros"
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
if zeros:                                   # zero biases + zero mel: every activation of the vocoder is exactly 0, the weights stay as they are
    hsd = {k: (np.zeros_like(v) if "bias" in k else v) for k, v in hsd.items()}
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
B, Pn = 32, 896
mel = (np.zeros((B, Pn, 80)) if zeros else np.random.default_rng(5).standard_normal((B, Pn, 80))).astype(np.float32)
P = np.full(B, Pn, np.int32)
samples, stop = [], False


def smi():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out); c = d[sorted(d)[0]]
            samples.append({k: v for k, v in c.items() if any(t in k.lower() for t in ("power", "sclk", "mclk", "temperature (sensor junction"))})
        except Exception as e:
            samples.append({"error": str(e)})
        time.sleep(0.2)


for _ in range(3): ctx.vocode_mel(mel, P)
th = threading.Thread(target=smi); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    ctx.vocode_mel(mel, P); n += 1
dt = time.time() - t0
stop = True; th.join()
print(f"{'zero' if zeros else 'random'} operands: {n} vocoder passes in {dt:.2f} s = {1e3 * dt / n:.2f} ms each (host copies included)")
for s_ in samples[1:]:
    print(s_)
