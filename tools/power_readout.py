"""Power / clock / throttle readout while the headline workload (or the vocoder alone) loops: which limiter holds the shader clock,
and how many joules one step costs.

    python tools/power_readout.py [--seconds 15] [--what step|vocoder] [--pairstream V] [--zeros]

Sources (all read with the GPU's own management tool, nothing inferred from kernel durations):
  * `amd-smi metric -p -c -V -l`   socket power, current / max gfx clocks, voltage, DPM level      (sampled from a side thread)
  * `amd-smi metric -E`            the energy accumulator, before and after the loop -> J per step
  * `amd-smi metric -v`            the THROTTLE (violation) accumulators, before and after: the counters that advanced during the
                                   loop name the limiter (PPT = package power, socket / VR / HBM thermal, PROCHOT, gfx-clock-below-
                                   host-limit by power / thermal / total), `per_*` fields give the share of time each was active
--pairstream -1 / 0 reproduce round 2's launch set for the C = 128 stage (two conv-slab launches, resfuse for k = 3); 2 is this
round's default.  --zeros: the same launches on all-zero activations (zero mel and biases, weights unchanged)."""
import argparse, json, os, subprocess, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd import config as zcfg, weights as zw, pack, _lib, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=15.0)
ap.add_argument("--what", default="step", choices=["step", "vocoder"])
ap.add_argument("--pairstream", type=int, default=2)
ap.add_argument("--zeros", action="store_true")
ap.add_argument("--voc-f16", type=int, default=1, help="0: the bf16 vocoder kernels of rounds 1-4 (A/B of the round-5 IEEE-half default)")
args = ap.parse_args()


def smi(flags):
    try:
        out = subprocess.run(["amd-smi", "metric", "-g", "0"] + flags + ["--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        d = d.get("gpu_data", d) if isinstance(d, dict) else d
        return d[0] if isinstance(d, list) else d
    except Exception as e:
        return {"error": repr(e)}


def flat(d, pre=""):
    out = {}
    if isinstance(d, dict):
        if set(d) == {"value", "unit"}:
            return {pre[:-1]: d["value"]}
        for k, v in d.items():
            out.update(flat(v, pre + str(k) + "."))
    elif isinstance(d, list):
        if all(not isinstance(v, (dict, list)) for v in d):
            if any(v not in ("N/A", 0) for v in d):
                out[pre[:-1]] = d
        else:
            for i, v in enumerate(d):
                out.update(flat(v, pre + str(i) + "."))
    elif d != "N/A":
        out[pre[:-1]] = d
    return out


cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
if args.zeros:
    hsd = {k: (np.zeros_like(v) if "bias" in k else v) for k, v in hsd.items()}
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
ctx.set_int("pairstream", 1 if args.pairstream == 2 else args.pairstream)
ctx.set_int("voc_f16", args.voc_f16)
B, T = 32, 128
if args.what == "step" and not args.zeros:
    ph, pu, Tlen, spk, dur = synthetic.batch(B, T, first_utt=0, dur_mode="const7")
    N = int(dur[0].sum()) * 256
    pad_to = np.full(B, 896, np.int32)
    wav_d = ctx.dev_alloc(B * N * 4)
    step = lambda: ctx.synthesize(ph, pu, Tlen, spk, dur, pad_to, want_mel=False, wav_device_ptr=wav_d, wav_stride=N, no_sync=True)
    what = "whole step (config 2: B = 32 x 128 phonemes -> 896 frames, styledec + HiFi-GAN V1, waveform left on the device)"
else:
    Pn = 896
    mel = (np.zeros((B, Pn, 80)) if args.zeros else np.random.default_rng(5).standard_normal((B, Pn, 80))).astype(np.float32)
    mel_d = ctx.dev_alloc(mel.nbytes); ctx.dev_from_host(mel_d, mel)
    wav_d = ctx.dev_alloc(B * Pn * 256 * 4)
    P = np.full(B, Pn, np.int32)
    step = lambda: ctx.vocode_mel_device(mel_d, P, Pn, wav_d, Pn * 256, no_sync=True)
    what = f"HiFi-GAN V1 vocoder alone on 32 x 896-frame {'all-zero' if args.zeros else 'N(0,1)'} mels (device-resident)"

for _ in range(20): step()
ctx.sync()
samples, stop = [], [False]


def sampler():
    while not stop[0]:
        samples.append((time.time(), flat(smi(["-p", "-c", "-V", "-l"]))))
        time.sleep(0.25)


before = flat(smi(["-E", "-v", "-p"]))
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < args.seconds:
    for _ in range(10): step()
    ctx.sync(); n += 10
dt = time.time() - t0
stop[0] = True; th.join()
after = flat(smi(["-E", "-v", "-p"]))

print(f"# tools/power_readout.py  --what {args.what} --pairstream {args.pairstream} --voc-f16 {args.voc_f16}{' --zeros' if args.zeros else ''}")
print(f"workload: {what}")
print(f"{n} steps in {dt:.2f} s = {1e3 * dt / n:.3f} ms per step (host loop, synchronised every 10 steps)")
ek = [k for k in before if "energy" in k.lower() and isinstance(before[k], (int, float)) and isinstance(after.get(k), (int, float))]
for k in ek:
    d = after[k] - before[k]
    print(f"energy accumulator {k}: {before[k]} -> {after[k]}  (delta {d:.3f}; if joules: {d / n:.3f} J per step, {d / dt:.1f} W average)")
print("throttle / violation accumulators that ADVANCED during the loop (the limiter), with the time share amd-smi reports:")
adv = 0
for k in sorted(after):
    if k in ek:
        continue
    a, b_ = before.get(k), after[k]
    if isinstance(a, (int, float)) and isinstance(b_, (int, float)) and ("acc" in k.lower() or "violation" in k.lower() or "throttle" in k.lower()):
        if b_ != a:
            adv += 1
            print(f"   {k}: {a} -> {b_}  (+{b_ - a})")
for k in sorted(after):
    if "per_" in k.lower() and after[k] not in (0, "N/A"):
        print(f"   {k}: {after[k]} (reading after the loop)")
if not adv:
    print("   none advanced")
print("all throttle fields after the loop:")
for k in sorted(after):
    if k not in ek:
        print(f"   {k} = {after[k]}")
print("samples during the loop (t s | " + "socket power W | gfx clocks MHz | voltage | perf level):")
keys = sorted({k for _, s in samples for k in s})
pk = [k for k in keys if "power" in k.lower()]
ck = [k for k in keys if "gfx" in k.lower() and "clk" in k.lower() and isinstance(samples[-1][1].get(k), (int, float))]
ok = [k for k in keys if k not in pk and k not in ck and ("volt" in k.lower() or "perf" in k.lower()) and samples[-1][1].get(k) not in (None, "N/A")]
for ts, s in samples[1:]:
    clk = [s.get(k) for k in ck if "clk" in k and k.endswith(".clk")]
    print(f"   {ts - t0:6.2f} | " + " ".join(f"{k.split('.')[-1]}={s.get(k)}" for k in pk) + " | gfx clk " +
          (f"min {min(clk)} max {max(clk)} mean {sum(clk) / len(clk):.0f}" if clk else str({k: s.get(k) for k in ck[:4]})) +
          " | " + " ".join(f"{k.split('.', 1)[-1]}={s.get(k)}" for k in ok[:6]))
if samples:
    print("one full sample (gfx / memory / fabric clocks, voltages, performance level):")
    for k in sorted(samples[-1][1]):
        if not any(t in k for t in ("dclk", "vclk")):
            print(f"   {k} = {samples[-1][1][k]}")
