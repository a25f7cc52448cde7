"""Dev aid: time the vocoder's launches per shape with a libzvx variant built by tools/exp_build.sh.
   python tools/exp_slab.py <mask> [<mask> ...]    (mask 0 = the product library).  Results of masks != 0 are wrong by design."""
import os, sys, subprocess, re, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 or (len(sys.argv) == 2 and sys.argv[1] != "child"):
    for m in sys.argv[1:]:
        env = dict(os.environ, ZVX_EXP_MASK=m)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        agg = collections.OrderedDict()
        for line in r.stderr.splitlines() + r.stdout.splitlines():
            mm = re.match(r"launch (\S+)\s+rows=(\d+)\s+N=(\d+)\s+K=(\d+)\s+taps=(\d+)\s+res=(\d+) fused=(\d+)\s+([\d.]+) ms\s+([\d.]+) TF/s", line)
            if mm:
                key = mm.group(1, 2, 3, 4, 5, 6)
                agg.setdefault(key, []).append((float(mm.group(8)), float(mm.group(9))))
            elif line.startswith("vocoder"):
                print(f"[exp {m}] {line}")
        for k, v in agg.items():
            if not k[0].startswith("convslab"): continue
            ms = sum(x[0] for x in v) / len(v); tf = sum(x[1] for x in v) / len(v)
            print(f"[exp {m}] {k[0]:24s} rows={k[1]:8s} N={k[2]:5s} K={k[3]:5s} taps={k[4]:3s} res={k[5]} x{len(v):2d}  {ms:7.3f} ms {tf:7.1f} TF/s")
        if r.returncode: print(r.stderr[-2000:])
    sys.exit(0)
import numpy as np
sys.path.insert(0, ROOT)
from zerovox_amd import config as zcfg, weights as zw, pack, _lib
m = os.environ.get("ZVX_EXP_MASK", "0")
if m != "0": _lib.LIB_PATH = os.path.join(ROOT, "zerovox_amd", f"libzvx_exp{m}.so")
cfg = zcfg.medium_modelcfg("styletts"); sd = zw.tts_state_dict(cfg, 0)
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
man, blob = pack.pack_model(cfg, sd, h, hsd, "bf16")
ctx = _lib.Context(man, blob, 0)
rng = np.random.default_rng(5)
B, Pn = 32, 896
mel = rng.standard_normal((B, Pn, 80)).astype(np.float32); P = np.full(B, Pn, np.int32)
for _ in range(3): ctx.vocode_mel(mel, P)
ctx.set_int("profile", 2); ctx.reset_stats()
for _ in range(3): ctx.vocode_mel(mel, P)
print(f"vocoder {ctx.stage_times()['vocoder']/1:.3f} ms (3 runs)")
ctx.set_int("shape_log", 1); ctx.reset_stats()
for _ in range(3): ctx.vocode_mel(mel, P)
ctx.stage_times()
