"""Digest of a refresh pass (profiles/<prefix>_bench_*.json, traffic*.json): the numbers DESIGN.md section 4 quotes, in one place.
    python tools/profile_digest.py [prefix=r04] [dir=profiles]"""
import glob, json, os, sys
pre = sys.argv[1] if len(sys.argv) > 1 else "r04"
d = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def load(name):
    f = os.path.join(d, f"{pre}_bench_{name}.json") if pre else os.path.join(d, f"bench_{name}.json")
    try:
        return json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        return None


for name in ("n1", "n1_serial", "n1_exact_encoder", "n1_bf16_planes", "n1_host_out", "n1_in_flight2", "n1_fast_box", "cfg4", "cfg4_b1", "cfg5", "fs2dec", "v2", "v3", "b1_t64"):
    j = load(name)
    if not j:
        print(f"{name:18s} (missing)")
        continue
    r = j.get("roofline", {})
    tr = r.get("alone", {})
    print(f"{name:18s} {j['ms_per_step']:8.3f} ms  {j['value']:.4g} {j['unit']:10s} dom {r.get('kernel', '-'):24s} frac {r.get('frac', 0):.3f} avg {r.get('avg_launch_ms', 0):.4f} ms"
          + (f"  alone-on-chip frac {tr['frac']:.3f} avg {tr['avg_launch_ms']:.4f}" if tr else "") + f"  traffic {r.get('traffic')}  sha {j.get('src_sha16')}")
j = load("n1")
if j:
    print("stage_ms_one_step_alone", {k: round(v, 3) for k, v in (j.get("stage_ms_one_step_alone") or {}).items()})
    print("clock smi / profiled / mfma busy:", j["roofline"].get("smi_gfx_clock_GHz"), j["roofline"].get("profiled_clock_GHz"), j["roofline"].get("profiled_mfma_busy_frac"))
    cb = j.get("cpu_baseline", {})
    print("cpu_baseline", round(cb.get("value", 0)), cb.get("unit"), cb.get("cores"), cb.get("sample", "")[:160])
    for s in j.get("roofline_per_stage", []):
        print(f"   {s['stage']:14s} {s['launches']:3d} {s['ms']:7.3f} ms {s['TFLOPs']:8.1f} TF/s  mfma {s['frac_mfma']:.3f}  hbm {s['frac_hbm']:.3f}")
for f in sorted(glob.glob(os.path.join(d, "traffic*.json"))):
    t = json.load(open(f))
    dom = max(((k, v) for k, v in t.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v), key=lambda kv: kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])
    print(os.path.basename(f), t.get("src_sha16"), dom[0], dom[1]["launches"], round(dom[1]["hbm_bytes_per_launch"] / 1e6, 1), "MB/launch", "clk", round(dom[1].get("eff_clock_GHz", 0), 3), "busy", round(dom[1].get("mfma_busy_frac", 0), 3))
