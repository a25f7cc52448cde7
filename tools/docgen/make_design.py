#!/usr/bin/env python3
"""Assemble DESIGN.md and README.md: tools/docgen/design_base.md (sections 1-3, 5-7) + design_sec4.tmpl.md with the figures of profiles/<round>_* + design_tail.md.
    python tools/docgen/make_design.py r06"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(ROOT, "profiles")
def line(name):
    return json.loads(open(os.path.join(P, f"{R}_bench_{name}.json")).read().strip().splitlines()[-1])
def ms(name):
    try: return f"{line(name)['ms_per_step']:.2f}"
    except Exception: return "n/a"
n1 = line("n1")
ab = n1.get("ab_vocoder_arithmetic", {})
rf = n1["roofline"]
sub = {
    "REFRESH_MS": f"{n1['ms_per_step']:.2f}", "REFRESH_VAL": f"{n1['value'] / 1e6:.1f}", "REFRESH_AGAIN": ms("n1_again"),
    "REFRESH_TAIL": f"{n1['busy_tail']['ms_per_step']:.2f}" if n1.get("busy_tail") else "n/a",
    "REFRESH_HALF": ms("n1_voc_half"), "REFRESH_ABHALF": f"{ab['all_half']['ms_per_step']:.2f}" if "all_half" in ab else "n/a",
    "REFRESH_BF16": ms("n1_voc_bf16"), "REFRESH_ABBF16": f"{ab['all_bf16']['ms_per_step']:.2f}" if "all_bf16" in ab else "n/a",
    "REFRESH_HOSTSYNC": ms("n1_host_out_sync"), "REFRESH_HOST": ms("n1_host_out"), "REFRESH_SERIAL": ms("n1_serial"),
    "REFRESH_F32MODE": f"{n1['f32_mode']['ms_per_step']:.1f}" if n1.get("f32_mode", {}).get("ms_per_step") else "n/a", "REFRESH_F32": ms("n1_f32"),
    "REFRESH_EXACT": ms("n1_exact_encoder"), "REFRESH_PLANES": ms("n1_bf16_planes"), "REFRESH_INFLIGHT": ms("n1_in_flight2"),
    "REFRESH_V3U": ms("v3_unfused"), "REFRESH_V3": ms("v3"),
    "DOM_MS": f"{rf['avg_launch_ms']:.3f}", "DOM_TF": f"{rf['achieved']:.0f}", "DOM_FRAC": f"{rf['frac']:.3f}",
    "DOM_ALONE_MS": f"{rf.get('alone', {}).get('avg_launch_ms', 0):.3f}", "DOM_ALONE_FRAC": f"{rf.get('frac_alone', 0):.3f}",
    "DOM_TRAFFIC": f"{(rf.get('traffic') or 0) / 1e6:.0f}", "DOM_BUSY": f"{100 * rf.get('profiled_mfma_busy_frac', 0):.1f}", "DOM_CLK": f"{rf.get('profiled_clock_GHz', 0):.2f}",
    "CPU_BASE": f"{n1.get('cpu_baseline', {}).get('value', 0) / 1e3:.0f}",
    "CFG4B1": ms("cfg4_b1"), "CFG4_FRAC": f"{line('cfg4')['roofline']['frac']:.3f}", "CFG4": ms("cfg4"),
    "CFG5_VAL": f"{line('cfg5')['value'] / 1e3:.1f}", "CFG5": ms("cfg5"), "FS2Y32": ms("fs2dec_y32"), "FS2": ms("fs2dec"),
    "V3U": ms("v3_unfused"), "V2": ms("v2"), "V3": ms("v3"), "B1T64": ms("b1_t64"),
    "STAGE_ALONE": ", ".join(f"{k} {v:.2f}" for k, v in n1["stage_ms_one_step_alone"].items() if v) + " ms",
}
names = {"voc.res2": "vocoder stage 2 ResBlocks (C = 128, `pairstream`, bf16)", "voc.res3": "vocoder stage 3 ResBlocks (C = 64, `resstream`)",
         "voc.res1": "vocoder stage 1 ResBlocks (C = 256, conv-slab × 18)", "decoder": "StyleTTS decoder convolutions", "voc.res4": "vocoder stage 4 ResBlocks (C = 32)",
         "encoder": "phoneme encoder (split products on half planes; ≈ 3× issued)", "decoder.norm": "decoder InstanceNorm / AdaIN passes", "variance": "variance adaptor (exact f32)"}
rows = ["| stage | launches | ms | TFLOP/s | frac MFMA | frac HBM (alg) |", "|---|---|---|---|---|---|"]
ups = [r for r in n1["roofline_per_stage"] if r["stage"].startswith(("voc.up", "voc.pre", "voc.post"))]
for r in n1["roofline_per_stage"]:
    if r["stage"] in names:
        rows.append(f"| {names[r['stage']]} | {r['launches']} | {r['ms']:.2f} | {r['TFLOPs']:.0f} | {r['frac_mfma']:.3f} | {r['frac_hbm']:.2f} |")
rows.append(f"| upsampling convolutions + conv_pre / conv_post | {sum(r['launches'] for r in ups)} | {sum(r['ms'] for r in ups):.2f} | – | – | {min(r['frac_hbm'] for r in ups):.2f}-{max(r['frac_hbm'] for r in ups):.2f} |")
sub["PERSTAGE_TABLE"] = "\n".join(rows)
t = open(os.path.join(ROOT, "tools/docgen/design_sec4.tmpl.md")).read()
for k in sorted(sub, key=len, reverse=True):
    t = t.replace("@" + k + "@", sub[k])
base = open(os.path.join(ROOT, "tools/docgen/design_base.md")).read()
tail = open(os.path.join(ROOT, "tools/docgen/design_tail.md")).read()
i5 = base.index("## 5. Multi-GPU")
i8 = len(base)
out = base[:i5] + t + "\n" + base[i5:].rstrip("\n") + "\n\n" + tail
open(os.path.join(ROOT, "DESIGN.md"), "w").write(out)
print("DESIGN.md", len(out.encode()), "bytes")
rt = open(os.path.join(ROOT, "tools/docgen/readme_numbers.tmpl.md")).read()
for k in sorted(sub, key=len, reverse=True):
    rt = rt.replace("@" + k + "@", sub[k])
open(os.path.join(ROOT, "README.md"), "w").write(open(os.path.join(ROOT, "tools/docgen/readme_head.md")).read() + rt)
print("README.md written")
