#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of `bench.py`.
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section)
prescribes for wide coalesced reads on gfx950; WRITE_SIZE is taken as reported (uncalibrated)."""
import json, re, sqlite3, sys

NAMES = {  # rocprof kernel name pattern -> bench variant name
    r"convslab_kernel<256, 128,": "convslab_bf16_256x128", r"convslab_kernel<128, 256,": "convslab_bf16_128x256",
    r"convslab_kernel<128, 128,": "convslab_bf16_128x128", r"flash_attn_kernel<": "flash_attn_bf16",
    r"convslab_kernel<256, 64,": "convslab_bf16_256x64", r"convslab_kernel<256, 32,": "convslab_bf16_256x32",
    r"resfuse_kernel<64,": "resfuse_bf16_c64", r"resfuse_kernel<32,": "resfuse_bf16_c32",
    r"resfuse_persist_kernel<64,": "resfuse_bf16_c64", r"resfuse_persist_kernel<32,": "resfuse_bf16_c32",
    r"resfuse_persist_kernel<128,": "resfuse_bf16_c128", r"gemm_kernel<0, 64, 64": "gemm_f32_64x64", r"gemm_kernel<1, 64, 64": "gemm_bf16_64x64",
    r"convreg_kernel<64,": "convreg_bf16_c64", r"convreg_kernel<32,": "convreg_bf16_c32",
    r"conv2d_persist_kernel<64,": "conv2d_persist_c64", r"conv2d_persist_kernel<32,": "conv2d_persist_c32",
    r"gemm_kernel<1, 128, 128": "gemm_bf16_128x128", r"gemm_kernel<0, 128, 128": "gemm_f32_128x128",
    r"resstream_kernel<32,": "resstream_bf16_c32", r"resstream_kernel<64,": "resstream_bf16_c64",
    r"pairstream128_kernel<": "pairstream_bf16_c128", r"gemm_kernel<1, 256, 64": "gemm_bf16_256x64", r"gemm_kernel<1, 256, 32": "gemm_bf16_256x32",
}


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for kn, n, sm in cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name", (counter,)):
        for pat, name in NAMES.items():
            if pat in kn:
                a = out.setdefault(name, [0, 0.0]); a[0] += n; a[1] += sm
    return out


def clocks(db):
    """Effective shader clock and MFMA-pipe duty per variant from the SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE pass:
    GRBM_GUI_ACTIVE is summed over the 8 XCDs (clock = sum / 8 / kernel time), MFMA busy cycles over the 1024 SIMDs."""
    cur = sqlite3.connect(db).cursor()
    busy = per_kernel(db, "SQ_VALU_MFMA_BUSY_CYCLES")
    gui = {}
    for kn, sm, ns in cur.execute("select kernel_name, sum(value), sum(duration) from counters_collection where counter_name='GRBM_GUI_ACTIVE' group by kernel_name"):
        for pat, name in NAMES.items():
            if pat in kn:
                a = gui.setdefault(name, [0.0, 0.0]); a[0] += sm; a[1] += ns
    out = {}
    for name, (g, ns) in gui.items():
        if g and ns:
            out[name] = {"eff_clock_GHz": g / 8.0 / ns, "mfma_busy_frac": busy.get(name, [0, 0.0])[1] / 1024.0 / (g / 8.0)}
    return out


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
res = {}
for name in fetch:
    n, f = fetch[name]; w = write.get(name, [n, 0.0])[1]
    res[name] = {"launches": n, "fetch_KiB_per_launch_raw": f / n, "write_KiB_per_launch": w / n,
                 "hbm_bytes_per_launch": (2 * f + w) * 1024 / n}
if len(sys.argv) > 5:
    for name, c in clocks(sys.argv[5]).items():
        if name in res:
            res[name].update(c)
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import argparse
import bench
# argv[4]: the bench.py options of the profiled command (e.g. "--config 4"); the key bench.py compares before quoting
ap = argparse.ArgumentParser()
for o, d in (("--config", 2), ("--batch", None), ("--phonemes", 128)):
    ap.add_argument(o, type=int, default=d)
for o, d in (("--decoder", "styletts"), ("--vocoder", "v1"), ("--precision", "bf16")):
    ap.add_argument(o, default=d)
ap.add_argument("--exact-encoder", action="store_true")
bargs, _ = ap.parse_known_args((sys.argv[4] if len(sys.argv) > 4 else "").split())
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on `python bench.py --steps 3 --warmup 1 --no-cpu-baseline " + (sys.argv[4] if len(sys.argv) > 4 else "") + "`",
           "src_sha16": bench.src_sha16(), "workload_key": bench.workload_key(bargs), **res}, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res, indent=1))
