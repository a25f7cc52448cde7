"""Build libzvx.so (HIP kernels + C-ABI) in-tree for gfx950 with hipcc.  `python -m zerovox_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libzvx.so")
SOURCES = ["gemm.hip", "resstream.hip", "pairstream.hip", "narrowstage.hip", "attention.hip", "ops.hip", "zvx.hip"]
# translation units: (object stem, source, extra -D flags).  gemm.hip is compiled as two units side by side (its fused ResBlock-pair kernels are
# a third of its instantiations): a clean build takes the time of the larger half
UNITS = [("gemm", "gemm.hip", ["-DZVX_GEMM_PART=1"]), ("gemm_resfuse", "gemm.hip", ["-DZVX_GEMM_PART=2"])] + \
        [(s_[:-4], s_, []) for s_ in SOURCES[1:]]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm"]
# gemm.hip: no NaN is ever a legitimate operand of its epilogue min/max (leaky-relu and its inverse); without this flag every
# fminf/fmaxf input coming from a bit operation (bf16 unpack) gets a canonicalising `v_max x, x, x` in front (IEEE mode)
EXTRA_FLAGS = {"gemm.hip": ["-fno-honor-nans"], "pairstream.hip": ["-fno-honor-nans"], "narrowstage.hip": ["-fno-honor-nans"], "resstream.hip": ["-fno-honor-nans"] + (["-DRS_PROFILE"] if os.environ.get("ZVX_RS_PROFILE_BUILD") else [])}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _parse_resources(text):
    """hipcc -Rpass-analysis=kernel-resource-usage remarks -> {kernel: {vgprs, agprs, sgprs, scratch, vgpr_spill, sgpr_spill, occupancy}}."""
    import re
    res, cur = {}, None
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
            "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}
    for line in text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    return res


def resources():
    """Register / scratch usage of every kernel as reported by hipcc when the objects were compiled (csrc/*.resources.json)."""
    import json
    out = {}
    for stem, _src, _defs in UNITS:
        f = os.path.join(CSRC, stem + ".resources.json")
        if os.path.exists(f):
            out.update(json.load(open(f)))
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "zvx_kernels.h"), os.path.join(CSRC, "mfma_util.h"), os.path.join(os.path.dirname(HERE), "include", "zvx.h")]
    def compile_one(unit):
        stem, src, defs = unit
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, stem + ".o")
        rj = os.path.join(CSRC, stem + ".resources.json")
        if force or _stale(o, [s] + headers) or not os.path.exists(rj):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + defs + ["-Rpass-analysis=kernel-resource-usage", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            p = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            # the remarks (one block per kernel, each followed by a two-line source excerpt) are parsed below, everything else is shown
            lines, keep, skip = p.stderr.splitlines(), [], 0
            for l in lines:
                if "[-Rpass-analysis=kernel-resource-usage]" in l:
                    skip = 2
                elif skip and (l.lstrip().startswith("|") or (l.strip()[:1].isdigit() and " | " in l)):
                    skip -= 1
                elif "remarks generated" not in l:
                    skip = 0
                    keep.append(l)
            if keep:
                print("\n".join(keep), file=sys.stderr)
            if p.returncode:
                raise subprocess.CalledProcessError(p.returncode, cmd)
            import json
            res = _parse_resources(p.stderr)
            json.dump(res, open(rj, "w"), indent=0, sort_keys=True)
            spilled = {k: v for k, v in res.items() if v.get("vgpr_spill")}
            if spilled and verbose:
                print(f"note: {len(spilled)} kernels of {stem} spill VGPRs: " + ", ".join(f"{k[:60]}({v['vgpr_spill']})" for k, v in list(spilled.items())[:6]), flush=True)
        return o

    # the translation units are independent: compile them side by side (gemm.hip is two units of ~1.5-2 minutes each; sequentially a build from
    # scratch is ~6).  ZVX_BUILD_JOBS caps the number of concurrent hipcc processes (each needs 1-2 GB).
    from concurrent.futures import ThreadPoolExecutor
    jobs = max(1, int(os.environ.get("ZVX_BUILD_JOBS", "5")))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(compile_one, UNITS))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
