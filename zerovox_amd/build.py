"""Build libzvx.so (HIP kernels + C-ABI) in-tree for gfx950 with hipcc.  `python -m zerovox_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libzvx.so")
SOURCES = ["gemm.hip", "resstream.hip", "attention.hip", "ops.hip", "zvx.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm"]
# gemm.hip: no NaN is ever a legitimate operand of its epilogue min/max (leaky-relu and its inverse); without this flag every
# fminf/fmaxf input coming from a bit operation (bf16 unpack) gets a canonicalising `v_max x, x, x` in front (IEEE mode)
EXTRA_FLAGS = {"gemm.hip": ["-fno-honor-nans"], "resstream.hip": ["-fno-honor-nans"] + (["-DRS_PROFILE"] if os.environ.get("ZVX_RS_PROFILE_BUILD") else [])}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "zvx_kernels.h"), os.path.join(CSRC, "mfma_util.h"), os.path.join(os.path.dirname(HERE), "include", "zvx.h")]
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
