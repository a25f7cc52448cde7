"""Where the 16-bit vocoder's waveform error comes from, stage by stage (CPU only: the NumPy oracle with rounding switched on at the points where
the HIP path stores a 16-bit tensor: weights, every convolution's stored output, the residual stream, the running sum).
    python tools/vocoder_error_budget.py [frames=48]
Prints max / rms waveform error against the f32 oracle for: all bf16, all IEEE half, and half from stage s on (bf16 before)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import zvx_oracle as O
from zerovox_amd import config as zcfg, weights as zw


def rbf(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).view(np.float32)


def rh(x):
    return np.clip(x, -65504, 65504).astype(np.float16).astype(np.float32)


RND = {"f32": lambda x: x, "bf16": rbf, "f16": rh}


def generator(mel, hsd, h, dts):
    """dts[i]: storage dtype of stage i (0 = conv_pre output, 1..ns = the upsampling stages)."""
    r = RND[dts[0]]
    w = lambda n, q: q(O.fold_wn(hsd, n))
    x = r(O.conv1d(r(mel), w("conv_pre", r), hsd["conv_pre.bias"], padding=3))
    nk = len(h["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        rin, r = RND[dts[i]], RND[dts[i + 1]]
        x = r(O.conv_transpose1d(rin(O.leaky_relu(x, 0.1)), w(f"ups.{i}", rin), hsd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2))
        xs = None
        for j, (rk, rd) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            y = x
            p = f"resblocks.{i * nk + j}"
            for t, d in enumerate(rd):
                xt = r(O.leaky_relu(O.conv1d(r(O.leaky_relu(y, 0.1)), w(f"{p}.convs1.{t}", r), hsd[f"{p}.convs1.{t}.bias"], dilation=d, padding=O.get_padding(rk, d)), 0.1))
                xt = O.conv1d(xt, w(f"{p}.convs2.{t}", r), hsd[f"{p}.convs2.{t}.bias"], dilation=1, padding=O.get_padding(rk, 1))
                y = xt + y
                if t + 1 < len(rd): y = O.leaky_relu(r(O.leaky_relu(y, 0.1)), 10.0) * 1.0 if False else r(y)      # stored (activated domain: same relative rounding)
            xs = r(y) if xs is None else r(xs + y)
        x = r(xs / np.float32(nk))
    x = O.leaky_relu(x, 0.01)
    return np.tanh(O.conv1d(x, O.fold_wn(hsd, "conv_post"), hsd["conv_post.bias"], padding=3))[0]


if __name__ == "__main__":
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
    mel = np.random.default_rng(7).standard_normal((80, P)).astype(np.float32)
    ref = O.hifigan_generator(mel, hsd, h)
    ns = len(h["upsample_rates"])
    def show(name, dts):
        e = generator(mel, hsd, h, dts) - ref
        print(f"{name:40s} max {np.abs(e).max():.3e}  rms {np.sqrt((e ** 2).mean()):.3e}", flush=True)
    show("f32 everywhere (sanity)", ["f32"] * (ns + 1))
    show("bf16 everywhere", ["bf16"] * (ns + 1))
    show("half everywhere", ["f16"] * (ns + 1))
    for s in range(1, ns + 1):
        show(f"bf16 up to stage {s - 1}, half from stage {s}", ["bf16"] * s + ["f16"] * (ns + 1 - s))
    for s in range(1, ns + 1):
        show(f"half up to stage {s - 1}, bf16 from stage {s}", ["f16"] * s + ["bf16"] * (ns + 1 - s))
    for s in range(0, ns + 1):
        show(f"only stage {s} in bf16, rest f32", ["f32"] * s + ["bf16"] + ["f32"] * (ns - s))
    # round 6: the per-stage choice of zvx_set_int("voc_f16_stages", mask) -- bit k: domain k in half, else bf16
    for mask in (0b11111, 0b11011, 0b11001, 0b10001, 0b11101, 0b11010, 0b00000):
        show(f"voc_f16_stages = {mask:#07b} (bit k = domain k in half)", ["f16" if (mask >> k) & 1 else "bf16" for k in range(ns + 1)])
