"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (bench.py's `cpu_baseline` leg runs this file in a child process; nothing else does).

The NumPy oracle (`oracle/zvx_oracle.py`) with its three convolution primitives and its leaky-ReLU evaluated by torch / oneDNN on the
host cores -- the library the reference itself runs on a CPU (zerovox/tts/hifigan.py, styletts.py, fs2.py build their layers from
torch.nn.Conv1d / ConvTranspose1d / Conv2d and F.leaky_relu) -- and with HiFi-GAN's weight norm folded once before the clock starts, as
the reference's `remove_weight_norm()` does at load (round 4: the NumPy `np.where` activation and the per-call fold were 60 % of the
port's time and are not what a CPU has to pay).  Everything else (normalisations, attention, length regulation, control flow) stays
the NumPy restatement.  This is the CPU baseline that is comparable to the reference's own CPU path; the plain NumPy oracle's im2col + sgemm
convolutions are 10-30x slower than oneDNN's and say little about what a CPU can do.

The child first checks the patched oracle against the unpatched one on a small case (same weights), then times the bounded sample,
and prints one JSON object.   usage: python oracle/onednn_port.py <config 2|4|5> <decoder> <vocoder> <threads> <units> [T]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    config, decoder, vocoder, threads, units = int(sys.argv[1]), sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    T = int(sys.argv[6]) if len(sys.argv) > 6 else 128
    import torch
    import torch.nn.functional as F
    torch.set_num_threads(threads)
    from oracle import zvx_oracle as O
    from zerovox_amd import config as zcfg, synthetic, weights as zw

    def tn(a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def conv1d(x, w, b=None, dilation=1, padding=0, stride=1):
        with torch.no_grad():
            return F.conv1d(tn(x)[None], tn(w), None if b is None else tn(b), stride=stride, padding=padding, dilation=dilation)[0].numpy()

    def conv_transpose1d(x, w, b=None, stride=1, padding=0):
        with torch.no_grad():
            return F.conv_transpose1d(tn(x)[None], tn(w), None if b is None else tn(b), stride=stride, padding=padding)[0].numpy()

    def conv2d(x, w, b=None, stride=1, padding=0):
        with torch.no_grad():
            return F.conv2d(tn(x)[None], tn(w), None if b is None else tn(b), stride=stride, padding=padding)[0].numpy()

    def leaky_relu(x, slope):                       # F.leaky_relu, as every activation of the reference (multi-threaded; np.where is one thread)
        with torch.no_grad():
            return F.leaky_relu(tn(x), float(slope)).numpy()

    cfg = zcfg.medium_modelcfg(decoder)
    sd = zw.tts_state_dict(cfg, 0)
    hcfg = zcfg.hifigan_config(vocoder)
    hsd = zw.hifigan_state_dict(hcfg, 0)
    plain = (O.conv1d, O.conv_transpose1d, O.conv2d, O.leaky_relu)
    # HiFi-GAN: the reference folds weight norm ONCE at load (Generator.remove_weight_norm, hifigan.py:132-139, called by get_meldec,
    # model.py:86-118); the oracle's fold_wn returns a stored ".weight" as it is.  (The StyleTTS convolutions keep their weight norm in
    # the reference too -- styletts.py:25-34 -- and are folded per call here as there.)
    hsd_folded = dict(hsd)
    for k in [k for k in hsd if k.endswith(".weight_v")]:
        pre = k[: -len(".weight_v")]
        hsd_folded[pre + ".weight"] = O.fold_wn(hsd, pre)
        del hsd_folded[pre + ".weight_v"], hsd_folded[pre + ".weight_g"]

    def patched(on):
        O.conv1d, O.conv_transpose1d, O.conv2d, O.leaky_relu = (conv1d, conv_transpose1d, conv2d, leaky_relu) if on else plain

    # equivalence on a small case first: the patched oracle is the same function
    if config == 5:
        m = np.random.default_rng(3).standard_normal((64, 80)).astype(np.float32)
        a = O.resnet_se34v2(m, sd, cfg); patched(True); b = O.resnet_se34v2(m, sd, cfg)
        diff = float(np.abs(np.asarray(a) - np.asarray(b)).max())
    elif config == 4:
        m = np.random.default_rng(3).standard_normal((80, 24)).astype(np.float32)
        a = O.hifigan_generator(m, hsd, hcfg); patched(True); b = O.hifigan_generator(m, hsd_folded, hcfg)
        diff = float(np.abs(a - b).max())
    else:
        ph, pu, spk, dur = synthetic.utterance(6, 0, "const7")
        a = O.inference_ex(sd, hsd, cfg, hcfg, ph, pu, spk, duration=dur, pad_to=8)["wav"]
        patched(True)
        b = O.inference_ex(sd, hsd_folded, cfg, hcfg, ph, pu, spk, duration=dur, pad_to=8)["wav"]
        diff = float(np.abs(a - b).max())
    if not diff < 1e-3:
        raise SystemExit(f"oneDNN-backed oracle differs from the NumPy oracle by {diff}")

    t0 = time.time()
    n = 0
    if config == 2:
        for u in range(units):
            ph, pu, spk, dur = synthetic.utterance(T, u, "const7")
            n += len(O.inference_ex(sd, hsd_folded, cfg, hcfg, ph, pu, spk, duration=dur, pad_to=896)["wav"])
        what = f"{units} utterances of the workload ({T} phonemes -> 896 frames -> 229376 samples each)"
    elif config == 4:
        for u in range(units):
            mel = np.random.default_rng(7 + u).standard_normal((80, 1024)).astype(np.float32)
            n += len(O.hifigan_generator(mel, hsd_folded, hcfg))
        what = f"{units} utterances of the workload (1024-frame N(0,1) mels -> 262144 samples each)"
    else:
        mels = np.random.default_rng(8).standard_normal((units, 258, 80)).astype(np.float32)
        for m in mels:
            O.resnet_se34v2(m, sd, cfg)
        n = units
        what = f"{units} clips of the workload (258-frame mels)"
    dt = time.time() - t0
    print(json.dumps({"value": n / dt, "seconds": dt, "what": what, "check_max_abs_diff": diff, "threads": threads,
                      "torch": torch.__version__}))


if __name__ == "__main__":
    main()
