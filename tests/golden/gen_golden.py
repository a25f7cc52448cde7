#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Runs ONLY in the build container (needs /root/reference + torch CPU).  The reference is imported
with the two in-process stubs of tools/ref_import.py, its modules are filled with the build's seeded
synthetic weights (zerovox_amd.weights -- the reference ships no weights offline), and inputs /
outputs of `ZeroVox.inference_ex` (model.py:308-347), `ResNetSE34V2.forward` and selected sub-modules
are stored as small .npz files.  A fixture is DATA: seeded inputs + the reference's outputs; the
weights are regenerated from (config name, seed) on the consuming side.

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.npz + MANIFEST.json
"""
import hashlib
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
warnings.filterwarnings("ignore")

import ref_import  # noqa: E402

ref_import.install()
import torch  # noqa: E402
from zerovox.tts.hifigan import Generator, ResBlock1, ResBlock2  # noqa: E402
from zerovox.tts.model import AttrDict, ZeroVox  # noqa: E402
from zerovox.tts.symbols import Symbols  # noqa: E402

from zerovox_amd import config as zcfg  # noqa: E402
from zerovox_amd import weights as zw  # noqa: E402

torch.set_num_threads(8)
SEED = 0
MANIFEST = {}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    MANIFEST[name] = {"sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(),
                      "bytes": os.path.getsize(path), "keys": sorted(arrays)}
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.0f} KiB)")


def t2n(t):
    return t.detach().cpu().numpy()


_MODELS = {}


def ref_model(kind):
    if kind not in _MODELS:
        cfg = zcfg.medium_modelcfg(kind)
        zv = ZeroVox(symbols=Symbols(zcfg.PHONES, zcfg.PUNCTS), meldec_model=None, **zcfg.zerovox_kwargs(cfg))
        sd = zw.tts_state_dict(cfg, SEED)
        zv.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
        zv.eval()
        _MODELS[kind] = zv
    return _MODELS[kind]


_GENS = {}


def ref_generator(name):
    if name not in _GENS:
        h = zcfg.hifigan_config(name)
        g = Generator(AttrDict(h))
        hsd = zw.hifigan_state_dict(h, SEED)
        g.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in hsd.items()}, strict=True)
        g.eval()
        g.remove_weight_norm()          # model.py:113-115
        _GENS[name] = g
    return _GENS[name]


def synth_inputs(T, utt, dur_mode):
    """BASELINE.md §3 generators: ids ~ U, default_rng(1234+utt); spk = L2-normalised N(0,1)^528."""
    r = np.random.default_rng(1234 + utt)
    phoneme = r.integers(0, 28, size=T).astype(np.int32)
    puncts = r.integers(0, 10, size=T).astype(np.int32)
    spk = r.standard_normal(528)
    spk = (spk / np.linalg.norm(spk)).astype(np.float32)
    if dur_mode == "const7":
        dur = np.full(T, 7, dtype=np.int32)
    elif dur_mode == "uniform":
        dur = r.integers(3, 11, size=T).astype(np.int32)
    elif dur_mode == "ragged":          # includes zeros (dropped phonemes) and a long one
        dur = r.integers(0, 6, size=T).astype(np.int32)
        dur[T // 2] = 17
    else:
        dur = None
    return phoneme, puncts, spk, dur


def e2e_case(name, kind, voc, T, utt, dur_mode, pad_to):
    print(f"[e2e] {name}")
    zv = ref_model(kind)
    zv._meldec = ref_generator(voc)
    zv._min_mel_len = pad_to            # the stateful value a fresh model holds is 689 (model.py:254)
    phoneme, puncts, spk, dur = synth_inputs(T, utt, dur_mode)
    cap = {}
    hooks = [
        zv._phoneme_encoder._encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("enc", t2n(o)[0])),
        zv._phoneme_encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("pe", o)),
    ]
    with torch.no_grad():
        x = {"phoneme": torch.from_numpy(phoneme[None]).int(), "puncts": torch.from_numpy(puncts[None]).int(),
             "duration": torch.from_numpy(dur[None]).int() if dur is not None else None}
        wav, mel_len, logd, mel = zv.inference_ex(x, style_embed=torch.from_numpy(spk).reshape(1, 1, -1),
                                                  force_duration=dur is not None)
    for h in hooks:
        h.remove()
    pe = cap["pe"]
    save(name, phoneme=phoneme, puncts=puncts, spk=spk,
         duration=dur if dur is not None else np.zeros(0, np.int32),
         forced=np.array(dur is not None), pad_to=np.array(pad_to), utt=np.array(utt),
         decoder_kind=np.array(kind), vocoder=np.array(voc), seed=np.array(SEED),
         wav=t2n(wav).astype(np.float32), mel=t2n(mel).astype(np.float32), mel_len=np.array(mel_len),
         log_duration=t2n(logd)[0], pitch=t2n(pe["pitch"])[0], energy=t2n(pe["energy"])[0],
         encoder_raw=cap["enc"], features=t2n(pe["features"])[0],
         min_mel_len_after=np.array(zv._min_mel_len))


def spk_case(name, Tr, seed):
    print(f"[spk] {name}")
    zv = ref_model("styletts")
    mel = np.random.default_rng(seed).standard_normal((Tr, 80)).astype(np.float32)
    with torch.no_grad():
        e = zv._spkemb(torch.from_numpy(mel[None]))
    save(name, ref_mel=mel, embed=t2n(e)[0, 0], seed=np.array(SEED))


def spk_sap_case(name, Tr, seed):
    """encoder_type 'SAP' (ResNetSE34V2.py:135-143, 199-200): a second reference model whose speaker encoder pools with the
    attention-weighted mean only (fc over 2560 inputs); everything else as the medium config."""
    import copy
    print(f"[spk-sap] {name}")
    cfg = copy.deepcopy(zcfg.medium_modelcfg("styletts"))
    cfg["model"]["resnet"]["encoder_type"] = "SAP"
    zv = ZeroVox(symbols=Symbols(zcfg.PHONES, zcfg.PUNCTS), meldec_model=None, **zcfg.zerovox_kwargs(cfg))
    sd = zw.tts_state_dict(cfg, SEED)
    zv.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    zv.eval()
    mel = np.random.default_rng(seed).standard_normal((Tr, 80)).astype(np.float32)
    with torch.no_grad():
        e = zv._spkemb(torch.from_numpy(mel[None]))
    save(name, ref_mel=mel, embed=t2n(e)[0, 0], seed=np.array(SEED), encoder_type=np.array("SAP"))


def v3_cases():
    """HiFi-GAN V3 at its published width (jik876 config_v3: ResBlock2, hifigan.py:65-86, kernels 3 / 5 / 7, dilations up to 12):
    the last ResBlock2 of stage 1 (C = 128, k = 7, dilations [3, 12]) on its own, and the whole generator on a short mel."""
    print("[v3]")
    r = np.random.default_rng(123)
    g = ref_generator("v3")
    h = zcfg.hifigan_config("v3")
    out = {}
    with torch.no_grad():
        C1 = h["upsample_initial_channel"] // 2
        xr = r.standard_normal((1, C1, 96)).astype(np.float32)
        out["rb_x"] = xr[0]
        out["rb2_y"] = t2n(g.resblocks[2](torch.from_numpy(xr)))[0]          # k = 7, dilations [3, 12]
        out["rb0_y"] = t2n(g.resblocks[0](torch.from_numpy(xr)))[0]          # k = 3, dilations [1, 2]
        mel = r.standard_normal((80, 20)).astype(np.float32)
        out["mel"] = mel
        out["wav"] = t2n(g(torch.from_numpy(mel)))[0]
    save("blocks_hifigan_v3", **out)


def block_cases():
    print("[blocks]")
    r = np.random.default_rng(99)
    zs, zf = ref_model("styletts"), ref_model("fastspeech2")
    spk = r.standard_normal(528)
    spk = (spk / np.linalg.norm(spk)).astype(np.float32)
    spk_t = torch.from_numpy(spk).reshape(1, 1, -1)
    out = {"spk": spk}
    with torch.no_grad():
        # FFTBlock with LayerNorm (encoder layer 0) and with SCLN (decoder layer 0)   fs2.py:221-230
        x = r.standard_normal((1, 24, 528)).astype(np.float32)
        mask = torch.zeros(1, 24, dtype=torch.bool)
        sam = mask.unsqueeze(1).expand(-1, 24, -1)
        out["fft_x"] = x[0]
        out["fft_ln_y"] = t2n(zs._phoneme_encoder._encoder.layer_stack[0](torch.from_numpy(x), None, mask, sam)[0])[0]
        out["fft_scln_y"] = t2n(zf._mel_decoder.layer_stack[0](torch.from_numpy(x), spk_t, mask, sam)[0])[0]
        # SCLN alone   fs2.py:76-90
        out["scln_y"] = t2n(zf._mel_decoder.layer_stack[0].slf_attn.layer_norm(torch.from_numpy(x), spk_t))[0]
        # VariancePredictor (duration)   fs2.py:555-563
        out["vp_y"] = t2n(zs._phoneme_encoder._variance_adaptor.duration_predictor(torch.from_numpy(x), None))[0]
        # LengthRegulator   fs2.py:432-459
        dur = np.array([[2, 0, 3, 1] + [1] * 20], dtype=np.int32)
        y, ml = zs._phoneme_encoder._variance_adaptor.length_regulator(torch.from_numpy(x), torch.from_numpy(dur), None)
        out["lr_dur"], out["lr_y"], out["lr_len"] = dur[0], t2n(y)[0], t2n(ml)
        # StyleTTS blocks   styletts.py:11-69, 95-139
        xc = r.standard_normal((1, 528, 20)).astype(np.float32)
        out["sty_x"] = xc[0]
        out["resblk1d_y"] = t2n(zs._mel_decoder.encode[0](torch.from_numpy(xc)))[0]
        xc2 = r.standard_normal((1, 1120, 20)).astype(np.float32)
        out["adain_x"] = xc2[0]
        out["adain_y"] = t2n(zs._mel_decoder.decode[2](torch.from_numpy(xc2), torch.from_numpy(spk)[None]))[0]
        # whole decoders on random features
        feats = r.standard_normal((1, 20, 528)).astype(np.float32)
        out["dec_x"] = feats[0]
        out["dec_styletts_y"] = t2n(zs._mel_decoder(torch.from_numpy(feats), torch.zeros(1, 20, dtype=torch.bool), spk_t)[0])[0]
        out["dec_fs2_y"] = t2n(zf._mel_decoder(torch.from_numpy(feats), torch.zeros(1, 20, dtype=torch.bool), spk_t)[0])[0]
        # SEBasicBlock with stride-2 + downsample (layer2.0) and a plain one (layer1.1)  ResNetSE34V2.py:83-99
        xm = r.standard_normal((1, 32, 12, 10)).astype(np.float32)
        out["se_x"] = xm[0]
        out["se_l2_y"] = t2n(zs._spkemb.layer2[0](torch.from_numpy(xm)))[0]
        out["se_l1_y"] = t2n(zs._spkemb.layer1[1](torch.from_numpy(xm)))[0]
    save("blocks_tts", **out)

    out = {}
    with torch.no_grad():
        for voc in ("tiny", "tiny2"):
            g = ref_generator(voc)
            h = zcfg.hifigan_config(voc)
            mel = r.standard_normal((80, 12)).astype(np.float32)
            out[f"{voc}_mel"] = mel
            out[f"{voc}_wav"] = t2n(g(torch.from_numpy(mel)))[0]
            C1 = h["upsample_initial_channel"] // 2
            xr = r.standard_normal((1, C1, 40)).astype(np.float32)
            out[f"{voc}_rb_x"] = xr[0]
            out[f"{voc}_rb0_y"] = t2n(g.resblocks[0](torch.from_numpy(xr)))[0]
            out[f"{voc}_rb1_y"] = t2n(g.resblocks[1](torch.from_numpy(xr)))[0]
            xu = r.standard_normal((1, h["upsample_initial_channel"], 9)).astype(np.float32)
            out[f"{voc}_up_x"] = xu[0]
            out[f"{voc}_up0_y"] = t2n(g.ups[0](torch.from_numpy(xu)))[0]
            xu2 = r.standard_normal((1, h["upsample_initial_channel"] // 4, 9)).astype(np.float32)
            out[f"{voc}_up2_x"] = xu2[0]
            out[f"{voc}_up2_y"] = t2n(g.ups[2](torch.from_numpy(xu2)))[0]
    save("blocks_hifigan", **out)


def text_cases():
    """transcript2phonemids needs no reference import (synthesize.py pulls librosa/uroman); its
    documented example (synthesize.py:201-203) is encoded in tests/test_host_api.py instead."""


def main():
    # (every fixture below is overwritten in place by save(); nothing else in this directory is touched -- refckpt_expected.npz and
    # refckpt/ belong to gen_ref_checkpoint.py, which keeps its own REFCKPT_MANIFEST.json)
    e2e_case("e2e_styletts_tiny_T8", "styletts", "tiny", 8, 0, "uniform", 32)
    e2e_case("e2e_fs2_tiny_T8", "fastspeech2", "tiny", 8, 1, "uniform", 32)
    e2e_case("e2e_styletts_tiny_T16_pred", "styletts", "tiny", 16, 2, None, 16)
    e2e_case("e2e_fs2_tiny_T16_pred", "fastspeech2", "tiny", 16, 3, None, 16)
    e2e_case("e2e_fs2_tiny2_T12_ragged", "fastspeech2", "tiny2", 12, 4, "ragged", 48)
    e2e_case("e2e_styletts_tiny2_T12_ragged", "styletts", "tiny2", 12, 5, "ragged", 8)
    e2e_case("e2e_styletts_v1_T64", "styletts", "v1", 64, 0, "const7", 689)      # BASELINE config #1
    e2e_case("e2e_fs2_v2_T24", "fastspeech2", "v2", 24, 6, "uniform", 689)
    spk_case("spkemb_T96", 96, 7)
    spk_case("spkemb_T258", 258, 8)
    spk_sap_case("spkemb_sap_T96", 96, 9)
    block_cases()
    v3_cases()
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump({"generator": "tests/golden/gen_golden.py", "reference": "gooofy/zerovox @ 2025-04-18",
                   "torch": torch.__version__, "numpy": np.__version__, "weight_seed": SEED,
                   "fixtures": MANIFEST}, f, indent=1, sort_keys=True)
    print("total KiB:", sum(v["bytes"] for v in MANIFEST.values()) // 1024)


if __name__ == "__main__":
    main()
