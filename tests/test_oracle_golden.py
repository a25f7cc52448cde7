"""Pin the CPU oracle (oracle/zvx_oracle.py) against fixtures produced by the reference itself
(tests/golden/gen_golden.py).  fp32 tolerance 1e-4 relative-to-scale on floats (measured <= 3e-5);
durations / mel_len / bucket-driven lengths exact."""
import os

import numpy as np
import pytest

from oracle import zvx_oracle as O
from zerovox_amd import config as zcfg
from zerovox_amd import weights as zw

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
_cache = {}


def tts_sd(kind):
    if kind not in _cache:
        cfg = zcfg.medium_modelcfg(kind)
        _cache[kind] = (cfg, zw.tts_state_dict(cfg, 0))
    return _cache[kind]


def voc_sd(name):
    if name not in _cache:
        h = zcfg.hifigan_config(name)
        _cache[name] = (h, zw.hifigan_state_dict(h, 0))
    return _cache[name]


def close(a, b, tol=1e-4):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max()) / scale
    assert err <= tol, f"max err {err:.3e} > {tol}"
    return err


E2E = ["e2e_styletts_tiny_T8", "e2e_fs2_tiny_T8", "e2e_styletts_tiny_T16_pred", "e2e_fs2_tiny_T16_pred",
       "e2e_fs2_tiny2_T12_ragged", "e2e_styletts_tiny2_T12_ragged", "e2e_fs2_v2_T24", "e2e_styletts_v1_T64"]


@pytest.mark.parametrize("name", E2E)
def test_e2e_matches_reference(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg, sd = tts_sd(str(g["decoder_kind"]))
    h, hsd = voc_sd(str(g["vocoder"]))
    dur = g["duration"] if bool(g["forced"]) else None
    out = O.inference_ex(sd, hsd, cfg, h, g["phoneme"], g["puncts"], g["spk"], duration=dur,
                         pad_to=int(g["pad_to"]))
    assert out["mel_len"] == int(g["mel_len"])
    assert max(int(g["pad_to"]), out["mel_len"]) == int(g["min_mel_len_after"])   # model.py:331-335
    close(out["encoder_out"] - g["spk"][None], g["encoder_raw"])
    close(out["log_duration"], g["log_duration"])
    close(out["pitch"], g["pitch"])
    close(out["energy"], g["energy"])
    close(out["features"], g["features"])
    close(out["mel"], g["mel"], 2e-4)
    assert out["wav"].shape == g["wav"].shape == (out["mel_len"] * 256,)
    close(out["wav"], g["wav"], 2e-4)


@pytest.mark.parametrize("name", ["spkemb_T96", "spkemb_T258"])
def test_speaker_encoder_matches_reference(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg, sd = tts_sd("styletts")
    e = O.resnet_se34v2(g["ref_mel"], sd, cfg)
    assert abs(np.linalg.norm(e) - 1.0) < 1e-5
    close(e, g["embed"], 1e-5)


def test_tts_blocks_match_reference():
    g = np.load(os.path.join(GOLDEN, "blocks_tts.npz"))
    cfg_s, sd_s = tts_sd("styletts")
    cfg_f, sd_f = tts_sd("fastspeech2")
    spk, x = g["spk"], g["fft_x"]
    close(O.fft_block(x, sd_s, "_phoneme_encoder._encoder.layer_stack.0", 2), g["fft_ln_y"])
    close(O.fft_block(x, sd_f, "_mel_decoder.layer_stack.0", 2, spk), g["fft_scln_y"])
    close(O.scln(x, spk, sd_f["_mel_decoder.layer_stack.0.slf_attn.layer_norm.affine_layer.linear.weight"]), g["scln_y"])
    close(O.variance_predictor(x, sd_s, "_phoneme_encoder._variance_adaptor.duration_predictor"), g["vp_y"])
    y = O.length_regulate(x, g["lr_dur"])
    assert y.shape[0] == int(g["lr_len"][0]) and np.array_equal(y, g["lr_y"])     # pure copy: bit-exact
    close(O.resblk1d(g["sty_x"], sd_s, "_mel_decoder.encode.0"), g["resblk1d_y"])
    close(O.adain_resblk1d(g["adain_x"], spk, sd_s, "_mel_decoder.decode.2"), g["adain_y"])
    close(O.styletts_decoder(g["dec_x"], spk, sd_s), g["dec_styletts_y"])
    close(O.fs2_decoder(g["dec_x"], spk, sd_f, cfg_f), g["dec_fs2_y"])
    close(O.se_basic_block(g["se_x"], sd_s, "_spkemb.layer2.0", 2), g["se_l2_y"])
    close(O.se_basic_block(g["se_x"], sd_s, "_spkemb.layer1.1", 1), g["se_l1_y"])


@pytest.mark.parametrize("voc", ["tiny", "tiny2"])
def test_hifigan_blocks_match_reference(voc):
    g = np.load(os.path.join(GOLDEN, "blocks_hifigan.npz"))
    h, hsd = voc_sd(voc)
    close(O.hifigan_generator(g[f"{voc}_mel"], hsd, h), g[f"{voc}_wav"])
    rb = O.resblock1 if h["resblock"] == "1" else O.resblock2
    for j in (0, 1):
        close(rb(g[f"{voc}_rb_x"], hsd, f"resblocks.{j}", h["resblock_kernel_sizes"][j],
                 h["resblock_dilation_sizes"][j]), g[f"{voc}_rb{j}_y"])
    for i, key in ((0, "up"), (2, "up2")):
        u, k = h["upsample_rates"][i], h["upsample_kernel_sizes"][i]
        y = O.conv_transpose1d(g[f"{voc}_{key}_x"], O.fold_wn(hsd, f"ups.{i}"), hsd[f"ups.{i}.bias"], stride=u,
                               padding=(k - u) // 2)
        close(y, g[f"{voc}_{key if i else 'up0'}_y"] if i else g[f"{voc}_up0_y"])


def test_speaker_encoder_sap_pooling_matches_reference():
    """encoder_type 'SAP' (ResNetSE34V2.py:135-143, 199-200) against the reference built with that option."""
    import copy
    g = np.load(os.path.join(GOLDEN, "spkemb_sap_T96.npz"))
    cfg = copy.deepcopy(zcfg.medium_modelcfg("styletts"))
    cfg["model"]["resnet"]["encoder_type"] = str(g["encoder_type"])
    e = O.resnet_se34v2(g["ref_mel"], zw.tts_state_dict(cfg, 0), cfg)
    assert abs(np.linalg.norm(e) - 1.0) < 1e-5
    close(e, g["embed"], 1e-5)


def test_hifigan_v3_resblock2_at_published_width_matches_reference():
    """ResBlock2 (hifigan.py:65-86) at V3's width, incl. the k = 7 block with dilation 12, and the whole V3 generator."""
    g = np.load(os.path.join(GOLDEN, "blocks_hifigan_v3.npz"))
    h, hsd = voc_sd("v3")
    for j in (0, 2):
        close(O.resblock2(g["rb_x"], hsd, f"resblocks.{j}", h["resblock_kernel_sizes"][j], h["resblock_dilation_sizes"][j]), g[f"rb{j}_y"])
    close(O.hifigan_generator(g["mel"], hsd, h), g["wav"], 2e-4)


def test_manifest_hashes():
    import hashlib, json
    man = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))
    for name, meta in man["fixtures"].items():
        data = open(os.path.join(GOLDEN, name + ".npz"), "rb").read()
        assert hashlib.sha256(data).hexdigest() == meta["sha256"], name
    # the reference-written checkpoint, its configs and the reference's outputs on it (tests/golden/gen_ref_checkpoint.py)
    ref = json.load(open(os.path.join(GOLDEN, "REFCKPT_MANIFEST.json")))
    assert set(ref["files"]) >= {"refckpt_expected.npz", "refckpt/checkpoints/epoch=0-step=0.ckpt"}
    for name, meta in ref["files"].items():
        data = open(os.path.join(GOLDEN, name), "rb").read()
        assert hashlib.sha256(data).hexdigest() == meta["sha256"] and len(data) == meta["bytes"], name
    # every committed fixture is covered by one of the two manifests
    on_disk = {f for f in os.listdir(GOLDEN) if f.endswith(".npz")}
    assert on_disk == {n + ".npz" for n in man["fixtures"]} | {n for n in ref["files"] if n.endswith(".npz")}, on_disk


def test_onednn_backed_oracle_is_the_same_function():
    """bench.py's cpu_baseline times oracle/onednn_port.py (the oracle with its convolutions evaluated by torch / oneDNN) in a child
    process: the child checks itself against the NumPy oracle before timing and reports the difference."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "oracle", "onednn_port.py"), "5", "styletts", "v1", "4", "1"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-400:]
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j["check_max_abs_diff"] < 1e-4 and j["value"] > 0
