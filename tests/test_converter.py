"""Checkpoint converter round trip on a synthetic Lightning-style checkpoint (no real weights exist offline)."""
import json
import os
import sys

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from zerovox_amd import config as zcfg, weights as zw
from zerovox_amd.model import load_meldec_weights, load_tts_weights


def test_convert_roundtrip(tmp_path):
    import convert_checkpoint as cc
    cfg = zcfg.medium_modelcfg("styletts")
    h = zcfg.hifigan_config("tiny")
    sd, hsd = zw.tts_state_dict(cfg, 3), zw.hifigan_state_dict(h, 3)
    state = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    state.update({"_meldec." + k: torch.from_numpy(np.array(v)) for k, v in hsd.items()})    # vocoder baked in
    ck = tmp_path / "checkpoint.pkl"
    cc._install_symbols_stub()                                            # a real checkpoint pickles a zerovox.tts.symbols.Symbols instance
    sym = sys.modules["zerovox.tts.symbols"].Symbols(zcfg.PHONES, zcfg.PUNCTS)
    torch.save({"state_dict": state, "hyper_parameters": {"lr": 1e-4, "symbols": sym}}, ck)
    mc = tmp_path / "modelcfg.yaml"
    yaml.safe_dump(cfg, open(mc, "w"))
    n_tts, n_voc = cc.convert_tts(str(ck), str(mc), str(tmp_path / "out"))
    assert n_tts == len(sd) and n_voc == len(hsd)
    cfg2, sd2 = load_tts_weights(str(tmp_path / "out"))
    assert cfg2["model"]["decoder"]["kind"] == "styletts" and set(sd2) == set(sd)
    assert all(np.array_equal(sd2[k], sd[k]) for k in sd)
    gk = tmp_path / "generator.ckpt"
    torch.save({"generator": {k: torch.from_numpy(np.array(v)) for k, v in hsd.items()}}, gk)
    cj = tmp_path / "config.json"
    json.dump(h, open(cj, "w"))
    assert cc.convert_vocoder(str(gk), str(cj), str(tmp_path / "voc")) == len(hsd)
    h2, hsd2 = load_meldec_weights(str(tmp_path / "voc"))
    assert h2 == h and all(np.array_equal(hsd2[k], hsd[k]) for k in hsd)


def test_reference_model_directories_load_without_conversion(tmp_path):
    """synthesize.py:295-326 / model.py:90-111: a model directory as the reference downloads it (modelcfg.yaml +
    checkpoints/*.ckpt, the newest wins; config.json + generator.ckpt) is read directly."""
    import time
    cfg = zcfg.medium_modelcfg("fastspeech2")
    h = zcfg.hifigan_config("tiny2")
    sd, hsd = zw.tts_state_dict(cfg, 1), zw.hifigan_state_dict(h, 1)
    mdir = tmp_path / "tts_en"
    (mdir / "checkpoints").mkdir(parents=True)
    yaml.safe_dump(cfg, open(mdir / "modelcfg.yaml", "w"))
    old = {k: torch.zeros_like(torch.from_numpy(np.array(v))) for k, v in sd.items()}
    torch.save({"state_dict": old}, mdir / "checkpoints" / "epoch=1.ckpt")
    time.sleep(0.05)
    torch.save({"state_dict": {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}}, mdir / "checkpoints" / "epoch=2.ckpt")
    cfg2, sd2 = load_tts_weights(str(mdir))
    assert cfg2["model"]["decoder"]["kind"] == "fastspeech2" and all(np.array_equal(sd2[k], sd[k]) for k in sd)
    vdir = tmp_path / "hifigan"
    vdir.mkdir()
    json.dump(h, open(vdir / "config.json", "w"))
    torch.save({"generator": {k: torch.from_numpy(np.array(v)) for k, v in hsd.items()}}, vdir / "generator.ckpt")
    h2, hsd2 = load_meldec_weights(str(vdir), tts_modelpath=str(mdir))
    assert h2 == h and all(np.array_equal(hsd2[k], hsd[k]) for k in hsd)


def test_baked_in_vocoder_with_folded_weights_as_the_reference_stores_it(tmp_path):
    """model.py:115, 247 / edit_meldec_in_checkpoint.py:77-84: the reference attaches the generator AFTER remove_weight_norm(), so
    a Lightning checkpoint holds plain `_meldec.*.weight` tensors while generator.ckpt holds weight_g / weight_v pairs.  The
    baked-in copy must still be accepted (and win), and pack to the same device weights as the weight-normed original."""
    from zerovox_amd import pack
    cfg = zcfg.medium_modelcfg("styletts")
    h = zcfg.hifigan_config("tiny")
    sd, hsd = zw.tts_state_dict(cfg, 5), zw.hifigan_state_dict(h, 5)
    baked = zw.folded(hsd)
    assert any(k.endswith(".weight_g") for k in hsd) and not any(k.endswith(".weight_g") for k in baked)
    mdir = tmp_path / "tts_en"
    (mdir / "checkpoints").mkdir(parents=True)
    yaml.safe_dump(cfg, open(mdir / "modelcfg.yaml", "w"))
    state = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    state.update({"_meldec." + k: torch.from_numpy(np.array(v)) for k, v in baked.items()})
    torch.save({"state_dict": state}, mdir / "checkpoints" / "epoch=3.ckpt")
    vdir = tmp_path / "hifigan"
    vdir.mkdir()
    json.dump(h, open(vdir / "config.json", "w"))
    torch.save({"generator": {k: torch.from_numpy(np.array(v)) for k, v in hsd.items()}}, vdir / "generator.ckpt")
    h2, hsd2 = load_meldec_weights(str(vdir), tts_modelpath=str(mdir))
    assert h2 == h and set(hsd2) == set(baked) and all(np.array_equal(hsd2[k], baked[k]) for k in baked)
    m1, b1 = pack.pack_model(cfg, sd, h, hsd, "f32")
    m2, b2 = pack.pack_model(cfg, sd, h, hsd2, "f32")
    assert m1 == m2 and np.allclose(np.frombuffer(b1, np.float32), np.frombuffer(b2, np.float32), rtol=0, atol=1e-6)
    # a baked-in vocoder of another architecture is still refused
    other = zw.folded(zw.hifigan_state_dict(zcfg.hifigan_config("tiny2"), 5))
    state = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    state.update({"_meldec." + k: torch.from_numpy(np.array(v)) for k, v in other.items()})
    torch.save({"state_dict": state}, mdir / "checkpoints" / "epoch=4.ckpt")
    import pytest
    with pytest.raises(ValueError):
        load_meldec_weights(str(vdir), tts_modelpath=str(mdir))


def test_reference_written_checkpoint_reads_back_as_the_seeded_weights(tmp_path):
    """tests/golden/refckpt/ was written by the IMPORTED REFERENCE (tests/golden/gen_ref_checkpoint.py: ZeroVox(**kwargs) ->
    load_state_dict(strict=True) -> torch.save of its own state_dict() + hyper_parameters with the real Symbols object, a generator
    baked in after remove_weight_norm()).  Reading it -- directly and through tools/convert_checkpoint.py -- must give back the seeded
    weights it was filled with, bit for bit; the baked-in vocoder arrives with plain (folded) weights."""
    import convert_checkpoint as cc
    from zerovox_amd.convert import read_tts_checkpoint
    d = os.path.join(ROOT, "tests", "golden", "refckpt")
    cfg = zcfg.reduced_modelcfg("styletts")
    assert yaml.safe_load(open(os.path.join(d, "modelcfg.yaml"))) == cfg
    sd = zw.tts_state_dict(cfg, 11)
    tts, voc = read_tts_checkpoint(d)
    assert set(tts) == set(sd) and all(np.array_equal(tts[k], sd[k]) for k in sd)
    h = zcfg.hifigan_config("tiny3")
    folded = zw.folded(zw.hifigan_state_dict(h, 11))
    assert set(voc) == set(folded) and not any(k.endswith("weight_g") for k in voc)
    assert all(np.allclose(voc[k], folded[k], rtol=1e-6, atol=1e-7) for k in folded)        # torch's remove_weight_norm vs the NumPy fold
    ck = os.path.join(d, "checkpoints", os.listdir(os.path.join(d, "checkpoints"))[0])
    assert cc.convert_tts(ck, os.path.join(d, "modelcfg.yaml"), str(tmp_path / "out")) == (len(sd), len(voc))
    cfg2, sd2 = load_tts_weights(str(tmp_path / "out"))
    assert cfg2 == cfg and all(np.array_equal(sd2[k], sd[k]) for k in sd)
    h2, hsd2 = load_meldec_weights("synthetic:tiny3:5", tts_modelpath=str(tmp_path / "out"))    # the baked-in generator wins over the external one
    assert h2 == h and all(np.array_equal(hsd2[k], voc[k]) for k in voc)
