#!/usr/bin/env python3
"""Checkpoint converter: reference checkpoints -> the torch-free files zerovox_amd loads.

    tools/convert_checkpoint.py tts  <model_dir | checkpoint.(ckpt|pkl)> <modelcfg.yaml> <out_dir>
    tools/convert_checkpoint.py voc  <generator.ckpt> <config.json> <out_dir>

* TTS: a Lightning checkpoint (`state_dict` / `hyper_parameters`, layouts in utils/dump_pkl.py:8-30) is read with
  torch.load; `_meldec.*` keys (a vocoder baked into the checkpoint, utils/edit_meldec_in_checkpoint.py:77-90) are
  split off into `<out_dir>/generator.npz` -- zerovox_amd.model.load_meldec_weights PREFERS that file over the external
  vocoder's weights, as `ZeroVox.load_from_checkpoint(strict=False)` does (the external model then only supplies
  config.json); every other tensor is written to `<out_dir>/weights.npz` under its reference key; modelcfg.yaml is copied.
  The pickled `hyper_parameters` reference `zerovox.tts.symbols.Symbols`; a stand-in class is registered so that
  unpickling works without the reference package.
* HiFi-GAN: `generator.ckpt['generator']` (weight-norm parametrised, model.py:111) -> `generator.npz`, config.json copied.
Weight-norm folding happens later in zerovox_amd.pack, so the npz files stay faithful to the checkpoint.
Runs wherever torch is importable; its output is torch-free.
"""
import glob
import os
import shutil
import sys
import types

import numpy as np


class _SymbolsStub:
    """Stand-in for `zerovox.tts.symbols.Symbols` instances pickled inside a checkpoint's hyper_parameters: it only has to
    unpickle (the symbol tables themselves come from modelcfg.yaml)."""

    def __init__(self, *a, **k):
        pass


_SymbolsStub.__module__ = "zerovox.tts.symbols"
_SymbolsStub.__qualname__ = _SymbolsStub.__name__ = "Symbols"


def _install_symbols_stub():
    if "zerovox.tts.symbols" in sys.modules:
        return
    pkg = types.ModuleType("zerovox"); tts = types.ModuleType("zerovox.tts"); sym = types.ModuleType("zerovox.tts.symbols")
    sym.Symbols = _SymbolsStub
    pkg.tts = tts; tts.symbols = sym
    sys.modules.update({"zerovox": pkg, "zerovox.tts": tts, "zerovox.tts.symbols": sym})


def _safe_load(path):
    """torch.load restricted to tensors/containers (weights_only=True): a downloaded checkpoint must not be able to run
    code.  The only non-tensor class a ZeroVOX Lightning checkpoint pickles is `Symbols` (hyper_parameters); the stand-in
    is allow-listed explicitly."""
    import torch
    _install_symbols_stub()
    sym = sys.modules["zerovox.tts.symbols"].Symbols
    if hasattr(torch.serialization, "safe_globals"):
        with torch.serialization.safe_globals([sym]):
            return torch.load(path, map_location="cpu", weights_only=True)
    torch.serialization.add_safe_globals([sym])
    return torch.load(path, map_location="cpu", weights_only=True)


def convert_tts(src, modelcfg, out_dir):
    import torch
    _install_symbols_stub()
    if os.path.isdir(src):               # synthesize.py:295-299: newest checkpoints/*.ckpt
        files = glob.glob(os.path.join(src, "checkpoints", "*.ckpt"))
        src = max(files, key=os.path.getctime)
    ck = _safe_load(src)
    sd = ck.get("state_dict", ck.get("model", ck))
    tts, voc = {}, {}
    for k, v in sd.items():
        a = v.detach().cpu().numpy()
        (voc if k.startswith("_meldec.") else tts)[k[len("_meldec."):] if k.startswith("_meldec.") else k] = a
    os.makedirs(out_dir, exist_ok=True)
    np.savez(os.path.join(out_dir, "weights.npz"), **tts)
    if voc:
        np.savez(os.path.join(out_dir, "generator.npz"), **voc)
    shutil.copyfile(modelcfg, os.path.join(out_dir, "modelcfg.yaml"))
    return len(tts), len(voc)


def convert_vocoder(gen_ckpt, config_json, out_dir):
    import torch
    ck = _safe_load(gen_ckpt)
    sd = ck["generator"] if "generator" in ck else ck
    os.makedirs(out_dir, exist_ok=True)
    np.savez(os.path.join(out_dir, "generator.npz"), **{k: v.detach().cpu().numpy() for k, v in sd.items()})
    shutil.copyfile(config_json, os.path.join(out_dir, "config.json"))
    return len(sd)


if __name__ == "__main__":
    if len(sys.argv) != 5 or sys.argv[1] not in ("tts", "voc"):
        raise SystemExit(__doc__)
    if sys.argv[1] == "tts":
        print("tensors (tts, baked-in vocoder):", convert_tts(sys.argv[2], sys.argv[3], sys.argv[4]))
    else:
        print("tensors:", convert_vocoder(sys.argv[2], sys.argv[3], sys.argv[4]))
