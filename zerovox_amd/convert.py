"""Reading the reference's checkpoints: Lightning `.ckpt` / `checkpoint.pkl` and HiFi-GAN `generator.ckpt`
-> the NumPy state dicts zerovox_amd.pack consumes.  torch is imported only here, only to unpickle; everything downstream is
torch-free.  `tools/convert_checkpoint.py` is the command-line wrapper that writes the dicts as .npz files.

* TTS: `state_dict` / `hyper_parameters` (layouts in utils/dump_pkl.py:8-30); `_meldec.*` keys (a vocoder baked into the
  checkpoint, utils/edit_meldec_in_checkpoint.py:77-90) are split off -- zerovox_amd.model.load_meldec_weights prefers them
  over the external vocoder's weights, as `ZeroVox.load_from_checkpoint(strict=False)` does.  The pickled `hyper_parameters`
  reference `zerovox.tts.symbols.Symbols`; a stand-in class is allow-listed so that unpickling works without the reference
  package and without executing arbitrary pickles (`weights_only=True`).
* HiFi-GAN: `generator.ckpt['generator']` (weight-norm parametrised, model.py:111).  Weight-norm folding happens later in
  zerovox_amd.pack, so the dicts stay faithful to the checkpoint.
"""
import glob
import os
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
    """Stand-in modules for the one class a ZeroVOX checkpoint pickles; returns the names it added (so that a real `zerovox`
    package imported later in the same process is not shadowed: the caller removes them again)."""
    if "zerovox.tts.symbols" in sys.modules:
        return []
    pkg = types.ModuleType("zerovox"); tts = types.ModuleType("zerovox.tts"); sym = types.ModuleType("zerovox.tts.symbols")
    sym.Symbols = _SymbolsStub
    pkg.tts = tts; tts.symbols = sym
    added = {k: v for k, v in {"zerovox": pkg, "zerovox.tts": tts, "zerovox.tts.symbols": sym}.items() if k not in sys.modules}
    sys.modules.update(added)
    return list(added)


def _safe_load(path):
    """torch.load restricted to tensors/containers (weights_only=True): a downloaded checkpoint must not be able to run
    code.  The only non-tensor class a ZeroVOX Lightning checkpoint pickles is `Symbols` (hyper_parameters); the stand-in
    is allow-listed explicitly and its stub modules leave sys.modules again once the file is read."""
    import torch
    added = _install_symbols_stub()
    try:
        sym = sys.modules["zerovox.tts.symbols"].Symbols
        if hasattr(torch.serialization, "safe_globals"):
            with torch.serialization.safe_globals([sym]):
                return torch.load(path, map_location="cpu", weights_only=True)
        torch.serialization.add_safe_globals([sym])
        return torch.load(path, map_location="cpu", weights_only=True)
    finally:
        for k in added:
            sys.modules.pop(k, None)


def find_tts_checkpoint(modeldir):
    """synthesize.py:295-304: the newest `checkpoints/*.ckpt` of a model directory, else `checkpoint.pkl`; None if neither."""
    files = glob.glob(os.path.join(modeldir, "checkpoints", "*.ckpt"))
    if files:
        return max(files, key=os.path.getctime)
    p = os.path.join(modeldir, "checkpoint.pkl")
    return p if os.path.exists(p) else None


def read_tts_checkpoint(src):
    """-> (tts_state_dict, baked_in_vocoder_state_dict) as NumPy arrays under the reference's key names."""
    if os.path.isdir(src):
        src = find_tts_checkpoint(src)
        if src is None:
            raise FileNotFoundError("no checkpoints/*.ckpt or checkpoint.pkl in the model directory")
    ck = _safe_load(src)
    sd = ck.get("state_dict", ck.get("model", ck))
    tts, voc = {}, {}
    for k, v in sd.items():
        a = v.detach().cpu().numpy()
        (voc if k.startswith("_meldec.") else tts)[k[len("_meldec."):] if k.startswith("_meldec.") else k] = a
    return tts, voc


def read_generator_checkpoint(gen_ckpt):
    ck = _safe_load(gen_ckpt)
    sd = ck["generator"] if "generator" in ck else ck
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}
