"""In-container oracle aid: make the reference importable with two in-process stubs.

Used ONLY by tests/golden/gen_golden.py and ad-hoc probes in the build container.  Never shipped
on a product path and never importable on the GPU box (/root/reference does not exist there).
The stubs replace modules whose functionality is not exercised on the inference path:
torchaudio.transforms.MelSpectrogram (ctor-only, ResNetSE34V2.py:125) and
lightning.LightningModule (ZeroVox only needs nn.Module + save_hyperparameters, model.py:204).
"""
import sys
import types

import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


def install():
    if "zerovox.tts.model" in sys.modules:
        return
    ta = types.ModuleType("torchaudio")
    ta.transforms = types.ModuleType("torchaudio.transforms")

    class _MelSpectrogram(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    ta.transforms.MelSpectrogram = _MelSpectrogram
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.transforms"] = ta.transforms

    lt = types.ModuleType("lightning")

    class _LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

    lt.LightningModule = _LightningModule
    sys.modules["lightning"] = lt
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
