#!/usr/bin/env python3
"""Checkpoint converter: reference checkpoints -> the torch-free files zerovox_amd loads.

    tools/convert_checkpoint.py tts  <model_dir | checkpoint.(ckpt|pkl)> <modelcfg.yaml> <out_dir>
    tools/convert_checkpoint.py voc  <generator.ckpt> <config.json> <out_dir>

Thin wrapper around zerovox_amd/convert.py (which documents the layouts): writes `<out_dir>/weights.npz` (+ `generator.npz`
when a vocoder is baked into the TTS checkpoint) + modelcfg.yaml, resp. `generator.npz` + config.json.
`ZeroVoxTTS.load_model` also reads a reference model directory (checkpoints/*.ckpt + modelcfg.yaml) directly.
"""
import os
import shutil
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zerovox_amd.convert import _install_symbols_stub, read_generator_checkpoint, read_tts_checkpoint  # noqa: E402,F401


def convert_tts(src, modelcfg, out_dir):
    tts, voc = read_tts_checkpoint(src)
    os.makedirs(out_dir, exist_ok=True)
    np.savez(os.path.join(out_dir, "weights.npz"), **tts)
    if voc:
        np.savez(os.path.join(out_dir, "generator.npz"), **voc)
    shutil.copyfile(modelcfg, os.path.join(out_dir, "modelcfg.yaml"))
    return len(tts), len(voc)


def convert_vocoder(gen_ckpt, config_json, out_dir):
    sd = read_generator_checkpoint(gen_ckpt)
    os.makedirs(out_dir, exist_ok=True)
    np.savez(os.path.join(out_dir, "generator.npz"), **sd)
    shutil.copyfile(config_json, os.path.join(out_dir, "config.json"))
    return len(sd)


if __name__ == "__main__":
    if len(sys.argv) != 5 or sys.argv[1] not in ("tts", "voc"):
        raise SystemExit(__doc__)
    if sys.argv[1] == "tts":
        print("tensors (tts, baked-in vocoder):", convert_tts(sys.argv[2], sys.argv[3], sys.argv[4]))
    else:
        print("tensors:", convert_vocoder(sys.argv[2], sys.argv[3], sys.argv[4]))
