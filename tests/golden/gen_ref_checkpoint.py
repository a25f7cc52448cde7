#!/usr/bin/env python3
"""A checkpoint WRITTEN BY THE REFERENCE.  Build container only (needs /root/reference + torch CPU).

The imported reference builds `ZeroVox(**kwargs)` for the reduced-width config of zerovox_amd.config.reduced_modelcfg (a few MB), takes the
seeded synthetic weights through its own `load_state_dict(strict=True)`, gets a HiFi-GAN generator baked in as `_meldec` the way
`get_meldec` leaves it (weight norm REMOVED: plain `weight` keys, model.py:111-115; utils/edit_meldec_in_checkpoint.py:77-90), and the
file is what Lightning writes (utils/dump_pkl.py:8-30): {"state_dict": zv.state_dict(), "hyper_parameters": <ctor kwargs incl. the real
`Symbols` object>, ...} through torch.save.  Next to it: modelcfg.yaml (the directory layout synthesize.py:295-304 reads) and the
reference's own `inference_ex` output on seeded inputs.  The GPU test converts / loads the directory and must reproduce that output.

    python tests/golden/gen_ref_checkpoint.py        # rewrites tests/golden/refckpt/ and tests/golden/refckpt_expected.npz
"""
import hashlib
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
warnings.filterwarnings("ignore")


def write_manifest():
    """sha256 of everything this script writes -> tests/golden/REFCKPT_MANIFEST.json (checked by tests/test_oracle_golden.py::test_manifest_hashes;
    `python tests/golden/gen_ref_checkpoint.py --manifest-only` re-hashes the committed files without importing the reference)."""
    import json
    files = ["refckpt_expected.npz", "refckpt/checkpoints/epoch=0-step=0.ckpt", "refckpt/modelcfg.yaml", "refckpt/meldec_config.yaml"]
    man = {f: {"sha256": hashlib.sha256(open(os.path.join(HERE, f), "rb").read()).hexdigest(), "bytes": os.path.getsize(os.path.join(HERE, f))} for f in files}
    with open(os.path.join(HERE, "REFCKPT_MANIFEST.json"), "w") as f:
        json.dump({"generator": "tests/golden/gen_ref_checkpoint.py", "files": man}, f, indent=1, sort_keys=True)
    print("wrote REFCKPT_MANIFEST.json")



if __name__ == "__main__" and "--manifest-only" in sys.argv:
    write_manifest()
    sys.exit(0)

import ref_import  # noqa: E402

ref_import.install()
import torch  # noqa: E402
import yaml  # noqa: E402
from zerovox.tts.hifigan import Generator  # noqa: E402
from zerovox.tts.model import AttrDict, ZeroVox  # noqa: E402
from zerovox.tts.symbols import Symbols  # noqa: E402

from zerovox_amd import config as zcfg  # noqa: E402
from zerovox_amd import weights as zw  # noqa: E402

SEED = 11
KIND, VOC = "styletts", "tiny3"


def main():
    torch.set_num_threads(4)
    cfg = zcfg.reduced_modelcfg(KIND)
    kwargs = zcfg.zerovox_kwargs(cfg)
    symbols = Symbols(zcfg.PHONES, zcfg.PUNCTS)
    zv = ZeroVox(symbols=symbols, meldec_model=None, **kwargs)
    sd = zw.tts_state_dict(cfg, SEED)
    zv.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    h = zcfg.hifigan_config(VOC)
    gen = Generator(AttrDict(h))
    gen.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in zw.hifigan_state_dict(h, SEED).items()}, strict=True)
    gen.eval(); gen.remove_weight_norm()                                   # model.py:113-115: what get_meldec returns
    zv._meldec = gen
    zv.eval()

    out = os.path.join(HERE, "refckpt")
    os.makedirs(os.path.join(out, "checkpoints"), exist_ok=True)
    hp = dict(kwargs); hp["symbols"] = symbols                             # save_hyperparameters(ignore=['meldec_model', 'verbose']), model.py:204
    ck = os.path.join(out, "checkpoints", "epoch=0-step=0.ckpt")
    torch.save({"epoch": 0, "global_step": 0, "pytorch-lightning_version": "2.2.0", "state_dict": zv.state_dict(), "hyper_parameters": hp}, ck)
    yaml.safe_dump(cfg, open(os.path.join(out, "modelcfg.yaml"), "w"))
    yaml.safe_dump(h, open(os.path.join(out, "meldec_config.yaml"), "w"))   # the baked-in generator's topology (config.json of the vocoder directory)
    n_meldec = sum(1 for k in zv.state_dict() if k.startswith("_meldec."))
    print(f"wrote {ck}: {os.path.getsize(ck) / 1e6:.2f} MB, {len(zv.state_dict())} tensors ({n_meldec} under _meldec.)")

    # the reference's own synthesis on seeded inputs
    r = np.random.default_rng(77)
    T = 24
    phoneme = r.integers(1, len(zcfg.PHONES) + 1, size=T).astype(np.int32)
    puncts = r.integers(1, len(zcfg.PUNCTS) + 2, size=T).astype(np.int32)
    spk = r.standard_normal(32); spk = (spk / np.linalg.norm(spk)).astype(np.float32)
    dur = r.integers(2, 7, size=T).astype(np.int32)
    res = {}
    for forced in (True, False):
        zv._min_mel_len = 689
        with torch.no_grad():
            x = {"phoneme": torch.from_numpy(phoneme[None]).int(), "puncts": torch.from_numpy(puncts[None]).int(),
                 "duration": torch.from_numpy(dur[None]).int() if forced else None}
            wav, mel_len, logd, mel = zv.inference_ex(x, style_embed=torch.from_numpy(spk).reshape(1, 1, -1), force_duration=forced)
        tag = "forced" if forced else "pred"
        ml = int(mel_len[0]) if hasattr(mel_len, "__len__") else int(mel_len)
        res[f"{tag}_wav"] = np.asarray(wav.detach().cpu().numpy()).reshape(-1)[: ml * 256].astype(np.float32)
        res[f"{tag}_mel"] = np.asarray(mel.detach().cpu().numpy()).astype(np.float32)             # [80, mel_len] (model.py:347)
        res[f"{tag}_mel_len"] = np.int32(ml)
        res[f"{tag}_log_duration"] = np.asarray(logd.detach().cpu().numpy())[0].astype(np.float32)
        print(f"  {tag}: mel_len {ml}, wav rms {float(np.sqrt((res[f'{tag}_wav'] ** 2).mean())):.3f}, mel shape {res[f'{tag}_mel'].shape}")
    path = os.path.join(HERE, "refckpt_expected.npz")
    np.savez_compressed(path, phoneme=phoneme, puncts=puncts, spk=spk, duration=dur, seed=np.int32(SEED), **res)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB), checkpoint sha256 {hashlib.sha256(open(ck, 'rb').read()).hexdigest()[:16]}")
    write_manifest()


if __name__ == "__main__":
    main()
