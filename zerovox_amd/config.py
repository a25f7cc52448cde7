"""Model / vocoder hyper-parameters for the ZeroVOX synthesis path.

The dict layout mirrors the reference's ``modelcfg.yaml`` (configs/tts_medium_styledec.yaml:1-59)
and the kwargs mapping of ``utils/train_tts.py:202-241`` (yaml key -> ``ZeroVox`` ctor argument).
HiFi-GAN generator configs are data (the reference downloads ``config.json`` at run time,
``model.py:86-118``); V1/V2/V3 are the published jik876 configurations, ``tiny`` is a test-size one.
"""
from __future__ import annotations

import copy

PHONES = "'-abcdefghijklmnopqrstuvwxyz"   # configs/tts_medium_styledec.yaml:18
PUNCTS = " ,.;:-!?\""                      # configs/tts_medium_styledec.yaml:19


def medium_modelcfg(decoder_kind: str = "styletts") -> dict:
    """``tts_medium_styledec.yaml`` (kind='styletts') / ``tts_medium.yaml`` (kind='fastspeech2')."""
    if decoder_kind not in ("styletts", "fastspeech2"):
        raise Exception(f"unknown decoder kind: '{decoder_kind}'")   # model.py:244
    return {
        "lang": ["en"],
        "audio": {"sampling_rate": 22050, "fft_size": 1024, "fmax": 8000, "fmin": 0,
                  "win_length": 1024, "num_mels": 80, "hop_size": 256},
        "model": {
            "max_txt_len": 512, "min_mel_len": 100, "max_mel_len": 1750,
            "phones": PHONES, "puncts": PUNCTS,
            "emb_dim": 512, "emb_reduction": 1, "punct_emb_dim": 16, "dpe_emb_dim": 32,
            "encoder": {"fs2_layer": 4, "fs2_head": 2, "fs2_dropout": 0.2, "vp_filter_size": 256,
                        "vp_kernel_size": 3, "vp_dropout": 0.5, "ve_n_bins": 256},
            "decoder": {"kind": decoder_kind, "n_layers": 6, "n_head": 2, "conv_filter_size": 1024,
                        "conv_kernel_size": [9, 1], "dropout": 0.2, "scln": True},
            "resnet": {"layers": [3, 4, 6, 3], "num_filters": [32, 64, 128, 256],
                       "encoder_type": "ASP"},
        },
    }


def reduced_modelcfg(decoder_kind: str = "styletts") -> dict:
    """A reduced-width model of the same topology (hidden 32, 2 + 2 layers, a one-block-per-level speaker encoder): ~2 MB of
    weights, so that a checkpoint WRITTEN BY THE REFERENCE ITSELF fits the repository (tests/golden/gen_ref_checkpoint.py)."""
    cfg = medium_modelcfg(decoder_kind)
    m = cfg["model"]
    m.update({"max_txt_len": 64, "max_mel_len": 256, "emb_dim": 16, "punct_emb_dim": 16})
    m["encoder"].update({"fs2_layer": 2, "vp_filter_size": 32, "ve_n_bins": 64})
    m["decoder"].update({"n_layers": 2, "conv_filter_size": 64})
    m["resnet"].update({"layers": [1, 1, 1, 1], "num_filters": [8, 8, 16, 16]})
    return cfg


def zerovox_kwargs(modelcfg: dict) -> dict:
    """yaml -> ``ZeroVox.__init__`` kwargs (utils/train_tts.py:202-241), training-only ones stubbed."""
    m = modelcfg["model"]
    return dict(
        sampling_rate=modelcfg["audio"]["sampling_rate"], hop_length=modelcfg["audio"]["hop_size"],
        n_mels=modelcfg["audio"]["num_mels"], lr=1e-4, weight_decay=0.0, betas=[0.0, 0.99], eps=1e-9,
        max_epochs=1, warmup_epochs=1,
        embed_dim=m["emb_dim"], punct_embed_dim=m["punct_emb_dim"], dpe_embed_dim=m["dpe_emb_dim"],
        emb_reduction=m["emb_reduction"], max_txt_len=m["max_txt_len"], max_mel_len=m["max_mel_len"],
        fs2enc_layer=m["encoder"]["fs2_layer"], fs2enc_head=m["encoder"]["fs2_head"],
        fs2enc_dropout=m["encoder"]["fs2_dropout"], vp_filter_size=m["encoder"]["vp_filter_size"],
        vp_kernel_size=m["encoder"]["vp_kernel_size"], vp_dropout=m["encoder"]["vp_dropout"],
        ve_n_bins=m["encoder"]["ve_n_bins"],
        resnet_layers=m["resnet"]["layers"], resnet_num_filters=m["resnet"]["num_filters"],
        resnet_encoder_type=m["resnet"]["encoder_type"],
        decoder_kind=m["decoder"]["kind"], decoder_n_layers=m["decoder"]["n_layers"],
        decoder_n_head=m["decoder"]["n_head"], decoder_conv_filter_size=m["decoder"]["conv_filter_size"],
        decoder_conv_kernel_size=m["decoder"]["conv_kernel_size"], decoder_dropout=m["decoder"]["dropout"],
        decoder_scln=m["decoder"]["scln"],
    )


_HIFIGAN = {
    # jik876/hifi-gan config_v1.json / v2 / v3 (not in the reference repo)
    "v1": {"resblock": "1", "upsample_rates": [8, 8, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4],
           "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
           "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]]},
    "v2": {"resblock": "1", "upsample_rates": [8, 8, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4],
           "upsample_initial_channel": 128, "resblock_kernel_sizes": [3, 7, 11],
           "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]]},
    "v3": {"resblock": "2", "upsample_rates": [8, 8, 4], "upsample_kernel_sizes": [16, 16, 8],
           "upsample_initial_channel": 256, "resblock_kernel_sizes": [3, 5, 7],
           "resblock_dilation_sizes": [[1, 2], [2, 6], [3, 12]]},
    # test-size generators (same topology rules, hop 256)
    "tiny": {"resblock": "1", "upsample_rates": [8, 8, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4],
             "upsample_initial_channel": 128, "resblock_kernel_sizes": [3, 7],
             "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5]]},
    "tiny3": {"resblock": "1", "upsample_rates": [8, 8, 4], "upsample_kernel_sizes": [16, 16, 8],
              "upsample_initial_channel": 64, "resblock_kernel_sizes": [3, 7],
              "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5]]},
    "tiny2": {"resblock": "2", "upsample_rates": [8, 8, 4], "upsample_kernel_sizes": [16, 16, 8],
              "upsample_initial_channel": 64, "resblock_kernel_sizes": [3, 5],
              "resblock_dilation_sizes": [[1, 2], [2, 6]]},
}


def hifigan_config(name: str) -> dict:
    return copy.deepcopy(_HIFIGAN[name])
