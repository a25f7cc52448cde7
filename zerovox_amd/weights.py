"""Synthetic, seeded model weights in the reference's ``state_dict`` naming.

No pretrained checkpoints are reachable offline (the reference downloads them, model.py:66-82), so
parity and benchmarks run on weights drawn from a counter-based NumPy PRNG: every tensor is
``w(seed, tensor_name, shape)`` and therefore reproducible on any box without shipping data.
The init is variance-preserving (fan-in scaled) so that signals neither vanish nor explode through
the ~80 layers of the path; the stock HiFi-GAN ``init_weights`` std 0.01 (hifigan.py:17-20) would
make outputs vanish and parity tests meaningless.

Names/shapes follow the reference modules:
  _phoneme_encoder.*  fs2.py:317-401 (Encoder), :506-563 (VariancePredictor), :575-693
  _spkemb.*           ResNetSE34V2.py:102-152
  _mel_decoder.*      fs2.py:232-315 (FS2Decoder) | styletts.py:142-205 (StyleTTSDecoder)
  generator           hifigan.py:89-139 (weight-norm parametrised: weight_g / weight_v)
"""
from __future__ import annotations

import zlib

import numpy as np

from .config import PHONES, PUNCTS


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([int(seed), zlib.crc32(name.encode("utf-8"))])


def sinusoid_table(n_position: int, d_hid: int) -> np.ndarray:
    """fs2.py:17-37 -- angle = pos / 10000^(2*(j//2)/d_hid); sin on even j, cos on odd j (float64 -> f32)."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    table = pos / np.power(10000.0, 2.0 * (j // 2) / d_hid)[None, :]
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return table.astype(np.float32)


class _Spec:
    """Ordered (name -> ndarray) builder with init helpers."""

    def __init__(self, seed):
        self.seed = seed
        self.sd: dict[str, np.ndarray] = {}

    def normal(self, name, shape, std=1.0, mean=0.0):
        a = _rng(self.seed, name).standard_normal(shape) * std + mean
        self.sd[name] = a.astype(np.float32)
        return self.sd[name]

    def fanin(self, name, shape, gain=1.0, fan_in=None):
        if fan_in is None:
            fan_in = int(np.prod(shape[1:]))
        return self.normal(name, shape, std=gain / np.sqrt(fan_in))

    def const(self, name, arr):
        self.sd[name] = np.asarray(arr)

    def linear(self, prefix, out_f, in_f, gain=1.0, bias=True, bias_std=0.05):
        self.fanin(prefix + ".weight", (out_f, in_f), gain)
        if bias:
            self.normal(prefix + ".bias", (out_f,), bias_std)

    def conv(self, prefix, cout, cin, k, gain=1.0, bias=True, bias_std=0.05):
        self.fanin(prefix + ".weight", (cout, cin, k), gain)
        if bias:
            self.normal(prefix + ".bias", (cout,), bias_std)

    def ln(self, prefix, n):
        self.normal(prefix + ".weight", (n,), 0.1, 1.0)
        self.normal(prefix + ".bias", (n,), 0.1)

    def bn(self, prefix, n):
        self.normal(prefix + ".weight", (n,), 0.1, 1.0)
        self.normal(prefix + ".bias", (n,), 0.1)
        self.normal(prefix + ".running_mean", (n,), 0.1)
        self.sd[prefix + ".running_var"] = (1.0 + 0.2 * np.abs(
            _rng(self.seed, prefix + ".running_var").standard_normal(n))).astype(np.float32)
        self.const(prefix + ".num_batches_tracked", np.zeros((), dtype=np.int64))

    def wn(self, prefix, shape, gain=1.0, fan_in=None, bias=None, bias_std=0.05):
        """weight-norm pair (torch weight_norm dim=0: g has shape [shape[0],1,1], w = g*v/||v||)."""
        if fan_in is None:
            fan_in = int(np.prod(shape[1:]))
        w = _rng(self.seed, prefix + ".weight").standard_normal(shape) * (gain / np.sqrt(fan_in))
        r = _rng(self.seed, prefix + ".weight_r").uniform(0.5, 2.0, size=(shape[0],) + (1,) * (len(shape) - 1))
        g = np.sqrt((w ** 2).sum(axis=tuple(range(1, len(shape))), keepdims=True))
        if bias is not None:
            self.normal(prefix + ".bias", (bias,), bias_std)
        self.sd[prefix + ".weight_g"] = g.astype(np.float32)
        self.sd[prefix + ".weight_v"] = (w * r).astype(np.float32)


def _fft_block(s: _Spec, p: str, H: int, F: int, ksz, scln: bool, spk: int):
    for nm in ("w_qs", "w_ks", "w_vs", "fc"):
        s.linear(f"{p}.slf_attn.{nm}", H, H)
        if nm == "w_vs":   # keep ordering of the reference's state_dict (layer_norm sits before fc)
            if scln:
                _scln(s, f"{p}.slf_attn.layer_norm", spk, H)
            else:
                s.ln(f"{p}.slf_attn.layer_norm", H)
    s.conv(f"{p}.pos_ffn.w_1", F, H, ksz[0], gain=np.sqrt(2.0))
    s.conv(f"{p}.pos_ffn.w_2", H, F, ksz[1], gain=1.0)
    if scln:
        _scln(s, f"{p}.pos_ffn.layer_norm", spk, H)
    else:
        s.ln(f"{p}.pos_ffn.layer_norm", H)


def _scln(s: _Spec, p: str, spk: int, H: int):
    # affine_layer = Linear(spk, 2H, bias=False); [b | g] = W s   (fs2.py:69-73,85)
    w = np.concatenate([
        _rng(s.seed, p + ".b").standard_normal((H, spk)) * 0.3,
        _rng(s.seed, p + ".g").standard_normal((H, spk)) * 1.0,
    ], axis=0)
    s.const(p + ".affine_layer.linear.weight", w.astype(np.float32))


def _variance_predictor(s: _Spec, p: str, H: int, Fv: int, k: int, out_bias: float, out_std: float):
    s.conv(f"{p}.conv_layer.conv1d_1.conv", Fv, H, k, gain=1.0)
    s.ln(f"{p}.conv_layer.layer_norm_1", Fv)
    s.conv(f"{p}.conv_layer.conv1d_2.conv", Fv, Fv, k, gain=np.sqrt(2.0))
    s.ln(f"{p}.conv_layer.layer_norm_2", Fv)
    s.normal(f"{p}.linear_layer.weight", (1, Fv), out_std / np.sqrt(Fv))
    s.const(f"{p}.linear_layer.bias", np.array([out_bias], dtype=np.float32))


def tts_state_dict(modelcfg: dict, seed: int = 0) -> dict[str, np.ndarray]:
    """Synthetic ``ZeroVox.state_dict()`` (without ``_meldec.*``) for ``modelcfg`` (see config.py)."""
    m = modelcfg["model"]
    enc, dec, rn = m["encoder"], m["decoder"], m["resnet"]
    H = m["emb_dim"] + m["punct_emb_dim"]
    n_mels = modelcfg["audio"]["num_mels"]
    s = _Spec(seed)

    # ---- _phoneme_encoder._encoder (fs2.py:350-368) ----
    pe = "_phoneme_encoder._encoder"
    s.const(f"{pe}.position_enc", sinusoid_table(m["max_txt_len"] + 1, H)[None])
    emb = s.normal(f"{pe}.src_word_emb.weight", (len(m["phones"]) + 1, m["emb_dim"]), 1.0)
    emb[0] = 0.0   # padding_idx=0 row (fs2.py:350) -- never trained, zero in a real checkpoint
    pemb = s.normal(f"{pe}.punct_embed.weight", (len(m["puncts"]) + 2, m["punct_emb_dim"]), 1.0)
    pemb[0] = 0.0
    for i in range(enc["fs2_layer"]):
        # NB: the encoder FFN uses the DECODER's conv_filter_size / kernel_size (model.py:211-212)
        _fft_block(s, f"{pe}.layer_stack.{i}", H, dec["conv_filter_size"], dec["conv_kernel_size"], False, 0)

    # ---- _phoneme_encoder._variance_adaptor (fs2.py:586-625) ----
    va = "_phoneme_encoder._variance_adaptor"
    Fv, kv = enc["vp_filter_size"], enc["vp_kernel_size"]
    # output heads are biased so synthetic predictions are meaningful: log-duration ~ 1.9 +- 0.3
    # (=> 3..10 frames per phoneme), pitch/energy ~ 0.5 +- 0.25 (=> buckets spread over 0..255)
    _variance_predictor(s, f"{va}.duration_predictor", H, Fv, kv, 1.9, 0.3)
    _variance_predictor(s, f"{va}.pitch_predictor", H, Fv, kv, 0.5, 0.25)
    _variance_predictor(s, f"{va}.energy_predictor", H, Fv, kv, 0.5, 0.25)
    s.normal(f"{va}.pitch_embedding.weight", (enc["ve_n_bins"], H), 0.5)
    s.normal(f"{va}.energy_embedding.weight", (enc["ve_n_bins"], H), 0.5)

    # ---- _spkemb (ResNetSE34V2.py:102-152) ----
    sp = "_spkemb"
    nf, layers = rn["num_filters"], rn["layers"]
    s.fanin(f"{sp}.conv1.weight", (nf[0], 1, 3, 3), gain=np.sqrt(2.0))
    s.normal(f"{sp}.conv1.bias", (nf[0],), 0.05)
    s.bn(f"{sp}.bn1", nf[0])
    inplanes = nf[0]
    for li, (planes, nblk) in enumerate(zip(nf, layers), start=1):
        for bi in range(nblk):
            p = f"{sp}.layer{li}.{bi}"
            cin = inplanes if bi == 0 else planes
            s.fanin(f"{p}.conv1.weight", (planes, cin, 3, 3), gain=np.sqrt(2.0))
            s.bn(f"{p}.bn1", planes)
            s.fanin(f"{p}.conv2.weight", (planes, planes, 3, 3), gain=1.0)
            s.bn(f"{p}.bn2", planes)
            s.linear(f"{p}.se.fc.0", planes // 8, planes, gain=1.0)
            s.linear(f"{p}.se.fc.2", planes, planes // 8, gain=2.0)
            if bi == 0 and (li > 1 or inplanes != planes):
                s.fanin(f"{p}.downsample.0.weight", (planes, cin, 1, 1), gain=1.0)
                s.bn(f"{p}.downsample.1", planes)
        inplanes = planes
    s.const(f"{sp}.torchfb.0.flipped_filter", np.array([[[-0.97, 1.0]]], dtype=np.float32))
    D = nf[3] * (n_mels // 8)
    s.conv(f"{sp}.attention.0", 128, D, 1, gain=1.0)
    s.bn(f"{sp}.attention.2", 128)
    s.conv(f"{sp}.attention.3", D, 128, 1, gain=2.0)
    n_asp = D * 2 if rn["encoder_type"] == "ASP" else D
    s.linear(f"{sp}.fc", H, n_asp, gain=1.0)

    # ---- _mel_decoder ----
    md = "_mel_decoder"
    if dec["kind"] == "fastspeech2":
        s.const(f"{md}.position_enc", sinusoid_table(m["max_mel_len"] + 1, H)[None])
        for i in range(dec["n_layers"]):
            _fft_block(s, f"{md}.layer_stack.{i}", H, dec["conv_filter_size"], dec["conv_kernel_size"],
                       dec["scln"], H)
        s.linear(f"{md}.mel_linear", n_mels, H)
    elif dec["kind"] == "styletts":
        B2, R = 2 * H, 64
        g_act = 1.3   # LeakyReLU(0.2) after a unit-variance norm keeps ~0.6 of the variance

        def adain_blk(p, cin, cout):
            s.wn(f"{p}.conv1", (cout, cin, 3), gain=g_act, bias=cout)
            s.wn(f"{p}.conv2", (cout, cout, 3), gain=g_act, bias=cout)
            s.linear(f"{p}.norm1.fc", 2 * cin, H, gain=0.5, bias_std=0.1)
            s.linear(f"{p}.norm2.fc", 2 * cout, H, gain=0.5, bias_std=0.1)
            if cin != cout:
                s.wn(f"{p}.conv1x1", (cout, cin, 1), gain=1.0)

        def res_blk(p, cin, cout):
            s.wn(f"{p}.conv1", (cin, cin, 3), gain=g_act, bias=cin)
            s.wn(f"{p}.conv2", (cout, cin, 3), gain=g_act, bias=cout)
            s.ln(f"{p}.norm1", cin)   # InstanceNorm1d(affine=True): weight/bias per channel
            s.ln(f"{p}.norm2", cin)
            if cin != cout:
                s.wn(f"{p}.conv1x1", (cout, cin, 1), gain=1.0)

        adain_blk(f"{md}.decode.0", B2 + R, B2)
        adain_blk(f"{md}.decode.1", B2 + R, B2)
        adain_blk(f"{md}.decode.2", B2 + R, H)
        adain_blk(f"{md}.decode.3", H, H)
        adain_blk(f"{md}.decode.4", H, H)
        res_blk(f"{md}.encode.0", H, B2)
        res_blk(f"{md}.encode.1", B2, B2)
        s.wn(f"{md}.asr_res.0", (R, H, 1), gain=1.0, bias=R)
        s.ln(f"{md}.asr_res.1", R)
        s.wn(f"{md}.to_out.0", (n_mels, H, 1), gain=1.0, bias=n_mels, bias_std=0.5)
    else:
        raise Exception(f"unknown decoder kind: '{dec['kind']}'")
    return s.sd


def hifigan_state_dict(h: dict, seed: int = 0) -> dict[str, np.ndarray]:
    """Synthetic ``{'generator': ...}`` payload of a HiFi-GAN ``generator.ckpt`` (weight-norm form).

    Layout as built by hifigan.py:95-110.  ConvTranspose1d weights are [Cin, Cout, k] and torch's
    weight_norm(dim=0) therefore normalises per INPUT channel (weight_g is [Cin,1,1]).
    """
    s = _Spec(seed)
    C0 = h["upsample_initial_channel"]
    s.wn("conv_pre", (C0, 80, 7), gain=1.0, bias=C0)
    ch = C0
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        cin, cout = C0 // (2 ** i), C0 // (2 ** (i + 1))
        # each output sample sees k/u taps x cin channels; leaky-relu(0.1) input keeps ~half the variance
        s.wn(f"ups.{i}", (cin, cout, k), gain=np.sqrt(2.0), fan_in=cin * k // u, bias=cout)
    nk = len(h["resblock_kernel_sizes"])
    for i in range(len(h["upsample_rates"])):
        ch = C0 // (2 ** (i + 1))
        for j, (k, d) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            p = f"resblocks.{i * nk + j}"
            if h["resblock"] == "1":
                for t in range(len(d)):
                    s.wn(f"{p}.convs1.{t}", (ch, ch, k), gain=1.0, bias=ch)
                for t in range(len(d)):
                    s.wn(f"{p}.convs2.{t}", (ch, ch, k), gain=0.7, bias=ch)
            else:
                for t in range(len(d)):
                    s.wn(f"{p}.convs.{t}", (ch, ch, k), gain=0.7, bias=ch)
    # keep the pre-tanh signal ~0.3 rms so tanh is exercised but not saturated
    s.wn("conv_post", (1, ch, 7), gain=0.15, bias=1, bias_std=0.01)
    # order like the reference's state_dict: bias, weight_g, weight_v per module is irrelevant for loading
    return s.sd


def fold_weight_norm(g: np.ndarray, v: np.ndarray) -> np.ndarray:
    """w = g * v / ||v||, norm over all dims but 0 (torch weight_norm dim=0; hifigan.py:132-139)."""
    v64 = v.astype(np.float64)
    nrm = np.sqrt((v64 ** 2).sum(axis=tuple(range(1, v.ndim)), keepdims=True))
    return (g.astype(np.float64) * v64 / nrm).astype(np.float32)


def folded(sd: dict[str, np.ndarray]) -> dict[str, np.ndarray]:
    """Replace every (weight_g, weight_v) pair by its folded ``weight``."""
    out = {}
    for k, a in sd.items():
        if k.endswith(".weight_g"):
            base = k[: -len(".weight_g")]
            out[base + ".weight"] = fold_weight_norm(a, sd[base + ".weight_v"])
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = a
    return out


__all__ = ["tts_state_dict", "hifigan_state_dict", "fold_weight_norm", "folded", "sinusoid_table",
           "PHONES", "PUNCTS"]
