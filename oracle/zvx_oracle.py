"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  NumPy restatement of the ZeroVOX synthesis path.

This file is the checker, not the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path (``zerovox_amd``) never does
and fails loudly when the HIP library is missing.

Parity pin: the reference ships NO tests or golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, imported in the build container with
seeded synthetic weights (``tests/golden/gen_golden.py`` -> ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every fixture).

Each function cites the reference lines it restates (paths relative to /root/reference/zerovox/tts).
Layout convention here follows the reference: sequences are [T, C] ("time-major") for the
FastSpeech2 blocks and [C, T] for the conv stacks; all functions are batch-1 like
``ZeroVox.inference_ex`` (model.py:308-347).
"""
from __future__ import annotations

import math

import numpy as np

LRELU_SLOPE = 0.1   # hifigan.py:15


# --------------------------------------------------------------------------------------------
# primitive ops (semantics of the torch ops the reference relies on, SURVEY.md §8c last row)
# --------------------------------------------------------------------------------------------

def linear(x, w, b=None):
    """torch.nn.Linear: y = x W^T + b."""
    y = x @ w.T
    return y if b is None else y + b


def conv1d(x, w, b=None, dilation=1, padding=0, stride=1):
    """torch.nn.Conv1d on x [Cin, T], w [Cout, Cin, K] (cross-correlation, zero padding).

    Evaluated as time-chunked im2col + one sgemm per chunk (chunks sized to stay cache-resident)."""
    cin, T = x.shape
    cout, _, K = w.shape
    xp = np.zeros((cin, T + 2 * padding), dtype=x.dtype)
    xp[:, padding:padding + T] = x
    Tout = (T + 2 * padding - dilation * (K - 1) - 1) // stride + 1
    w2 = np.ascontiguousarray(w.transpose(0, 2, 1)).reshape(cout, K * cin)      # [cout][k][cin]
    y = np.empty((cout, Tout), dtype=x.dtype)
    chunk = max(256, min(Tout, (1 << 21) // (K * cin)))
    for t0 in range(0, Tout, chunk):
        n = min(chunk, Tout - t0)
        col = np.empty((K, cin, n), dtype=x.dtype)
        for k in range(K):
            s0 = t0 * stride + k * dilation
            col[k] = xp[:, s0: s0 + (n - 1) * stride + 1: stride]
        y[:, t0:t0 + n] = w2 @ col.reshape(K * cin, n)
    if b is not None:
        y += b[:, None]
    return y


def conv_transpose1d(x, w, b=None, stride=1, padding=0):
    """torch.nn.ConvTranspose1d on x [Cin, T], w [Cin, Cout, K]; Tout = (T-1)*s - 2p + K."""
    cin, T = x.shape
    _, cout, K = w.shape
    full = np.zeros((cout, (T - 1) * stride + K), dtype=x.dtype)
    for k in range(K):
        full[:, k: k + (T - 1) * stride + 1: stride] += w[:, :, k].T @ x
    y = full[:, padding: full.shape[1] - padding]
    if b is not None:
        y = y + b[:, None]
    return y


def conv2d(x, w, b=None, stride=1, padding=0):
    """torch.nn.Conv2d on x [Cin, H, W], w [Cout, Cin, kh, kw]."""
    cin, H, W = x.shape
    cout, _, kh, kw = w.shape
    xp = np.zeros((cin, H + 2 * padding, W + 2 * padding), dtype=x.dtype)
    xp[:, padding:padding + H, padding:padding + W] = x
    Ho = (H + 2 * padding - kh) // stride + 1
    Wo = (W + 2 * padding - kw) // stride + 1
    y = np.zeros((cout, Ho * Wo), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            seg = xp[:, i: i + (Ho - 1) * stride + 1: stride, j: j + (Wo - 1) * stride + 1: stride]
            y += w[:, :, i, j] @ seg.reshape(cin, Ho * Wo)
    y = y.reshape(cout, Ho, Wo)
    if b is not None:
        y = y + b[:, None, None]
    return y


def leaky_relu(x, slope):
    return np.where(x >= 0, x, x * x.dtype.type(slope))


def relu(x):
    return np.maximum(x, 0)


def layer_norm(x, g, b, eps=1e-5):
    """torch.nn.LayerNorm over the last dim: biased variance, eps inside the sqrt."""
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + x.dtype.type(eps)) * g + b


def instance_norm1d(x, g=None, b=None, eps=1e-5):
    """torch.nn.InstanceNorm1d on x [C, T]: per-channel stats over time, biased var, no running stats."""
    mu = x.mean(axis=1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=1, keepdims=True)
    y = (x - mu) / np.sqrt(var + x.dtype.type(eps))
    if g is not None:
        y = y * g[:, None] + b[:, None]
    return y


def batch_norm_eval(x, p, sd, eps=1e-5):
    """torch BatchNorm (eval): running stats; channel axis 0."""
    shp = (-1,) + (1,) * (x.ndim - 1)
    scale = sd[p + ".weight"] / np.sqrt(sd[p + ".running_var"] + np.float32(eps))
    shift = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return x * scale.reshape(shp).astype(x.dtype) + shift.reshape(shp).astype(x.dtype)


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def fold_wn(sd, prefix):
    """weight_norm(dim=0) fold, w = g*v/||v|| (hifigan.py:132-139; StyleTTS convs never remove it,
    styletts.py:25-34, so the reference recomputes this per call -- same value)."""
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"]
    v = sd[prefix + ".weight_v"].astype(np.float64)
    g = sd[prefix + ".weight_g"].astype(np.float64)
    nrm = np.sqrt((v ** 2).sum(axis=tuple(range(1, v.ndim)), keepdims=True))
    return (g * v / nrm).astype(np.float32)


def _cast(sd, dtype):
    return {k: (v.astype(dtype) if v.dtype.kind == "f" else v) for k, v in sd.items()}


# --------------------------------------------------------------------------------------------
# FastSpeech2 blocks (fs2.py)
# --------------------------------------------------------------------------------------------

def sinusoid_table(n_position, d_hid):
    """fs2.py:17-37."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    tab = pos / np.power(10000.0, 2.0 * (j // 2) / d_hid)[None, :]
    tab[:, 0::2] = np.sin(tab[:, 0::2])
    tab[:, 1::2] = np.cos(tab[:, 1::2])
    return tab.astype(np.float32)


def scln(x, s, w_affine, eps=1e-8):
    """SCLN.forward fs2.py:76-90: unbiased std, (sigma+eps); [b|g] = W s, b = first half (fs2.py:85)."""
    H = x.shape[-1]
    mu = x.mean(axis=-1, keepdims=True)
    sigma = np.sqrt(((x - mu) ** 2).sum(axis=-1, keepdims=True) / (H - 1))   # torch.std: unbiased
    y = (x - mu) / (sigma + x.dtype.type(eps))
    bg = linear(s.reshape(-1), w_affine)
    b, g = bg[:H], bg[H:]
    return g * y + b


def mha(x, sd, p, n_head, spk=None, key_len=None):
    """MultiHeadAttention.forward fs2.py:133-164 with ScaledDotProductAttention fs2.py:47-58.

    x [L, H]; key positions >= key_len are masked with -inf (fs2.py:52-53)."""
    L, H = x.shape
    d = H // n_head
    q = linear(x, sd[p + ".w_qs.weight"], sd[p + ".w_qs.bias"])
    k = linear(x, sd[p + ".w_ks.weight"], sd[p + ".w_ks.bias"])
    v = linear(x, sd[p + ".w_vs.weight"], sd[p + ".w_vs.bias"])
    out = np.empty_like(x)
    temp = x.dtype.type(np.power(d, 0.5))
    for h in range(n_head):
        sl = slice(h * d, (h + 1) * d)
        attn = (q[:, sl] @ k[:, sl].T) / temp
        if key_len is not None and key_len < L:
            attn[:, key_len:] = -np.inf
        out[:, sl] = softmax(attn, axis=1) @ v[:, sl]
    out = linear(out, sd[p + ".fc.weight"], sd[p + ".fc.bias"]) + x
    if spk is not None:
        return scln(out, spk, sd[p + ".layer_norm.affine_layer.linear.weight"])
    return layer_norm(out, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"])


def pos_ffn(x, sd, p, spk=None):
    """PositionwiseFeedForward.forward fs2.py:196-209: conv k9 -> ReLU -> conv k1, +res, norm."""
    w1, w2 = sd[p + ".w_1.weight"], sd[p + ".w_2.weight"]
    o = conv1d(x.T, w1, sd[p + ".w_1.bias"], padding=(w1.shape[2] - 1) // 2)
    o = relu(o)
    o = conv1d(o, w2, sd[p + ".w_2.bias"], padding=(w2.shape[2] - 1) // 2).T + x
    if spk is not None:
        return scln(o, spk, sd[p + ".layer_norm.affine_layer.linear.weight"])
    return layer_norm(o, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"])


def fft_block(x, sd, p, n_head, spk=None):
    """FFTBlock.forward fs2.py:221-230 (mask all-False for batch-1: fills are no-ops)."""
    x = mha(x, sd, p + ".slf_attn", n_head, spk)
    return pos_ffn(x, sd, p + ".pos_ffn", spk)


def encoder(phoneme, puncts, sd, cfg):
    """Encoder.forward fs2.py:370-401."""
    p = "_phoneme_encoder._encoder"
    m = cfg["model"]
    x = np.concatenate([sd[p + ".src_word_emb.weight"][phoneme], sd[p + ".punct_embed.weight"][puncts]], axis=1)
    T, H = x.shape
    if T > m["max_txt_len"]:                      # fs2.py:383-388
        pe = sinusoid_table(T, H).astype(x.dtype)
    else:
        pe = sd[p + ".position_enc"][0, :T]       # fs2.py:390-392
    x = x + pe
    for i in range(m["encoder"]["fs2_layer"]):
        x = fft_block(x, sd, f"{p}.layer_stack.{i}", m["encoder"]["fs2_head"])
    return x


def variance_predictor(x, sd, p):
    """VariancePredictor.forward fs2.py:555-563 (conv k3 -> ReLU -> LN -> conv k3 -> ReLU -> LN -> Linear)."""
    k = sd[p + ".conv_layer.conv1d_1.conv.weight"].shape[2]
    o = conv1d(x.T, sd[p + ".conv_layer.conv1d_1.conv.weight"], sd[p + ".conv_layer.conv1d_1.conv.bias"],
               padding=(k - 1) // 2).T
    o = layer_norm(relu(o), sd[p + ".conv_layer.layer_norm_1.weight"], sd[p + ".conv_layer.layer_norm_1.bias"])
    o = conv1d(o.T, sd[p + ".conv_layer.conv1d_2.conv.weight"], sd[p + ".conv_layer.conv1d_2.conv.bias"],
               padding=1).T                       # fs2.py:543: padding is the literal 1
    o = layer_norm(relu(o), sd[p + ".conv_layer.layer_norm_2.weight"], sd[p + ".conv_layer.layer_norm_2.bias"])
    return linear(o, sd[p + ".linear_layer.weight"], sd[p + ".linear_layer.bias"])[:, 0]


def bucketize(pred, n_bins):
    """fs2.py:639,649: clamp(round(pred*(n_bins-1)).long(), 0, n_bins-1); torch.round = half-to-even."""
    return np.clip(np.rint(pred * pred.dtype.type(n_bins - 1)).astype(np.int64), 0, n_bins - 1)


def length_regulate(x, duration):
    """LengthRegulator.expand fs2.py:447-455: row i repeated max(int(d_i), 0) times."""
    reps = np.maximum(np.asarray(duration).astype(np.int64), 0)
    return np.repeat(x, reps, axis=0)


def variance_adaptor(x, sd, cfg, duration=None):
    """VarianceAdaptor.forward fs2.py:652-693 (inference: targets None, optional forced duration)."""
    p = "_phoneme_encoder._variance_adaptor"
    nb = cfg["model"]["encoder"]["ve_n_bins"]
    log_d = variance_predictor(x, sd, p + ".duration_predictor")
    pitch = variance_predictor(x, sd, p + ".pitch_predictor")
    pitch_idx = bucketize(pitch, nb)
    x = x + sd[p + ".pitch_embedding.weight"][pitch_idx]
    energy = variance_predictor(x, sd, p + ".energy_predictor")      # sees the pitch-embedded x (fs2.py:668-671)
    energy_idx = bucketize(energy, nb)
    x = x + sd[p + ".energy_embedding.weight"][energy_idx]
    if duration is None:
        duration = np.maximum(np.rint(np.exp(log_d) - log_d.dtype.type(1)), 0)   # fs2.py:678-681
    feats = length_regulate(x, duration)
    return dict(features=feats, pitch=pitch, energy=energy, log_duration=log_d, mel_len=feats.shape[0],
                duration=np.asarray(duration).astype(np.int64), pitch_idx=pitch_idx, energy_idx=energy_idx,
                pre_lr=x)


def fs2_encoder(phoneme, puncts, style_embed, sd, cfg, duration=None):
    """FS2Encoder.forward fs2.py:732-775."""
    feats = encoder(phoneme, puncts, sd, cfg)
    feats = feats + style_embed.reshape(1, -1)            # fs2.py:740-741
    out = variance_adaptor(feats, sd, cfg, duration)
    out["encoder_out"] = feats
    return out


def fs2_decoder(x, spk, sd, cfg):
    """FS2Decoder.forward fs2.py:281-315 (batch-1, mask all-False)."""
    p = "_mel_decoder"
    m = cfg["model"]
    L, H = x.shape
    if L > m["max_mel_len"]:                              # fs2.py:287-294
        x = x + sinusoid_table(L, H).astype(x.dtype)
    else:
        x = x + sd[p + ".position_enc"][0, :L]            # fs2.py:296-302
    use_scln = m["decoder"]["scln"]
    for i in range(m["decoder"]["n_layers"]):
        x = fft_block(x, sd, f"{p}.layer_stack.{i}", m["decoder"]["n_head"], spk if use_scln else None)
    return linear(x, sd[p + ".mel_linear.weight"], sd[p + ".mel_linear.bias"])


# --------------------------------------------------------------------------------------------
# StyleTTS decoder (styletts.py)
# --------------------------------------------------------------------------------------------

def _wn_conv(x, sd, p, padding):
    return conv1d(x, fold_wn(sd, p), sd.get(p + ".bias"), padding=padding)


def resblk1d(x, sd, p):
    """ResBlk1d(normalize=True, downsample='none').forward styletts.py:44-69."""
    sc = _wn_conv(x, sd, p + ".conv1x1", 0) if (p + ".conv1x1.weight_v") in sd or (p + ".conv1x1.weight") in sd else x
    r = instance_norm1d(x, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])
    r = _wn_conv(leaky_relu(r, 0.2), sd, p + ".conv1", 1)
    r = instance_norm1d(r, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])
    r = _wn_conv(leaky_relu(r, 0.2), sd, p + ".conv2", 1)
    return (sc + r) / x.dtype.type(math.sqrt(2))


def adain1d(x, s, sd, p):
    """AdaIN1d.forward styletts.py:88-92: (1+gamma)*IN(x)+beta, gamma = first half of fc(s)."""
    h = linear(s.reshape(-1), sd[p + ".fc.weight"], sd[p + ".fc.bias"])
    C = x.shape[0]
    return (1 + h[:C, None]) * instance_norm1d(x) + h[C:, None]


def adain_resblk1d(x, s, sd, p):
    """AdainResBlk1d.forward styletts.py:119-139 (pool = Identity, dropout inactive)."""
    r = leaky_relu(adain1d(x, s, sd, p + ".norm1"), 0.2)
    r = _wn_conv(r, sd, p + ".conv1", 1)
    r = leaky_relu(adain1d(r, s, sd, p + ".norm2"), 0.2)
    r = _wn_conv(r, sd, p + ".conv2", 1)
    sc = _wn_conv(x, sd, p + ".conv1x1", 0) if (p + ".conv1x1.weight_v") in sd or (p + ".conv1x1.weight") in sd else x
    return (r + sc) / x.dtype.type(math.sqrt(2))


def styletts_decoder(feat, spk, sd):
    """StyleTTSDecoder.forward styletts.py:181-205. feat [L, H] -> mel [L, 80]."""
    p = "_mel_decoder"
    e = feat.T
    x = resblk1d(resblk1d(e, sd, p + ".encode.0"), sd, p + ".encode.1")
    asr = instance_norm1d(_wn_conv(e, sd, p + ".asr_res.0", 0), sd[p + ".asr_res.1.weight"], sd[p + ".asr_res.1.bias"])
    res = True
    for i in range(5):
        if res:
            x = np.concatenate([x, asr], axis=0)
        x = adain_resblk1d(x, spk, sd, f"{p}.decode.{i}")
        if i == 2:           # decode.2 is built with upsample=True (styletts.py:156) -> truthy type stops the concat
            res = False
    return _wn_conv(x, sd, p + ".to_out.0", 0).T


# --------------------------------------------------------------------------------------------
# HiFi-GAN generator (hifigan.py)
# --------------------------------------------------------------------------------------------

def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)       # hifigan.py:22-23


def resblock1(x, sd, p, k, dil):
    """ResBlock1.forward hifigan.py:49-56."""
    for t, d in enumerate(dil):
        xt = leaky_relu(x, LRELU_SLOPE)
        xt = conv1d(xt, fold_wn(sd, f"{p}.convs1.{t}"), sd[f"{p}.convs1.{t}.bias"], dilation=d, padding=get_padding(k, d))
        xt = leaky_relu(xt, LRELU_SLOPE)
        xt = conv1d(xt, fold_wn(sd, f"{p}.convs2.{t}"), sd[f"{p}.convs2.{t}.bias"], dilation=1, padding=get_padding(k, 1))
        x = xt + x
    return x


def resblock2(x, sd, p, k, dil):
    """ResBlock2.forward hifigan.py:77-82."""
    for t, d in enumerate(dil):
        xt = leaky_relu(x, LRELU_SLOPE)
        xt = conv1d(xt, fold_wn(sd, f"{p}.convs.{t}"), sd[f"{p}.convs.{t}.bias"], dilation=d, padding=get_padding(k, d))
        x = xt + x
    return x


def hifigan_generator(mel, hsd, h, return_stages=False):
    """Generator.forward hifigan.py:114-130. mel [80, P] -> wav [256*P]."""
    x = conv1d(mel, fold_wn(hsd, "conv_pre"), hsd["conv_pre.bias"], padding=3)
    nk = len(h["resblock_kernel_sizes"])
    rb = resblock1 if h["resblock"] == "1" else resblock2
    stages = []
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = leaky_relu(x, LRELU_SLOPE)
        x = conv_transpose1d(x, fold_wn(hsd, f"ups.{i}"), hsd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j, (rk, rd) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            y = rb(x, hsd, f"resblocks.{i * nk + j}", rk, rd)
            xs = y if xs is None else xs + y
        x = xs / x.dtype.type(nk)
        stages.append(x)
    x = leaky_relu(x, 0.01)                                    # hifigan.py:126: default slope, NOT 0.1
    x = conv1d(x, fold_wn(hsd, "conv_post"), hsd["conv_post.bias"], padding=3)
    wav = np.tanh(x)[0]
    return (wav, stages) if return_stages else wav


# --------------------------------------------------------------------------------------------
# ResNetSE34V2 speaker encoder (ResNetSE34V2.py)
# --------------------------------------------------------------------------------------------

def se_basic_block(x, sd, p, stride):
    """SEBasicBlock.forward ResNetSE34V2.py:83-99: conv -> ReLU -> BN -> conv -> BN -> SE -> +res -> ReLU."""
    out = conv2d(x, sd[p + ".conv1.weight"], stride=stride, padding=1)
    out = batch_norm_eval(relu(out), p + ".bn1", sd)
    out = batch_norm_eval(conv2d(out, sd[p + ".conv2.weight"], padding=1), p + ".bn2", sd)
    y = out.mean(axis=(1, 2))                                                       # SELayer :63-67
    y = relu(linear(y, sd[p + ".se.fc.0.weight"], sd[p + ".se.fc.0.bias"]))
    y = linear(y, sd[p + ".se.fc.2.weight"], sd[p + ".se.fc.2.bias"])
    y = 1.0 / (1.0 + np.exp(-y))
    out = out * y[:, None, None].astype(out.dtype)
    if (p + ".downsample.0.weight") in sd:
        x = batch_norm_eval(conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), p + ".downsample.1", sd)
    return relu(out + x)


def resnet_se34v2(mel, sd, cfg):
    """ResNetSE34V2.forward ResNetSE34V2.py:176-212 (log_input=False, model.py:223). mel [Tr, 80] -> [528]."""
    p = "_spkemb"
    rn = cfg["model"]["resnet"]
    x = instance_norm1d(mel.T)[None]                                     # [1, 80, Tr]
    x = conv2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    x = batch_norm_eval(relu(x), p + ".bn1", sd)
    for li, nblk in enumerate(rn["layers"], start=1):
        for bi in range(nblk):
            x = se_basic_block(x, sd, f"{p}.layer{li}.{bi}", 2 if (li > 1 and bi == 0) else 1)
    C, F, T = x.shape
    x = x.reshape(C * F, T)                                              # :195
    w = conv1d(x, sd[p + ".attention.0.weight"], sd[p + ".attention.0.bias"])
    w = batch_norm_eval(relu(w), p + ".attention.2", sd)
    w = softmax(conv1d(w, sd[p + ".attention.3.weight"], sd[p + ".attention.3.bias"]), axis=1)
    mu = (x * w).sum(axis=1)
    if rn["encoder_type"] == "SAP":
        feat = mu
    elif rn["encoder_type"] == "ASP":
        sg = np.sqrt(np.maximum((x ** 2 * w).sum(axis=1) - mu ** 2, x.dtype.type(1e-5)))
        feat = np.concatenate([mu, sg])
    else:
        raise ValueError("Undefined encoder")                            # ResNetSE34V2.py:143
    e = linear(feat, sd[p + ".fc.weight"], sd[p + ".fc.bias"])
    return e / max(np.sqrt((e ** 2).sum()), 1e-12)                       # F.normalize(p=2, dim=1)


# --------------------------------------------------------------------------------------------
# assembly (model.py / synthesize.py)
# --------------------------------------------------------------------------------------------

def mel_decoder(feats, spk, sd, cfg):
    kind = cfg["model"]["decoder"]["kind"]
    if kind == "fastspeech2":
        return fs2_decoder(feats, spk, sd, cfg)
    if kind == "styletts":
        return styletts_decoder(feats, spk, sd)
    raise Exception(f"unknown decoder kind: '{kind}'")                   # model.py:244


def inference_ex(sd, hsd, cfg, hcfg, phoneme, puncts, style_embed, duration=None, pad_to=689,
                 dtype=np.float32):
    """ZeroVox.inference_ex model.py:308-347 for one utterance.

    ``pad_to`` is the value of the reference's stateful ``_min_mel_len`` at call time (689 for a fresh
    model, model.py:254): mel rows are zero-padded up to it before vocoding (model.py:331-335).
    Returns dict(wav[:mel_len*hop], mel_len, log_duration, mel[80, mel_len], + intermediates).
    """
    sd, hsd = _cast(sd, dtype), _cast(hsd, dtype)
    spk = np.asarray(style_embed, dtype=dtype).reshape(-1)
    hop = cfg["audio"]["hop_size"]
    enc = fs2_encoder(np.asarray(phoneme), np.asarray(puncts), spk, sd, cfg, duration)
    mel = mel_decoder(enc["features"], spk, sd, cfg)                     # [L, 80]
    mel_len = enc["mel_len"]
    P = max(pad_to, mel_len)
    melp = np.zeros((P, mel.shape[1]), dtype=dtype)
    melp[:mel_len] = mel
    wav = hifigan_generator(melp.T, hsd, hcfg)
    return dict(wav=wav[: mel_len * hop], mel_len=mel_len, log_duration=enc["log_duration"], mel=mel.T,
                features=enc["features"], encoder_out=enc["encoder_out"], pitch=enc["pitch"],
                energy=enc["energy"], duration=enc["duration"], pitch_idx=enc["pitch_idx"],
                energy_idx=enc["energy_idx"], padded_len=P)


def transcript2phonemids(transcript, phones, puncts):
    """ZeroVoxTTS.transcript2phonemids synthesize.py:145-190: whitespace/punctuation runs collapse to the
    max punct id, attached to the PREVIOUS phone; unknown characters are skipped."""
    phone2id = {p: i for i, p in enumerate(phones)}
    punct2id = {p: i + 1 for i, p in enumerate(puncts)}
    out_ph, out_pu = [], []
    i, n = 0, len(transcript)
    best = 0          # the reference's `punct`: reset only when a phone is emitted (synthesize.py:185)
    while i < n:
        c = transcript[i]
        if c == " " or c in punct2id:
            while i < n and (transcript[i] == " " or transcript[i] in punct2id):
                best = max(best, punct2id[transcript[i]])
                i += 1
            if out_pu:
                out_pu[-1] = best
            continue
        if c in phone2id:
            best = 0
            out_ph.append(phone2id[c])
            out_pu.append(0)
        i += 1
    return out_ph, out_pu
