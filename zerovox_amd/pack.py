"""Weight packer: reference-named state dicts -> the flat fp32 blob + text manifest ``zvx_create`` loads.

What happens here (all host-side NumPy, once per model load):
  * weight-norm folding, w = g*v/||v|| (hifigan.py:132-139; the StyleTTS decoder never removes it,
    styletts.py:25-34) and BatchNorm(eval) -> per-channel scale/shift (ResNetSE34V2.py:75-78);
  * conv weights re-laid out as [tap][Cout][Cin] (K-contiguous "B^T" operand of the conv-GEMM kernel),
    ConvTranspose1d as 3-tap polyphase weights [tap in {-1,0,+1}][phase*Cout][Cin];
  * the speaker encoder's flattened (channel*freq) feature order (ResNetSE34V2.py:195) re-indexed to the
    device's (freq, channel) order.
Tensor kinds: ``w`` = contraction weight stored in the context's precision (bf16 or f32),
``f`` = contraction weight kept in f32 (encoder / variance adaptor / affine generators),
``p`` = f32 parameter vector/table.

Manifest line format:  ``cfg <key> <value>``  |  ``tensor <name> <kind> <ndim> <dims...> <float_offset>``
"""
from __future__ import annotations

import numpy as np

from .weights import fold_weight_norm


class _Blob:
    def __init__(self):
        self.lines = ["zvx_manifest 1"]
        self.chunks = []
        self.off = 0

    def cfg(self, key, value):
        if isinstance(value, (list, tuple)):
            value = ",".join(str(int(v)) for v in value)
        self.lines.append(f"cfg {key} {value}")

    def add(self, name, kind, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        dims = " ".join(str(d) for d in a.shape)
        self.lines.append(f"tensor {name} {kind} {a.ndim} {dims} {self.off}")
        flat = a.reshape(-1)
        pad = (-flat.size) % 4            # keep every tensor 16-byte aligned in the blob
        if pad:
            flat = np.concatenate([flat, np.zeros(pad, np.float32)])
        self.chunks.append(flat)
        self.off += flat.size

    def finish(self):
        return "\n".join(self.lines) + "\n", np.concatenate(self.chunks)


def _wn(sd, p):
    if p + ".weight" in sd:
        return sd[p + ".weight"].astype(np.float32)
    return fold_weight_norm(sd[p + ".weight_g"], sd[p + ".weight_v"])


def _taps(w):
    """[Cout, Cin, K] -> [K][Cout][Cin]"""
    return np.ascontiguousarray(np.transpose(w, (2, 0, 1)))


def _bn(sd, p, eps=1e-5):
    scale = sd[p + ".weight"] / np.sqrt(sd[p + ".running_var"] + np.float32(eps))
    shift = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return scale.astype(np.float32), shift.astype(np.float32)


def _fft_block(bl, out, sd, p, kind, scln):
    wq, wk, wv = sd[p + ".slf_attn.w_qs.weight"], sd[p + ".slf_attn.w_ks.weight"], sd[p + ".slf_attn.w_vs.weight"]
    bl.add(out + ".wqk", kind, np.concatenate([wq, wk], 0)[None])
    bl.add(out + ".bqk", "p", np.concatenate([sd[p + ".slf_attn.w_qs.bias"], sd[p + ".slf_attn.w_ks.bias"]]))
    bl.add(out + ".wv", kind, wv[None])
    bl.add(out + ".bv", "p", sd[p + ".slf_attn.w_vs.bias"])
    bl.add(out + ".wo", kind, sd[p + ".slf_attn.fc.weight"][None])
    bl.add(out + ".bo", "p", sd[p + ".slf_attn.fc.bias"])
    bl.add(out + ".w1", kind, _taps(sd[p + ".pos_ffn.w_1.weight"]))
    bl.add(out + ".b1", "p", sd[p + ".pos_ffn.w_1.bias"])
    bl.add(out + ".w2", kind, _taps(sd[p + ".pos_ffn.w_2.weight"]))
    bl.add(out + ".b2", "p", sd[p + ".pos_ffn.w_2.bias"])
    if not scln:
        for i, nm in ((1, "slf_attn"), (2, "pos_ffn")):
            bl.add(f"{out}.ln{i}_g", "p", sd[f"{p}.{nm}.layer_norm.weight"])
            bl.add(f"{out}.ln{i}_b", "p", sd[f"{p}.{nm}.layer_norm.bias"])


def pack_model(modelcfg: dict, tts_sd: dict, hifigan_cfg: dict, hifigan_sd: dict, precision: str = "bf16"):
    """Returns (manifest_text, blob float32[...])."""
    assert precision in ("bf16", "f32")
    m = modelcfg["model"]
    enc, dec, rn = m["encoder"], m["decoder"], m["resnet"]
    H = m["emb_dim"] + m["punct_emb_dim"]
    n_mels, hop = modelcfg["audio"]["num_mels"], modelcfg["audio"]["hop_size"]
    sd = tts_sd
    bl = _Blob()
    bl.cfg("precision", precision)
    for k, v in (("hidden", H), ("emb_dim", m["emb_dim"]), ("punct_dim", m["punct_emb_dim"]),
                 ("n_phone_rows", len(m["phones"]) + 1), ("n_punct_rows", len(m["puncts"]) + 2),
                 ("max_txt_len", m["max_txt_len"]), ("max_mel_len", m["max_mel_len"]),
                 ("enc_layers", enc["fs2_layer"]), ("enc_heads", enc["fs2_head"]),
                 ("ffn_dim", dec["conv_filter_size"]), ("ffn_k", dec["conv_kernel_size"]),
                 ("vp_dim", enc["vp_filter_size"]), ("vp_k", enc["vp_kernel_size"]), ("n_bins", enc["ve_n_bins"]),
                 ("dec_layers", dec["n_layers"]), ("dec_heads", dec["n_head"]), ("dec_scln", int(bool(dec["scln"]))),
                 ("n_mels", n_mels), ("hop", hop), ("res_dim", 64),
                 ("rn_layers", rn["layers"]), ("rn_filters", rn["num_filters"])):
        bl.cfg(k, v)
    if dec["kind"] not in ("fastspeech2", "styletts"):
        raise Exception(f"unknown decoder kind: '{dec['kind']}'")           # model.py:244
    bl.cfg("dec_kind", dec["kind"])
    if rn["encoder_type"] not in ("ASP", "SAP"):
        raise ValueError("Undefined encoder")                                # ResNetSE34V2.py:143
    bl.cfg("rn_asp", int(rn["encoder_type"] == "ASP"))

    # ---------------- log-mel front end of the reference audio (mels.py:357-395), always f32 ----------------
    # |STFT| as one GEMM: rows = frames (hop-strided views of the reflect-padded signal), columns = the windowed DFT basis
    # (cos rows, then -sin rows), followed by the Slaney mel basis as a second GEMM.
    a = modelcfg["audio"]
    n_fft, win_len = int(a["fft_size"]), int(a.get("win_length", a["fft_size"]))
    nf = n_fft // 2 + 1
    from .mels import mel_filterbank, stft_basis
    dft = stft_basis(n_fft, win_len)
    basis = np.zeros((n_mels, (nf + 3) // 4 * 4), np.float32)
    basis[:, :nf] = mel_filterbank(a["sampling_rate"], n_fft, n_mels, a["fmin"], a["fmax"])
    for k, v in (("fft_size", n_fft), ("sampling_rate", a["sampling_rate"])):
        bl.cfg(k, v)
    bl.add("mel.dft", "f", dft.astype(np.float32)[None])
    bl.add("mel.basis", "f", basis[None])

    # ---------------- phoneme encoder (always f32: it feeds discrete decisions) ----------------
    pe = "_phoneme_encoder._encoder"
    bl.add("enc.emb", "p", sd[pe + ".src_word_emb.weight"])
    bl.add("enc.pemb", "p", sd[pe + ".punct_embed.weight"])
    bl.add("enc.pe", "p", sd[pe + ".position_enc"][0])
    for i in range(enc["fs2_layer"]):
        _fft_block(bl, f"enc.{i}", sd, f"{pe}.layer_stack.{i}", "f", False)
    va = "_phoneme_encoder._variance_adaptor"
    for short, full in (("dur", "duration_predictor"), ("pitch", "pitch_predictor"), ("energy", "energy_predictor")):
        p = f"{va}.{full}"
        bl.add(f"va.{short}.c1", "f", _taps(sd[p + ".conv_layer.conv1d_1.conv.weight"]))
        bl.add(f"va.{short}.b1", "p", sd[p + ".conv_layer.conv1d_1.conv.bias"])
        bl.add(f"va.{short}.ln1_g", "p", sd[p + ".conv_layer.layer_norm_1.weight"])
        bl.add(f"va.{short}.ln1_b", "p", sd[p + ".conv_layer.layer_norm_1.bias"])
        bl.add(f"va.{short}.c2", "f", _taps(sd[p + ".conv_layer.conv1d_2.conv.weight"]))
        bl.add(f"va.{short}.b2", "p", sd[p + ".conv_layer.conv1d_2.conv.bias"])
        bl.add(f"va.{short}.ln2_g", "p", sd[p + ".conv_layer.layer_norm_2.weight"])
        bl.add(f"va.{short}.ln2_b", "p", sd[p + ".conv_layer.layer_norm_2.bias"])
        bl.add(f"va.{short}.lw", "p", sd[p + ".linear_layer.weight"][0])
        bl.add(f"va.{short}.lb", "p", sd[p + ".linear_layer.bias"])
    bl.add("va.pitch_emb", "p", sd[va + ".pitch_embedding.weight"])
    bl.add("va.energy_emb", "p", sd[va + ".energy_embedding.weight"])

    # ---------------- mel decoder ----------------
    md = "_mel_decoder"
    if dec["kind"] == "fastspeech2":
        bl.add("dec.pe", "p", sd[md + ".position_enc"][0])
        affines = []
        for i in range(dec["n_layers"]):
            _fft_block(bl, f"dec.{i}", sd, f"{md}.layer_stack.{i}", "w", dec["scln"])
            if dec["scln"]:
                affines.append(sd[f"{md}.layer_stack.{i}.slf_attn.layer_norm.affine_layer.linear.weight"])
                affines.append(sd[f"{md}.layer_stack.{i}.pos_ffn.layer_norm.affine_layer.linear.weight"])
        if dec["scln"]:
            bl.add("dec.scln_all", "f", np.concatenate(affines, 0)[None])      # [1][2*layers*2H][H]
        bl.add("dec.mel_w", "w", sd[md + ".mel_linear.weight"][None])
        bl.add("dec.mel_b", "p", sd[md + ".mel_linear.bias"])
    else:
        adain_w, adain_b = [], []

        def conv(name, p, bias=True):
            bl.add(name, "w", _taps(_wn(sd, p)))
            if bias:
                bl.add(name + "_b", "p", sd[p + ".bias"])

        for i in range(2):
            p = f"{md}.encode.{i}"
            for n in ("norm1", "norm2"):
                bl.add(f"sty.enc{i}.{n}_g", "p", sd[f"{p}.{n}.weight"])
                bl.add(f"sty.enc{i}.{n}_b", "p", sd[f"{p}.{n}.bias"])
            conv(f"sty.enc{i}.c1", p + ".conv1")
            conv(f"sty.enc{i}.c2", p + ".conv2")
            if (p + ".conv1x1.weight_v") in sd or (p + ".conv1x1.weight") in sd:
                conv(f"sty.enc{i}.sc", p + ".conv1x1", bias=False)
        conv("sty.asr", md + ".asr_res.0")
        bl.add("sty.asr_g", "p", sd[md + ".asr_res.1.weight"])
        bl.add("sty.asr_beta", "p", sd[md + ".asr_res.1.bias"])
        for i in range(5):
            p = f"{md}.decode.{i}"
            conv(f"sty.dec{i}.c1", p + ".conv1")
            conv(f"sty.dec{i}.c2", p + ".conv2")
            if (p + ".conv1x1.weight_v") in sd or (p + ".conv1x1.weight") in sd:
                conv(f"sty.dec{i}.sc", p + ".conv1x1", bias=False)
            for n in ("norm1", "norm2"):
                adain_w.append(sd[f"{p}.{n}.fc.weight"])
                adain_b.append(sd[f"{p}.{n}.fc.bias"])
        bl.add("sty.adain_w", "f", np.concatenate(adain_w, 0)[None])
        bl.add("sty.adain_b", "p", np.concatenate(adain_b, 0))
        conv("sty.out", md + ".to_out.0")

    # ---------------- speaker encoder ----------------
    sp = "_spkemb"
    nf, layers = rn["num_filters"], rn["layers"]
    w = sd[sp + ".conv1.weight"]                                # [C0,1,3,3]
    bl.add("spk.c1_w", "p", np.transpose(w[:, 0], (1, 2, 0)).reshape(9, nf[0]))
    bl.add("spk.c1_b", "p", sd[sp + ".conv1.bias"])
    s, t = _bn(sd, sp + ".bn1")
    bl.add("spk.bn1_s", "p", s)
    bl.add("spk.bn1_t", "p", t)

    def conv2d_taps(wt):       # [Cout, Cin, kh, kw] -> [kh*kw][Cout][Cin]
        return np.ascontiguousarray(np.transpose(wt, (2, 3, 0, 1)).reshape(wt.shape[2] * wt.shape[3], wt.shape[0], wt.shape[1]))

    for li, nblk in enumerate(layers, start=1):
        for bi in range(nblk):
            p, o = f"{sp}.layer{li}.{bi}", f"spk.l{li}.{bi}"
            bl.add(o + ".c1", "w", conv2d_taps(sd[p + ".conv1.weight"]))
            s, t = _bn(sd, p + ".bn1")                         # applied AFTER the ReLU (ResNetSE34V2.py:86-88)
            bl.add(o + ".bn1_s", "p", s)
            bl.add(o + ".bn1_t", "p", t)
            s, t = _bn(sd, p + ".bn2")                         # conv2 -> BN2 folds exactly
            bl.add(o + ".c2", "w", conv2d_taps(sd[p + ".conv2.weight"] * s[:, None, None, None]))
            bl.add(o + ".c2_b", "p", t)
            bl.add(o + ".se_w1", "p", sd[p + ".se.fc.0.weight"])
            bl.add(o + ".se_b1", "p", sd[p + ".se.fc.0.bias"])
            bl.add(o + ".se_w2", "p", sd[p + ".se.fc.2.weight"])
            bl.add(o + ".se_b2", "p", sd[p + ".se.fc.2.bias"])
            if (p + ".downsample.0.weight") in sd:
                s, t = _bn(sd, p + ".downsample.1")
                bl.add(o + ".ds", "w", conv2d_taps(sd[p + ".downsample.0.weight"] * s[:, None, None, None]))
                bl.add(o + ".ds_b", "p", t)
    C4, Fp = nf[3], n_mels // 8
    D = C4 * Fp
    # reference feature index c*Fp+f (reshape at ResNetSE34V2.py:195) -> device index f*C4+c
    perm = (np.arange(C4)[None, :] * Fp + np.arange(Fp)[:, None]).reshape(-1)      # perm[f*C4+c] = c*Fp+f
    a0 = sd[sp + ".attention.0.weight"][:, :, 0]               # [128, D]
    bl.add("spk.att1_w", "w", np.ascontiguousarray(a0[:, perm].reshape(128, Fp, C4).transpose(1, 0, 2)))   # [f][128][C4]
    bl.add("spk.att1_b", "p", sd[sp + ".attention.0.bias"])
    s, t = _bn(sd, sp + ".attention.2")
    bl.add("spk.att_bn_s", "p", s)
    bl.add("spk.att_bn_t", "p", t)
    a3 = sd[sp + ".attention.3.weight"][:, :, 0]               # [D, 128]
    bl.add("spk.att2_w", "w", a3[perm][None])
    bl.add("spk.att2_b", "p", sd[sp + ".attention.3.bias"][perm])
    fcw = sd[sp + ".fc.weight"]                                # [H, 2D] (ASP) or [H, D]
    if rn["encoder_type"] == "ASP":
        fcw = np.concatenate([fcw[:, :D][:, perm], fcw[:, D:][:, perm]], axis=1)
    else:
        fcw = fcw[:, perm]
    bl.add("spk.fc_w", "f", fcw[None])
    bl.add("spk.fc_b", "p", sd[sp + ".fc.bias"])

    # ---------------- HiFi-GAN generator ----------------
    h, hsd = hifigan_cfg, hifigan_sd
    C0 = h["upsample_initial_channel"]
    rates, ksz = h["upsample_rates"], h["upsample_kernel_sizes"]
    assert int(np.prod(rates)) == hop, "prod(upsample_rates) must equal hop (model.py:347)"
    bl.cfg("voc_resblock", h["resblock"])
    bl.cfg("voc_rates", rates)
    bl.cfg("voc_ksizes", ksz)
    bl.cfg("voc_c0", C0)
    bl.cfg("voc_rb_k", h["resblock_kernel_sizes"])
    bl.cfg("voc_rb_d", ";".join(",".join(str(x) for x in d) for d in h["resblock_dilation_sizes"]))
    bl.add("voc.pre_w", "w", _taps(_wn(hsd, "conv_pre")))
    bl.add("voc.pre_b", "p", hsd["conv_pre.bias"])
    nk = len(h["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(rates, ksz)):
        wt = _wn(hsd, f"ups.{i}")                              # [Cin, Cout, k]
        cin, cout, _ = wt.shape
        pad = (k - u) // 2
        # y[t*u+p] = sum_i x[i] w[:, :, p + pad + (t-i)*u]; taps m = t-i in {-1,0,+1} -> x row t-m ... stored
        # as row offsets dv in {-1,0,+1}: dv = -m
        poly = np.zeros((3, u * cout, cin), np.float32)
        for ti, dv in enumerate((-1, 0, 1)):
            mm = -dv
            for ph in range(u):
                kk = ph + pad + mm * u
                if 0 <= kk < k:
                    poly[ti, ph * cout:(ph + 1) * cout, :] = wt[:, :, kk].T
        assert abs(np.abs(poly).sum() - np.abs(wt).sum()) < 1e-3 * max(1.0, np.abs(wt).sum()), "polyphase taps must cover the kernel"
        bl.add(f"voc.up{i}_w", "w", poly)
        bl.add(f"voc.up{i}_b", "p", np.tile(hsd[f"ups.{i}.bias"], u))
        if k == 2 * u and u % 2 == 0 and cin >= 256:
            # k = 2u: phases [0, u/2) only touch rows (t-1, t), phases [u/2, u) only rows (t, t+1) -- two 2-tap GEMMs over
            # half the output columns each do 2/3 of the 3-tap form's MFMA work (worth it where the GEMM is MFMA-bound)
            hc = (u // 2) * cout
            assert not poly[2, :hc].any() and not poly[0, hc:].any()
            bl.add(f"voc.up{i}_wlo", "w", np.ascontiguousarray(poly[0:2, :hc]))
            bl.add(f"voc.up{i}_whi", "w", np.ascontiguousarray(poly[1:3, hc:]))
        for j in range(nk):
            p = f"resblocks.{i * nk + j}"
            nd = len(h["resblock_dilation_sizes"][j])
            for t in range(nd):
                if h["resblock"] == "1":
                    bl.add(f"voc.rb{i * nk + j}.c1_{t}_w", "w", _taps(_wn(hsd, f"{p}.convs1.{t}")))
                    bl.add(f"voc.rb{i * nk + j}.c1_{t}_b", "p", hsd[f"{p}.convs1.{t}.bias"])
                    bl.add(f"voc.rb{i * nk + j}.c2_{t}_w", "w", _taps(_wn(hsd, f"{p}.convs2.{t}")))
                    bl.add(f"voc.rb{i * nk + j}.c2_{t}_b", "p", hsd[f"{p}.convs2.{t}.bias"])
                else:
                    bl.add(f"voc.rb{i * nk + j}.c_{t}_w", "w", _taps(_wn(hsd, f"{p}.convs.{t}")))
                    bl.add(f"voc.rb{i * nk + j}.c_{t}_b", "p", hsd[f"{p}.convs.{t}.bias"])
    wp = _wn(hsd, "conv_post")                                  # [1, C, 7]
    bl.add("voc.post_w", "p", np.ascontiguousarray(wp[0].T))   # [7][C]
    bl.add("voc.post_b", "p", hsd["conv_post.bias"])
    return bl.finish()
