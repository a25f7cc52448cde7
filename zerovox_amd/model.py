"""Model-level host mirror of the reference's ``ZeroVox`` inference API (model.py:308-351).

``ZeroVox.inference_ex`` / ``inference`` keep the reference's argument meaning and return order; the
computation happens in libzvx on an MI355X.  There is no CPU path: constructing the model without the
HIP library or a GPU raises.
"""
from __future__ import annotations

import json
import os

import numpy as np
import yaml

from . import _lib, config as zcfg, pack, weights as zw

DEFAULT_MELDEC_MODEL_NAME = "zerovox-hifigan-vctk-v2-en-1"      # model.py:84


def parse_device(infer_device) -> int:
    """'cuda' | 'cuda:N' | 'hip:N' | int -> HIP device index.  'cpu' is refused: the product path is GPU-only."""
    if isinstance(infer_device, int):
        return infer_device
    s = str(infer_device)
    if s == "cpu":
        raise _lib.ZvxError(_lib.ZVX_E_UNSUPPORTED, "zerovox_amd has no CPU path; use infer_device='cuda[:N]'")
    if ":" in s:
        return int(s.split(":")[1])
    return 0


def load_tts_weights(modelpath):
    """-> (modelcfg, state_dict).  ``synthetic:<styletts|fastspeech2>[:seed]``, or a directory holding ``modelcfg.yaml``
    (synthesize.py:310-326) and either ``weights.npz`` (tools/convert_checkpoint.py output, torch-free) or the reference's
    own ``checkpoints/*.ckpt`` / ``checkpoint.pkl`` (synthesize.py:295-304; read with torch through zerovox_amd.convert)."""
    spec = str(modelpath)
    if spec.startswith("synthetic:"):
        parts = spec.split(":")
        cfg = zcfg.medium_modelcfg(parts[1])
        return cfg, zw.tts_state_dict(cfg, int(parts[2]) if len(parts) > 2 else 0)
    with open(os.path.join(spec, "modelcfg.yaml")) as f:
        cfg = yaml.load(f, Loader=yaml.FullLoader)
    if os.path.exists(os.path.join(spec, "weights.npz")):
        return cfg, dict(np.load(os.path.join(spec, "weights.npz")))
    from .convert import read_tts_checkpoint          # a reference model directory as downloaded: needs torch to unpickle
    sd, _ = read_tts_checkpoint(spec)
    return cfg, sd


def load_meldec_weights(modelspec, tts_modelpath=None):
    """-> (hifigan_cfg, state_dict).  ``synthetic:<v1|v2|v3|tiny|tiny2>[:seed]`` or a directory with
    ``config.json`` (model.py:90-105) + ``generator.npz`` (converted) or the reference's ``generator.ckpt``.

    A vocoder BAKED INTO the TTS checkpoint (``_meldec.*`` keys, utils/edit_meldec_in_checkpoint.py:77-90; split off
    by tools/convert_checkpoint.py into ``<tts_modelpath>/generator.npz``) overrides the external weights, exactly as
    ``ZeroVox.load_from_checkpoint(strict=False)`` loads those keys on top of ``get_meldec()`` (synthesize.py:78-91);
    the external model then only supplies ``config.json``.  A baked-in state dict whose keys/shapes do not fit that
    config is an error, not a silent fallback."""
    spec = str(modelspec)
    if spec.startswith("synthetic:"):
        parts = spec.split(":")
        h = zcfg.hifigan_config(parts[1])
        hsd = zw.hifigan_state_dict(h, int(parts[2]) if len(parts) > 2 else 0)
    else:
        with open(os.path.join(spec, "config.json")) as f:
            h = json.load(f)
        if os.path.exists(os.path.join(spec, "generator.npz")):
            hsd = dict(np.load(os.path.join(spec, "generator.npz")))
        else:                                           # the reference's own layout: generator.ckpt next to config.json (model.py:90-111)
            from .convert import read_generator_checkpoint
            hsd = read_generator_checkpoint(os.path.join(spec, "generator.ckpt"))
    bsd = None
    if tts_modelpath is not None and not str(tts_modelpath).startswith("synthetic:"):
        baked = os.path.join(str(tts_modelpath), "generator.npz")
        if os.path.exists(baked):
            bsd = dict(np.load(baked))
        elif not os.path.exists(os.path.join(str(tts_modelpath), "weights.npz")):
            from .convert import find_tts_checkpoint, read_tts_checkpoint
            if find_tts_checkpoint(str(tts_modelpath)):
                bsd = read_tts_checkpoint(str(tts_modelpath))[1] or None
                baked = find_tts_checkpoint(str(tts_modelpath))
    if bsd:
        # The reference bakes the vocoder in AFTER remove_weight_norm() (model.py:115, 247; edit_meldec_in_checkpoint.py:77-84
        # copies meldec.state_dict()): a Lightning / edited checkpoint holds plain `.weight` tensors, generator.ckpt holds
        # weight_g / weight_v pairs.  Compare the two in the folded domain (pack._wn accepts either form).
        fb, fh = zw.folded(bsd), zw.folded(hsd)
        bad = sorted(k for k in set(fb) | set(fh) if k not in fb or k not in fh or fb[k].shape != fh[k].shape)
        if bad:
            raise ValueError(f"{baked}: baked-in vocoder does not match {spec}/config.json (first mismatching keys: {bad[:4]})")
        hsd = bsd
    return h, hsd


class ZeroVox:
    """GPU-resident model: phoneme encoder, speaker encoder, mel decoder and HiFi-GAN (model.py:206-249)."""

    def __init__(self, modelcfg, tts_sd, meldec_cfg, meldec_sd, infer_device="cuda", precision="bf16", verbose=False):
        self._modelcfg = modelcfg
        self._hop_length = modelcfg["audio"]["hop_size"]
        self._verbose = verbose
        manifest, blob = pack.pack_model(modelcfg, tts_sd, meldec_cfg, meldec_sd, precision)
        self._packed, self._device = (manifest, blob), parse_device(infer_device)
        self._ctx = _lib.Context(manifest, blob, self._device)
        self._more_ctx = []                             # further contexts of the same model (synthesize_batches; release_contexts() frees them)
        self._streaming = False                         # a synthesize_batches generator is live: self._ctx belongs to its worker threads
        self._min_mel_len = 689                         # model.py:254 -- stateful, see inference_ex
        self.hidden = self._ctx.hidden

    @property
    def ctx(self):
        return self._ctx

    def _spkemb(self, x):
        """ResNetSE34V2.forward: x [B, Tr, 80] -> [B, 1, hidden] (ResNetSE34V2.py:176-212)."""
        x = np.asarray(x, np.float32)
        lens = np.full(x.shape[0], x.shape[1], np.int32)
        return self._ctx.spkemb(x, lens)[:, None, :]

    def inference_ex(self, x, style_embed, normalize_before=True, force_duration=False):
        """model.py:308-347.  x = {"phoneme" [1,T], "puncts" [1,T], "duration" [1,T]|None}; returns
        (wav[:mel_len*hop], mel_len, log_duration [1,T], mel [n_mels, mel_len]).  Batch-1 like the reference."""
        phoneme = np.asarray(x["phoneme"], np.int32)
        puncts = np.asarray(x["puncts"], np.int32)
        if phoneme.ndim != 2 or phoneme.shape[0] != 1:
            raise ValueError("inference_ex is batch-1 (model.py:325); use synthesize_batch for batches")
        T = np.array([phoneme.shape[1]], np.int32)
        dur = np.asarray(x["duration"], np.int32) if (force_duration and x.get("duration") is not None) else None
        spk = np.asarray(style_embed, np.float32).reshape(1, -1)
        mel_len, logd, _, _ = self._ctx.encode(phoneme, puncts, T, spk, dur)
        ml = int(mel_len[0])
        if ml < 2:
            # the reference raises inside InstanceNorm1d / conv stacks for degenerate lengths
            raise ValueError(f"predicted mel length {ml} is too short to synthesise")
        mel = self._ctx.decode(1, ml)
        pad_to = self._min_mel_len                       # model.py:331-335: pad up, or raise the floor
        if ml > self._min_mel_len:
            self._min_mel_len = ml
        wav = self._ctx.vocode(1, mel_len, np.array([pad_to], np.int32))
        return wav[0, : ml * self._hop_length], ml, logd, np.ascontiguousarray(mel[0, :ml].T)

    def inference(self, x, style_embed, normalize_before=True):
        wav, mel_len, log_duration, _ = self.inference_ex(x=x, style_embed=style_embed, normalize_before=normalize_before)
        return wav, mel_len, log_duration

    # HiFi-GAN's receptive field in mel frames (conv_pre k7: 3; upsample stacks: ~1; ResBlock k=11 dilations 1/3/5 at 8
    # samples per frame: 60 / 8 = 7.5; later stages < 2 in total): SURVEY.md 8(f-4) quotes 9-13.  A halo of 16 frames on
    # each side makes a chunk's interior identical to the same samples of a whole-utterance pass.
    STREAM_HALO = 16

    def vocode_stream(self, mel, chunk_frames=64, halo=STREAM_HALO, chunks_per_call=1):
        """Chunked vocoding for first-audio latency: mel [L, n_mels] -> yields waveform chunks (np.float32) that
        concatenate to ``vocode_mel(mel)``.  Every chunk is vocoded with ``halo`` extra frames on each side and only its
        interior is kept; ``chunks_per_call`` chunks ride in one launch sequence as independent batch rows."""
        mel = np.asarray(mel, np.float32)
        L, hop = mel.shape[0], self._hop_length
        starts = list(range(0, L, chunk_frames))
        for g in range(0, len(starts), max(1, chunks_per_call)):
            grp = starts[g:g + max(1, chunks_per_call)]
            spans = [(max(0, s - halo), min(L, s + chunk_frames + halo)) for s in grp]
            P = np.array([hi - lo for lo, hi in spans], np.int32)
            batch = np.zeros((len(grp), int(P.max()), mel.shape[1]), np.float32)
            for i, (lo, hi) in enumerate(spans):
                batch[i, :hi - lo] = mel[lo:hi]
            wav = self._ctx.vocode_mel(batch, P)
            for i, s in enumerate(grp):
                lo = spans[i][0]
                n = min(chunk_frames, L - s)
                yield wav[i, (s - lo) * hop:(s - lo + n) * hop].copy()

    def synthesize_batches(self, batches, in_flight=2, want_mel=False):
        """Throughput mode for a stream of batches: `in_flight` contexts of this model (one stream and one set of work buffers
        each; the first call creates them, +~0.5 GB of weights apiece), batches alternating between them from worker threads
        (the C calls release the GIL), so that batch i + 1's encoder / decoder -- latency-paced launches -- run under batch i's
        vocoder, which holds the chip at its power limit.  Measured at 32 x 128 phonemes with the waveforms delivered to host memory
        (tools/two_contexts.py): 24.5 ms per batch sequentially, 23.5 with two in flight, 22.5 with three (device-resident outputs
        through the C-ABI: 23.4 -> 21.4 with two).  `batches` yields dicts of synthesize_batch arguments (phoneme, puncts, T, style_embed[, duration, pad_to,
        Lmax_cap]); results come back IN ORDER, each bit-identical to synthesize_batch on the same arguments."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        n = max(1, int(in_flight))
        if self._streaming:
            raise RuntimeError("a synthesize_batches generator of this model is still active: its contexts are in use "
                               "(exhaust or close() it first; a context is not re-entrant, include/zvx.h)")
        while len(self._more_ctx) < n - 1:
            self._more_ctx.append(_lib.Context(self._packed[0], self._packed[1], self._device))
        ctxs = [self._ctx] + self._more_ctx[:n - 1]
        self._streaming = True
        try:
            yield from self._run_batches(ctxs, batches, n, want_mel)
        finally:
            self._streaming = False

    def release_contexts(self):
        """Free the extra contexts synthesize_batches created (weights + work buffers, ~0.5 GB each); the next call re-creates them."""
        if self._streaming:
            raise RuntimeError("a synthesize_batches generator is still active")
        for c in self._more_ctx:
            c.close()
        self._more_ctx = []

    def close(self):
        """Release every context of the model (device memory, streams).  The object is unusable afterwards."""
        self.release_contexts()
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    def _run_batches(self, ctxs, batches, n, want_mel):
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor

        def run(c, kw):
            B = np.asarray(kw["phoneme"]).shape[0]
            pad_to = kw.get("pad_to")
            if pad_to is None:
                pad_to = np.full(B, 689, np.int32)
            return c.synthesize(kw["phoneme"], kw["puncts"], kw["T"], kw["style_embed"], kw.get("duration"), pad_to, want_mel, kw.get("Lmax_cap", 0))

        with ThreadPoolExecutor(max_workers=n) as ex:
            pending = deque()
            for i, kw in enumerate(batches):
                if len(pending) >= n:                   # context i % n is the one whose batch i - n is the oldest pending
                    yield pending.popleft().result()
                pending.append(ex.submit(run, ctxs[i % n], kw))
            while pending:
                yield pending.popleft().result()

    def synthesize_batch(self, phoneme, puncts, T, style_embed, duration=None, pad_to=None, want_mel=True, Lmax_cap=0):
        """B independent utterances in one launch sequence; each equals a batch-1 ``inference_ex`` call with
        ``_min_mel_len == pad_to[b]`` (default: the fresh-model value 689).  Padded [B, Tmax] id arrays."""
        B = np.asarray(phoneme).shape[0]
        if pad_to is None:
            pad_to = np.full(B, 689, np.int32)
        if self._streaming:
            raise RuntimeError("synthesize_batch while a synthesize_batches generator is active: the context is not re-entrant")
        return self._ctx.synthesize(phoneme, puncts, T, style_embed, duration, pad_to, want_mel, Lmax_cap)
