"""Host synthesis API: drop-in for ``zerovox.tts.synthesize.ZeroVoxTTS`` (synthesize.py:38-328).

Same method names, argument meaning, return order and sentinel/error behaviour; tensors are NumPy arrays
instead of torch tensors and all model arithmetic runs in libzvx (HIP, gfx950).
"""
from __future__ import annotations

import glob
import os
import time
import wave

import numpy as np

from . import _lib
from .mels import trim_silence
from .model import ZeroVox, load_meldec_weights, load_tts_weights
from .normalize import ZeroVoxNormalizer
from .symbols import Symbols

DEFAULT_TTS_MODEL_NAME_EN = "tts_en_zerovox2_medium_2_styledec"          # synthesize.py:34-36
DEFAULT_TTS_MODEL_NAME_DE = "tts_de_zerovox2_medium_3_styledec"
DEFAULT_REFAUDIO = "en_kevin.wav"


class ZeroVoxTTS:

    @staticmethod
    def get_default_model(lang: str):
        if lang == "en":
            return os.getenv("ZEROVOX_TTS_MODEL_EN", DEFAULT_TTS_MODEL_NAME_EN)
        if lang == "de":
            return os.getenv("ZEROVOX_TTS_MODEL_DE", DEFAULT_TTS_MODEL_NAME_DE)
        return None

    def __init__(self, language, syms: Symbols, model: ZeroVox, meldec_model, hop_length, sampling_rate, n_mel_channels,
                 fft_size, win_length, mel_fmin, mel_fmax, infer_device="cuda", num_threads=-1, verbose=False):
        self._hop_length, self._infer_device, self._sampling_rate = hop_length, infer_device, sampling_rate
        self._language, self._meldec_model = language, meldec_model
        self._fft_size, self._win_length, self._num_mels = fft_size, win_length, n_mel_channels
        self._mel_fmin, self._mel_fmax, self._verbose = mel_fmin, mel_fmax, verbose
        self._model = model
        self._symbols = syms
        self._normalizer = ZeroVoxNormalizer(language)
        # num_threads: the reference sets torch's global intra-op thread count (synthesize.py:93-94);
        # there is no host compute here, the argument is accepted and ignored.

    @staticmethod
    def available_speakerrefs(refdir=None):
        """Bundled reference voices (synthesize.py:99-110).  The build ships no audio assets: lists ``refdir``."""
        if refdir is None or not os.path.isdir(refdir):
            return []
        return sorted((f for f in os.listdir(refdir) if f.endswith(".wav")), key=str.casefold)

    @staticmethod
    def get_speakerref(speakerref, sampling_rate):
        """Load a reference wav (8/16/32-bit PCM) as float32 mono at ``sampling_rate`` (synthesize.py:112-121).  The reference
        goes through ``librosa.load(sr=sampling_rate)``, which resamples with soxr_hq; here a file at another rate is
        resampled with a polyphase Kaiser filter (scipy.signal.resample_poly) -- the same band-limited signal, not
        bit-identical to soxr (parity of resampled references is unpinned: librosa/soxr are not installable here)."""
        with wave.open(str(speakerref), "rb") as w:
            sr, nch, sw = w.getframerate(), w.getnchannels(), w.getsampwidth()
            raw = w.readframes(w.getnframes())
        dt = {1: np.uint8, 2: np.int16, 4: np.int32}[sw]
        a = np.frombuffer(raw, dtype=dt).astype(np.float32)
        a = (a - 128.0) / 128.0 if sw == 1 else a / float(2 ** (8 * sw - 1))
        if nch > 1:
            a = a.reshape(-1, nch).mean(axis=1)
        if sr != sampling_rate:
            from math import gcd
            from scipy.signal import resample_poly
            g = gcd(int(sr), int(sampling_rate))
            a = resample_poly(a.astype(np.float64), int(sampling_rate) // g, int(sr) // g).astype(np.float32)
        return a

    def speaker_embed(self, wav: np.ndarray):
        """wav -> [1, 1, hidden] speaker embedding (synthesize.py:123-143)."""
        wav = trim_silence(wav, top_db=40)
        mel, frames = self._model.ctx.melspec([wav])                      # get_mel_from_wav on the device (zvx_melspec)
        return self._model._spkemb(mel[:, :int(frames[0])])

    def speaker_embed_from_mel(self, mel: np.ndarray):
        """[Tr, n_mels] log-mel -> [1, 1, hidden] (precomputed-mel entry used by the benchmarks)."""
        return self._model._spkemb(np.asarray(mel, np.float32)[None])

    def transcript2phonemids(self, transcript: str):
        """synthesize.py:145-190: whitespace/punctuation runs collapse to the max punct id on the previous phone."""
        phones, puncts = [], []
        punct = 0
        i, n = 0, len(transcript)
        while i < n:
            p = transcript[i]
            if p == " " or self._symbols.is_punct(p):
                while i < n and (transcript[i] == " " or self._symbols.is_punct(transcript[i])):
                    punct = max(punct, self._symbols.encode_punct(transcript[i]))
                    i += 1
                if puncts:
                    puncts[-1] = punct
                continue
            if self._symbols.is_phone(p):
                punct = 0
                phones.append(self._symbols.encode_phone(p))
                puncts.append(punct)
            i += 1
        return phones, puncts

    def text2phonemeids(self, text: str):
        transcript_uroman, _ = self._normalizer.normalize(text)
        phone_ids, punct_ids = self.transcript2phonemids(transcript_uroman)
        if self._verbose:
            print(f"Raw Text Sequence: {text}\nNormalized       : {transcript_uroman}")
            print(f"Phoneme IDs      : {phone_ids}\nPunct IDs        : {punct_ids}")
        return phone_ids, punct_ids

    def tts_ex(self, text: str, spkemb, duration=None):
        """-> (wav f32[N], phoneme i32[1,T], length, mel f32[n_mels, L]); empty text -> the reference's sentinel
        (synthesize.py:213-239)."""
        text = text.strip()
        t0 = time.time()
        phone_ids, punct_ids = self.text2phonemeids(text)
        if not phone_ids:
            return (np.array([[0.0]], dtype=np.float32), np.array([[0]], dtype=np.int32), 0,
                    np.array([[0.0]], dtype=np.float32))
        phoneme = np.array([phone_ids], dtype=np.int32)
        puncts = np.array([punct_ids], dtype=np.int32)
        duration = np.array([duration], dtype=np.int32) if duration is not None else None
        t1 = time.time()
        wav, length, _, mel = self._model.inference_ex({"phoneme": phoneme, "puncts": puncts, "duration": duration},
                                                       style_embed=spkemb, force_duration=duration is not None)
        if self._verbose:
            print(f"tts timing stats: g2p={t1 - t0}s, synth={time.time() - t1}s")
        return wav, phoneme, length, mel

    def tts(self, text: str, spkemb):
        wav, phoneme, length, _ = self.tts_ex(text=text, spkemb=spkemb)
        return wav, phoneme, length

    def tts_stream(self, text: str, spkemb, chunk_frames=64, chunks_per_call=1):
        """Streaming variant of ``tts`` (not in the reference): encoder + mel decoder run once, the vocoder
        runs chunk by chunk (16-frame halo), yielding float32 waveform pieces that concatenate to ``tts(text, spkemb)[0]``
        up to the reference's `_min_mel_len` zero-padding of short utterances."""
        text = text.strip()
        phone_ids, punct_ids = self.text2phonemeids(text)
        if not phone_ids:
            return
        phoneme, puncts = np.array([phone_ids], np.int32), np.array([punct_ids], np.int32)
        ctx = self._model.ctx
        mel_len, _, _, _ = ctx.encode(phoneme, puncts, np.array([len(phone_ids)], np.int32), np.asarray(spkemb, np.float32).reshape(1, -1))
        ml = int(mel_len[0])
        if ml < 2:
            raise ValueError(f"predicted mel length {ml} is too short to synthesise")
        mel = ctx.decode(1, ml)[0, :ml]
        yield from self._model.vocode_stream(mel, chunk_frames=chunk_frames, chunks_per_call=chunks_per_call)

    @property
    def normalizer(self):
        return self._normalizer

    @property
    def language(self):
        return self._normalizer.language

    @language.setter
    def language(self, value):
        if value != self._normalizer.language:
            self._normalizer = ZeroVoxNormalizer(lang=value)

    @property
    def meldec_model(self):
        return self._meldec_model

    @property
    def model(self):
        return self._model

    @classmethod
    def load_model(cls, modelpath, meldec_model, infer_device="cuda", num_threads=-1, verbose=False, precision="bf16"):
        """-> (modelcfg, synth)   (synthesize.py:285-328).  ``modelpath``: directory with modelcfg.yaml +
        weights.npz, or ``synthetic:<decoder_kind>[:seed]``; ``meldec_model``: directory with config.json +
        generator.npz, or ``synthetic:<v1|v2|v3|tiny>[:seed]``."""
        modelcfg, sd = load_tts_weights(modelpath)
        hcfg, hsd = load_meldec_weights(meldec_model, tts_modelpath=modelpath)
        model = ZeroVox(modelcfg, sd, hcfg, hsd, infer_device=infer_device, precision=precision, verbose=verbose)
        a = modelcfg["audio"]
        synth = cls(language=modelcfg["lang"][0], syms=Symbols(modelcfg["model"]["phones"], modelcfg["model"]["puncts"]),
                    model=model, meldec_model=str(meldec_model), hop_length=a["hop_size"], win_length=a["win_length"],
                    mel_fmin=a["fmin"], mel_fmax=a["fmax"], sampling_rate=a["sampling_rate"],
                    n_mel_channels=a["num_mels"], fft_size=a["fft_size"], infer_device=infer_device,
                    num_threads=num_threads, verbose=verbose)
        return modelcfg, synth


def write_wav_to_file(wav, length, filename, sample_rate, hop_length):
    """demo.py:29-35: int16 PCM, x32760, cut to length*hop."""
    pcm = (np.asarray(wav) * 32760).astype("int16")[: length * hop_length]
    with wave.open(str(filename), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sample_rate)
        w.writeframes(pcm.tobytes())
