"""ctypes binding of libzvx (include/zvx.h).  No fallback: if the HIP library or a GPU is missing,
every entry point raises -- the product path never routes through a CPU implementation."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libzvx.so")

ZVX_OK = 0
ZVX_E_INVALID, ZVX_E_MANIFEST, ZVX_E_HIP, ZVX_E_STATE, ZVX_E_BUFFER, ZVX_E_UNSUPPORTED = 1, 2, 3, 4, 5, 6
ZVX_DEVICE_OUT, ZVX_NO_SYNC, ZVX_PCM16, ZVX_DEVICE_IN, ZVX_HOST_ASYNC = 1, 2, 4, 8, 16
STAGES = ("encoder", "variance", "lenreg", "decoder", "vocoder", "spkemb")
ZVX_T_COUNT = 8

EXPORTS = ("zvx_create", "zvx_destroy", "zvx_last_error", "zvx_get_int", "zvx_set_int", "zvx_spkemb", "zvx_melspec", "zvx_encode",
           "zvx_decode", "zvx_decode_features", "zvx_vocode", "zvx_vocode_mel", "zvx_synthesize", "zvx_fetch",
           "zvx_sync", "zvx_stage_times", "zvx_kernel_stats", "zvx_tag_stats", "zvx_reset_stats",
           "zvx_comm_unique_id", "zvx_comm_init", "zvx_comm_gather", "zvx_comm_barrier", "zvx_comm_max_f64", "zvx_comm_info", "zvx_comm_destroy",
           "zvx_dev_alloc", "zvx_dev_free", "zvx_dev_from_host", "zvx_dev_to_host", "zvx_spkemb_ex", "zvx_wait_host")
ZVX_COMM_ID_BYTES = 128


class ZvxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"zvx error {code}: {msg}")
        self.code = code


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


_lib = None


def load():
    """dlopen libzvx.so (built in-tree by zerovox_amd.build).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ZvxError(ZVX_E_HIP, f"{LIB_PATH} not found: build it with `python -m zerovox_amd.build` "
                                  f"(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    vp, i32p, f32p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)
    lib.zvx_create.argtypes = [C.c_char_p, vp, C.c_size_t, C.c_int, C.POINTER(vp)]
    lib.zvx_destroy.argtypes = [vp]
    lib.zvx_destroy.restype = None
    lib.zvx_last_error.argtypes = [vp]
    lib.zvx_last_error.restype = C.c_char_p
    lib.zvx_get_int.argtypes = [vp, C.c_char_p]
    lib.zvx_get_int.restype = C.c_int64
    lib.zvx_set_int.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.zvx_spkemb.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp]
    lib.zvx_melspec.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp]
    lib.zvx_encode.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    lib.zvx_decode.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.zvx_decode_features.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int]
    lib.zvx_vocode.argtypes = [vp, vp, vp, C.c_int64, C.c_int]
    lib.zvx_vocode_mel.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int64, C.c_int]
    lib.zvx_synthesize.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_int64, vp, vp,
                                   C.c_int, vp, C.c_int]
    lib.zvx_fetch.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    lib.zvx_sync.argtypes = [vp]
    lib.zvx_stage_times.argtypes = [vp, vp]
    lib.zvx_kernel_stats.argtypes = [vp, C.POINTER(KernelStat), C.c_int]
    lib.zvx_reset_stats.argtypes = [vp]
    lib.zvx_tag_stats.argtypes = [vp, C.POINTER(KernelStat), C.c_int]
    lib.zvx_comm_unique_id.argtypes = [vp]
    lib.zvx_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.zvx_comm_gather.argtypes = [vp, vp, C.c_size_t, vp, C.c_int, C.c_int]
    lib.zvx_comm_barrier.argtypes = [vp]
    lib.zvx_comm_max_f64.argtypes = [vp, C.POINTER(C.c_double)]
    lib.zvx_comm_info.argtypes = [vp, C.POINTER(C.c_int64), C.c_int]
    lib.zvx_comm_destroy.argtypes = [vp]
    lib.zvx_comm_destroy.restype = None
    lib.zvx_dev_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.zvx_dev_free.argtypes = [vp, vp]
    lib.zvx_dev_to_host.argtypes = [vp, vp, vp, C.c_size_t]
    lib.zvx_dev_from_host.argtypes = [vp, vp, vp, C.c_size_t]
    lib.zvx_spkemb_ex.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int]
    lib.zvx_wait_host.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.int32)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Context:
    """One synthesis context on one HIP device (wraps zvx_ctx*)."""

    def __init__(self, manifest: str, blob: np.ndarray, device: int = 0):
        self._lib = load()
        self._h = C.c_void_p()
        blob = _f32(blob)
        rc = self._lib.zvx_create(manifest.encode(), _ptr(blob), blob.nbytes, device, C.byref(self._h))
        if rc != ZVX_OK:
            raise ZvxError(rc, self._lib.zvx_last_error(None).decode())
        self.hidden = self.get_int("hidden")
        self.n_mels = self.get_int("n_mels")
        self.hop = self.get_int("hop")
        self.device = device
        self.rank, self.world = 0, 1                 # until comm_init: comm_info() then reports the C side's ZVX_E_STATE, not an AttributeError

    def close(self):
        if getattr(self, "_h", None):
            self._lib.zvx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != ZVX_OK:
            raise ZvxError(rc, self._lib.zvx_last_error(self._h).decode())

    def get_int(self, key):
        return int(self._lib.zvx_get_int(self._h, key.encode()))

    def set_int(self, key, value):
        self._chk(self._lib.zvx_set_int(self._h, key.encode(), int(value)))

    # ---- stages -------------------------------------------------------------------------------
    def spkemb(self, ref_mels, lens):
        ref_mels = _f32(ref_mels)
        B, Tmax, F = ref_mels.shape
        assert F == self.n_mels
        lens = _i32(lens, (B,))
        out = np.empty((B, self.hidden), np.float32)
        self._chk(self._lib.zvx_spkemb(self._h, _ptr(ref_mels), _ptr(lens), B, Tmax, _ptr(out)))
        return out

    def melspec(self, wavs):
        """list of 1-D float waveforms -> (log-mel [B][Tmax][n_mels], frames [B])   (get_mel_from_wav on the device)"""
        B = len(wavs)
        n = np.array([len(w) for w in wavs], np.int32)
        Nmax = int(n.max())
        wav = np.zeros((B, Nmax), np.float32)
        for b, w in enumerate(wavs):
            wav[b, :n[b]] = np.asarray(w, np.float32)
        pad = (self.get_int("fft_size") - self.hop) // 2
        Tmax = max(1, 1 + (Nmax + 2 * pad - self.get_int("fft_size")) // self.hop)
        mel = np.zeros((B, Tmax, self.n_mels), np.float32)
        frames = np.zeros(B, np.int32)
        self._chk(self._lib.zvx_melspec(self._h, _ptr(wav), _ptr(n), B, Nmax, _ptr(mel), Tmax, _ptr(frames)))
        return mel, frames

    def encode(self, phoneme, puncts, T, spk, duration=None):
        phoneme = _i32(phoneme)
        B, Tmax = phoneme.shape
        puncts = _i32(puncts, (B, Tmax))
        T = _i32(T, (B,))
        spk = _f32(spk).reshape(B, self.hidden)
        dur = _i32(duration, (B, Tmax)) if duration is not None else None
        mel_len = np.zeros(B, np.int32)
        logd = np.zeros((B, Tmax), np.float32)
        pitch = np.zeros((B, Tmax), np.float32)
        energy = np.zeros((B, Tmax), np.float32)
        self._chk(self._lib.zvx_encode(self._h, _ptr(phoneme), _ptr(puncts), _ptr(dur), _ptr(T), B, Tmax, _ptr(spk),
                                       _ptr(mel_len), _ptr(logd), _ptr(pitch), _ptr(energy)))
        return mel_len, logd, pitch, energy

    def decode(self, B, Lmax):
        # the library writes rows [0, ctx Lmax) of every utterance; a larger (or degenerate) request keeps zeros in the rest
        mel = np.zeros((B, max(Lmax, 1), self.n_mels), np.float32)
        self._chk(self._lib.zvx_decode(self._h, _ptr(mel), max(Lmax, 1), 0))
        return mel

    def decode_features(self, features, L, spk):
        features = _f32(features)
        B, Lmax, H = features.shape
        assert H == self.hidden
        L = _i32(L, (B,))
        spk = _f32(spk).reshape(B, H)
        mel = np.empty((B, Lmax, self.n_mels), np.float32)
        self._chk(self._lib.zvx_decode_features(self._h, _ptr(features), _ptr(L), B, Lmax, _ptr(spk), _ptr(mel), Lmax))
        return mel

    def vocode(self, B, mel_len, pad_to=None, pcm16=False):
        """wav [B][max(mel_len)*hop]: float32, or int16 PCM (x32760, truncated) with pcm16.  The array is created with
        np.empty on purpose: the library owns every byte it hands back (valid samples, then zeros)."""
        n0 = int(np.max(mel_len)) * self.hop
        n = max(n0, 1)
        wav = (np.empty if n0 > 0 else np.zeros)((B, n), np.int16 if pcm16 else np.float32)    # all lengths 0: nothing is copied back
        pt = _i32(pad_to, (B,)) if pad_to is not None else None
        self._chk(self._lib.zvx_vocode(self._h, _ptr(pt), _ptr(wav), n, ZVX_PCM16 if pcm16 else 0))
        return wav

    def vocode_mel(self, mel, P, pcm16=False, host_async=False):
        """host_async: the call only queues work and returns the pinned host slot (wait_host(slot) hands out the rows)."""
        mel = _f32(mel)
        B, Pmax, nm = mel.shape
        assert nm == self.n_mels
        P = _i32(P, (B,))
        if host_async:
            self._chk(self._lib.zvx_vocode_mel(self._h, _ptr(mel), _ptr(P), B, Pmax, None, 0, ZVX_HOST_ASYNC | (ZVX_PCM16 if pcm16 else 0)))
            return self.get_int("host_slot")
        n = int(P.max()) * self.hop
        wav = np.empty((B, n), np.int16 if pcm16 else np.float32)
        self._chk(self._lib.zvx_vocode_mel(self._h, _ptr(mel), _ptr(P), B, Pmax, _ptr(wav), n, ZVX_PCM16 if pcm16 else 0))
        if n < Pmax * self.hop:                                      # callers index rows up to Pmax*hop
            wav = np.concatenate([wav, np.zeros((B, Pmax * self.hop - n), wav.dtype)], axis=1)
        return wav

    def synthesize(self, phoneme, puncts, T, spk, duration=None, pad_to=None, want_mel=True, Lmax_cap=0,
                   wav_device_ptr=None, wav_stride=None, no_sync=False, pcm16=False, mel_device_ptr=None, host_async=False):
        """Batched phoneme -> waveform.  Returns dict(wav [B][N] (None if device output), mel_len, mel, log_duration).
        host_async: the call only queues work and returns dict(..., slot=s); wait_host(s) hands out the waveform rows in the
        context's pinned host memory (ZVX_HOST_ASYNC: forced durations, no mel / log-duration output).
        With a device waveform (wav_device_ptr) the mel, if wanted, is a device buffer too (ZVX_DEVICE_OUT covers both outputs):
        mel_device_ptr -> [B][Lmax][n_mels] f32 with Lmax = the longest utterance's forced-duration sum (or Lmax_cap)."""
        phoneme = _i32(phoneme)
        B, Tmax = phoneme.shape
        puncts = _i32(puncts, (B, Tmax))
        T = _i32(T, (B,))
        spk = _f32(spk).reshape(B, self.hidden)
        dur = _i32(duration, (B, Tmax)) if duration is not None else None
        pt = _i32(pad_to, (B,)) if pad_to is not None else None
        if dur is not None:
            Lmax = int(max(np.maximum(dur[b, :T[b]], 0).sum() for b in range(B)))
        else:
            Lmax = int(Lmax_cap)
            if Lmax <= 0:
                raise ZvxError(ZVX_E_INVALID, "predicted durations need Lmax_cap (or use encode/decode/vocode)")
        mel_len = np.zeros(B, np.int32)
        if host_async:
            if want_mel or wav_device_ptr is not None:
                raise ZvxError(ZVX_E_INVALID, "host_async delivers the waveform only (want_mel=False, no device pointer)")
            self._chk(self._lib.zvx_synthesize(self._h, _ptr(phoneme), _ptr(puncts), _ptr(dur), _ptr(T), B, Tmax, _ptr(spk),
                                               _ptr(pt), Lmax, None, 0, _ptr(mel_len), None, max(Lmax, 1), None,
                                               ZVX_HOST_ASYNC | (ZVX_PCM16 if pcm16 else 0)))
            return dict(wav=None, mel_len=mel_len, mel=None, log_duration=None, slot=self.get_int("host_slot"))
        # a queued call (device output, no_sync) must not ask for host outputs: a copy into pageable memory would wait for the stream
        logd = None if (wav_device_ptr is not None and no_sync) else np.zeros((B, Tmax), np.float32)
        mel = np.zeros((B, max(Lmax, 1), self.n_mels), np.float32) if (want_mel and wav_device_ptr is None) else None
        mptr = _ptr(mel)
        flags = ZVX_PCM16 if pcm16 else 0
        if wav_device_ptr is not None:
            wav, wptr, stride = None, C.c_void_p(int(wav_device_ptr)), int(wav_stride)
            flags |= ZVX_DEVICE_OUT | (ZVX_NO_SYNC if no_sync else 0)
            if want_mel:
                if mel_device_ptr is None:
                    raise ZvxError(ZVX_E_INVALID, "a device waveform output takes a device mel output (mel_device_ptr) or want_mel=False")
                mptr = C.c_void_p(int(mel_device_ptr))
        else:
            stride = max(Lmax * self.hop, 1)
            wav = np.zeros((B, stride), np.int16 if pcm16 else np.float32)
            wptr = _ptr(wav)
        self._chk(self._lib.zvx_synthesize(self._h, _ptr(phoneme), _ptr(puncts), _ptr(dur), _ptr(T), B, Tmax, _ptr(spk),
                                           _ptr(pt), Lmax, wptr, stride, _ptr(mel_len), mptr, max(Lmax, 1),
                                           _ptr(logd), flags))
        return dict(wav=wav, mel_len=mel_len, mel=mel, log_duration=logd)

    def wait_host(self, slot: int, pcm16=False):
        """Waveform rows of the ZVX_HOST_ASYNC call that used `slot`: an ndarray VIEW [B][valid samples] of the context's pinned host
        memory (no copy; valid until the second next host_async call -- copy it to keep it)."""
        rows, stride, nrows, valid = C.c_void_p(), C.c_int64(), C.c_int32(), C.c_int64()
        self._chk(self._lib.zvx_wait_host(self._h, int(slot), C.byref(rows), C.byref(stride), C.byref(nrows), C.byref(valid)))
        ct = C.c_int16 if pcm16 else C.c_float
        n = int(nrows.value) * int(stride.value)
        flat = np.ctypeslib.as_array(C.cast(rows, C.POINTER(ct)), shape=(n,))
        return flat.reshape(int(nrows.value), int(stride.value))[:, :max(int(valid.value), 1)]

    # ---- introspection ------------------------------------------------------------------------
    def fetch(self, what, shape):
        out = np.zeros(shape, np.float32)
        self._chk(self._lib.zvx_fetch(self._h, what.encode(), _ptr(out), out.size))
        return out

    def sync(self):
        self._chk(self._lib.zvx_sync(self._h))

    def stage_times(self):
        ms = np.zeros(ZVX_T_COUNT, np.float32)
        self._chk(self._lib.zvx_stage_times(self._h, _ptr(ms)))
        return {n: float(ms[i]) for i, n in enumerate(STAGES)}

    def kernel_stats(self):
        arr = (KernelStat * 32)()
        n = self._lib.zvx_kernel_stats(self._h, arr, 32)
        return [dict(name=arr[i].name.decode(), launches=int(arr[i].launches), ms=float(arr[i].ms),
                     flops=float(arr[i].flops), bytes=float(arr[i].bytes)) for i in range(n)]

    def tag_stats(self):
        """per pipeline stage: launches, ms, algorithmic FLOPs and bytes (profile 2, profile_only -1)"""
        arr = (KernelStat * 64)()
        n = self._lib.zvx_tag_stats(self._h, arr, 64)
        return [dict(name=arr[i].name.decode(), launches=int(arr[i].launches), ms=float(arr[i].ms),
                     flops=float(arr[i].flops), bytes=float(arr[i].bytes)) for i in range(n)]

    def reset_stats(self):
        self._chk(self._lib.zvx_reset_stats(self._h))

    # ---- device buffers + the multi-GPU waveform gather (RCCL inside libzvx; no torch) ----------------------------
    def dev_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._chk(self._lib.zvx_dev_alloc(self._h, int(nbytes), C.byref(p)))
        return int(p.value)

    def dev_free(self, ptr: int):
        self._chk(self._lib.zvx_dev_free(self._h, C.c_void_p(int(ptr))))

    def dev_from_host(self, ptr: int, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self._lib.zvx_dev_from_host(self._h, C.c_void_p(int(ptr)), _ptr(arr), arr.nbytes))

    def spkemb_device(self, mels_ptr: int, lens, B: int, Tmax: int, out_ptr: int, no_sync=False):
        """speaker encoder on device-resident mels [B][Tmax][n_mels] -> device embeddings [B][hidden]"""
        lens = _i32(lens, (B,))
        self._chk(self._lib.zvx_spkemb_ex(self._h, C.c_void_p(int(mels_ptr)), _ptr(lens), B, Tmax, C.c_void_p(int(out_ptr)),
                                          ZVX_DEVICE_IN | ZVX_DEVICE_OUT | (ZVX_NO_SYNC if no_sync else 0)))

    def vocode_mel_device(self, mel_ptr: int, P, Pmax: int, wav_ptr: int, wav_stride: int, no_sync=False, pcm16=False):
        """stand-alone vocoder on a device-resident mel [B][Pmax][n_mels] -> device waveform rows"""
        P = _i32(P)
        self._chk(self._lib.zvx_vocode_mel(self._h, C.c_void_p(int(mel_ptr)), _ptr(P), len(P), int(Pmax), C.c_void_p(int(wav_ptr)), int(wav_stride),
                                           ZVX_DEVICE_IN | ZVX_DEVICE_OUT | (ZVX_NO_SYNC if no_sync else 0) | (ZVX_PCM16 if pcm16 else 0)))

    def dev_to_host(self, ptr: int, shape, dtype):
        out = np.empty(shape, dtype)
        self._chk(self._lib.zvx_dev_to_host(self._h, _ptr(out), C.c_void_p(int(ptr)), out.nbytes))
        return out

    @staticmethod
    def comm_unique_id() -> bytes:
        """rank 0: a fresh RCCL communicator id (128 bytes) to hand to every rank's comm_init."""
        lib = load()
        buf = C.create_string_buffer(ZVX_COMM_ID_BYTES)
        rc = lib.zvx_comm_unique_id(buf)
        if rc != ZVX_OK:
            raise ZvxError(rc, lib.zvx_last_error(None).decode())
        return buf.raw

    def comm_init(self, comm_id, rank: int, world: int):
        buf = C.create_string_buffer(bytes(comm_id), ZVX_COMM_ID_BYTES) if comm_id is not None else None
        self._chk(self._lib.zvx_comm_init(self._h, buf, int(rank), int(world)))
        self.rank, self.world = int(rank), int(world)

    def comm_gather(self, local_ptr: int, nbytes: int, recv_ptr, root: int = 0, no_sync: bool = False):
        self._chk(self._lib.zvx_comm_gather(self._h, C.c_void_p(int(local_ptr)), int(nbytes),
                                            C.c_void_p(int(recv_ptr)) if recv_ptr else None, int(root),
                                            ZVX_NO_SYNC if no_sync else 0))

    def comm_barrier(self):
        self._chk(self._lib.zvx_comm_barrier(self._h))

    def comm_info(self) -> dict:
        """Collective: what the communicator itself reports (ranks, RCCL version) and every rank's device PCI address."""
        n = 4 + self.world
        out = (C.c_int64 * n)()
        self._chk(self._lib.zvx_comm_info(self._h, out, n))
        ver = int(out[2])
        pci = [int(out[4 + r]) for r in range(self.world)]
        return {"world": int(out[0]), "comm_count": int(out[1]), "version_code": ver,
                "version": (f"{ver // 10000}.{(ver // 100) % 100}.{ver % 100}" if ver > 0 else None),
                "ranks_seen": int(out[3]),
                "device_pci": [(f"{p >> 16:04x}:{(p >> 8) & 0xff:02x}:{(p >> 3) & 0x1f:02x}.{p & 7}" if p >= 0 else None) for p in pci]}

    def comm_max(self, value: float) -> float:
        v = C.c_double(float(value))
        self._chk(self._lib.zvx_comm_max_f64(self._h, C.byref(v)))
        return float(v.value)
