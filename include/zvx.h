/*
 * zvx.h -- C-ABI of libzvx, the MI355X-native ZeroVOX synthesis path.
 *
 * The reference (gooofy/zerovox) has no FFI/plugin interface: its seam is the Python object API
 * (SURVEY.md 8b).  Each entry point below names the reference call it replaces.  A maintainer binds
 * them with ctypes; no torch types appear in any signature.
 *
 * Conventions
 *   - every function returns a zvx_status (0 = ok); zvx_last_error() returns the message of the last
 *     failure on that context (or of zvx_create when ctx is NULL);
 *   - one context per device, NOT thread-safe (like the reference object); no mutable process-wide state: every
 *     switch of zvx_set_int lives in its context, several contexts may be driven from several threads;
 *   - buffers are caller-allocated.  Host pointers by default; outputs flagged ZVX_DEVICE_OUT are
 *     device pointers on the context's device (used for the RCCL waveform gather);
 *   - batches are padded row-major: [B][Tmax] ids, [B][Lmax][80] mels, wav rows of `wav_stride` floats;
 *     every utterance is computed exactly as an independent batch-1 reference call
 *     (model.py:325-328 is batch-1 only), i.e. no statistic ever sees padding;
 *   - activations are time-major / channel-contiguous on the device ([time][channel]).
 */
#ifndef ZVX_H
#define ZVX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zvx_ctx zvx_ctx;
typedef int zvx_status;

enum {
    ZVX_OK = 0,
    ZVX_E_INVALID = 1,   /* bad argument / id out of range (reference: torch IndexError, assert) */
    ZVX_E_MANIFEST = 2,  /* manifest or weight blob malformed, tensor missing, unknown decoder kind (model.py:244) */
    ZVX_E_HIP = 3,       /* HIP runtime failure (message carries hipGetErrorString) */
    ZVX_E_STATE = 4,     /* call order violated (e.g. zvx_decode before zvx_encode) */
    ZVX_E_BUFFER = 5,    /* caller buffer too small for the produced length */
    ZVX_E_UNSUPPORTED = 6
};

enum {                   /* flags */
    ZVX_DEVICE_OUT = 1,  /* wav (and mel, if given) output pointers are device pointers */
    ZVX_NO_SYNC = 2,     /* do not hipStreamSynchronize before returning (device outputs only) */
    ZVX_DEVICE_IN = 8,   /* the bulk input (zvx_vocode_mel: mel, zvx_spkemb_ex: ref_mels) is a device pointer */
    ZVX_PCM16 = 4,       /* wav rows are int16 PCM: (int16)(sample * 32760), truncated like numpy astype (demo.py:29-35,
                            model.py:44-63); halves the bytes of the multi-GPU waveform gather.  wav_stride stays in samples */
    ZVX_HOST_ASYNC = 16  /* zvx_synthesize / zvx_vocode / zvx_vocode_mel: the waveform is delivered to HOST memory without the call waiting
                            for it (round 6) -- `wav` / `wav_stride` are ignored (wav may be NULL): the rows go to one of the context's two
                            pinned host slots on a copy stream of their own, the call returns after queueing (as with
                            ZVX_DEVICE_OUT | ZVX_NO_SYNC; no host mel / log_duration output), zvx_get_int("host_slot") names the slot
                            and zvx_wait_host(ctx, slot, ...) is where the host meets the rows.  Slots alternate (call i: slot i & 1);
                            a slot's rows stay valid until the second next ZVX_HOST_ASYNC call */
};

enum {                   /* zvx_stage_times indices (milliseconds, hipEvent-timed on the ctx stream) */
    ZVX_T_ENCODER = 0, ZVX_T_VARIANCE = 1, ZVX_T_LENREG = 2, ZVX_T_DECODER = 3, ZVX_T_VOCODER = 4,
    ZVX_T_SPKEMB = 5, ZVX_T_COUNT = 8
};

/* Build a synthesis context on HIP device `device` from a text manifest + fp32 weight blob produced
 * by zerovox_amd.pack (weight-norm folded, conv weights laid out [tap][Cout][Cin]).
 * Replaces ZeroVoxTTS.__init__ / ZeroVox.load_from_checkpoint + get_meldec
 * (synthesize.py:48-97, model.py:86-118, model.py:206-249). */
zvx_status zvx_create(const char* manifest, const void* weights, size_t nbytes, int device, zvx_ctx** out);
void       zvx_destroy(zvx_ctx* ctx);
const char* zvx_last_error(const zvx_ctx* ctx);
/* "precision" -> 0 bf16 / 1 f32; "hidden", "n_mels", "hop", "device" ... ; -1 if unknown.
 * "host_slot": the pinned host slot the last ZVX_HOST_ASYNC call writes (-1: none yet).
 * "f16_sat_events" (round 6): the telltale of the half mode's clamp.  Every 16-bit store of the IEEE-half kernels saturates at +-65504
 *   (below); after zvx_set_int("f16_sat_check", 1) every convolution of the vocoder and the mel decoders runs as its own launch (no
 *   LDS-resident intermediate; same arithmetic per convolution, ~3x the time: a debug mode) and every 16-bit tensor it writes is
 *   scanned for clamped values.  This key drains the context and returns how many were seen since the switch was last set: 0 means no
 *   clamp engaged on the inputs run so far -- the check to make once on a real checkpoint (the reference, fp32, has no such failure
 *   mode).  "f16_sat_check" returns the switch. */
int64_t    zvx_get_int(const zvx_ctx* ctx, const char* key);
/* "profile" 0/1/2 (0 off, 1 per-stage events, 2 + per-GEMM-launch events); "profile_only" variant id (-1 = all);
 * "shape_log" 0/1 (one stderr line per timed launch); "max_frames" hard cap on a predicted mel length (default 2^18:
 * the reference has none, fs2.py:678-681 -- a garbage log-duration must not drive an allocation -> ZVX_E_BUFFER).
 * "f16_sat_check" 0/1: the saturation audit of the half mode (see zvx_get_int "f16_sat_events"); setting it (re)zeroes the counter.
 * Every other key is an A/B switch of a scheduling / tiling / arithmetic choice (INTEGRATION.md has the table: "enc_split",
 * "front_overlap", "front_prio", "dec_flat", "dec_sc_fuse", "dec_f16", "dec_y16", "dec_qkv", "voc_f16", "voc_f16_stages", "stagefuse", "rb2fuse", "pairstream", "resstream", "slab_small", "slab_flat",
 * "poison_pads", "spk_pool_fuse", "spk_s2_fuse", ...);
 * all of them live in the context.  Unknown keys: ZVX_E_INVALID. */
zvx_status zvx_set_int(zvx_ctx* ctx, const char* key, int64_t value);

/* Speaker encoder: ref_mels [B][Tmax][80] log-mels, lens[B] frames -> out [B][hidden], L2-normalised.
 * Replaces ResNetSE34V2.forward (ResNetSE34V2.py:176-212) as called by ZeroVoxTTS.speaker_embed
 * (synthesize.py:139-141). */
zvx_status zvx_spkemb(zvx_ctx* ctx, const float* ref_mels, const int32_t* lens, int B, int Tmax, float* out);
/* same with flags: ZVX_DEVICE_IN (ref_mels on the device), ZVX_DEVICE_OUT (out on the device), ZVX_NO_SYNC */
zvx_status zvx_spkemb_ex(zvx_ctx* ctx, const float* ref_mels, const int32_t* lens, int B, int Tmax, float* out, int flags);

/* Log-mel front end of reference audio: wav [B][Nmax] f32 in [-1, 1] with nsamples[b] valid samples ->
 * mel [B][Tmax][n_mels] = log(clip(mel_basis . |STFT|, 1e-5)) (reflect padding (n_fft-hop)/2, hann window, center=False)
 * and frames[b] = 1 + (nsamples[b] + 2*pad - n_fft) / hop.  Rows >= frames[b] are zero.  ZVX_E_INVALID if an utterance is
 * shorter than pad + 1 samples or has more than Tmax frames.
 * Replaces get_mel_from_wav (mels.py:357-395) as called by ZeroVoxTTS.speaker_embed (synthesize.py:128-137). */
zvx_status zvx_melspec(zvx_ctx* ctx, const float* wav, const int32_t* nsamples, int B, int Nmax, float* mel, int Tmax,
                       int32_t* frames);

/* Phoneme encoder + variance adaptor + length regulator.  duration == NULL -> predicted durations
 * (fs2.py:678-681), else forced (force_duration=True, fs2.py:745).  Writes mel_len[B]; optional
 * log_duration / pitch / energy [B][Tmax].  The expanded features stay in the context.
 * Replaces FS2Encoder.forward (fs2.py:732-775). */
zvx_status zvx_encode(zvx_ctx* ctx, const int32_t* phoneme, const int32_t* puncts, const int32_t* duration,
                      const int32_t* T, int B, int Tmax, const float* spk,
                      int32_t* mel_len, float* log_duration, float* pitch, float* energy);

/* Mel decoder on the context's features; mel_out [B][Lstride][n_mels] may be NULL; rows in [mel_len[b], max_b mel_len)
 * of utterance b are written as zeros.
 * Replaces FS2Decoder.forward (fs2.py:281-315) / StyleTTSDecoder.forward (styletts.py:181-205). */
zvx_status zvx_decode(zvx_ctx* ctx, float* mel_out, int Lstride, int flags);

/* Same on caller-supplied features [B][Lmax][hidden] (L[B] valid frames, spk [B][hidden]). */
zvx_status zvx_decode_features(zvx_ctx* ctx, const float* features, const int32_t* L, int B, int Lmax,
                               const float* spk, float* mel_out, int Lstride);

/* HiFi-GAN on the context's mel.  pad_to[B]: utterance b is vocoded on max(pad_to[b], mel_len[b]) frames,
 * rows >= mel_len zero (the reference's stateful `_min_mel_len`, model.py:331-335; NULL = no padding);
 * wav row b receives mel_len[b]*hop samples (model.py:347) followed by ZEROS up to max_b(mel_len[b])*hop -- the bytes
 * handed back never depend on earlier calls on the context; samples beyond that bound are not touched.  wav_stride
 * samples between rows; float rows, or int16 rows with ZVX_PCM16.
 * Replaces hifigan.Generator.forward (hifigan.py:114-130). */
zvx_status zvx_vocode(zvx_ctx* ctx, const int32_t* pad_to, void* wav, int64_t wav_stride, int flags);

/* Stand-alone vocoder: mel [B][Pmax][n_mels], P[B] frames -> wav rows of P[b]*hop samples.
 * Arithmetic of the 16-bit mode: weights, activations and the running sum of the generator are IEEE half on the f16 MFMA (round 5) in
 * every stage but a ResBlock1 stage of 128 channels, which computes in bf16 (round 6; zvx_set_int "voc_f16_stages": a mask of the stages
 * in half, "voc_f16" 0: the bf16 kernels of rounds 1-4 everywhere).  Every half store SATURATES at +-65504 (MODE.FP16_OVFL in the
 * kernels) -- a mel scaled far past the trained range gives a finite, clipped waveform, never Inf / NaN; whether a clamp ever engaged on
 * given weights and inputs: zvx_set_int "f16_sat_check" / zvx_get_int "f16_sat_events".
 * Non-finite input (NaN / Inf in a mel): the call succeeds and nothing faults; the samples of THAT utterance are unspecified (finite or
 * not -- the leaky-relu forms are compiled without NaN propagation guarantees); every other utterance of the batch and every later call
 * are bit for bit what they are without it.  The reference would propagate the NaN through that utterance as well. */
zvx_status zvx_vocode_mel(zvx_ctx* ctx, const float* mel, const int32_t* P, int B, int Pmax,
                          void* wav, int64_t wav_stride, int flags);

/* encode + decode + vocode.  mel_out / log_duration may be NULL.  With predicted durations the caller
 * sizes wav for Lmax_cap frames per utterance; ZVX_E_BUFFER if a prediction exceeds it.
 * Queued calls: with forced durations, ZVX_DEVICE_OUT | ZVX_NO_SYNC and mel_out == log_duration == NULL the call only
 * QUEUES work on the context's stream and returns (the host inputs are copied into pinned staging before it returns and may be
 * reused at once; mel_len is filled from the durations): successive calls keep the GPU fed whatever the host thread's timing.
 * With predicted durations the call waits once, for the predicted mel lengths.
 * Queued calls overlap (round 4): the context issues encoder / variance adaptor / mel decoder on its front stream and the vocoder on
 * its main stream; call i + 1's front end waits only for call i's vocoder to have copied the mel, so it runs UNDER that vocoder
 * (23.1 -> 21.2 ms per 32 x 128-phoneme batch on one context; bit-identical to the serial schedule; zvx_set_int "front_overlap").
 * A device mel output (mel_out with ZVX_DEVICE_OUT) is written on the front stream; zvx_sync drains every stream.
 * Replaces ZeroVox.inference_ex (model.py:308-347) over B independent utterances. */
zvx_status zvx_synthesize(zvx_ctx* ctx, const int32_t* phoneme, const int32_t* puncts, const int32_t* duration,
                          const int32_t* T, int B, int Tmax, const float* spk, const int32_t* pad_to,
                          int Lmax_cap, void* wav, int64_t wav_stride, int32_t* mel_len,
                          float* mel_out, int Lstride, float* log_duration, int flags);

/* Host side of ZVX_HOST_ASYNC: blocks until the waveform copy into `slot` (0 / 1) has landed, then hands out the slot's pinned rows:
 * *rows -> [*nrows][*stride] samples (f32, or int16 for a ZVX_PCM16 call), the first *valid samples of a row defined as for zvx_vocode
 * (mel_len[b]*hop samples, then zeros up to max_b).  Any out pointer may be NULL.  The memory belongs to the context (freed by
 * zvx_destroy, reused by the second next ZVX_HOST_ASYNC call).  ZVX_E_STATE if no call has used the slot.
 * Replaces the `.cpu().numpy()` hand-over of ZeroVoxTTS.tts_ex (synthesize.py:233-239) for a host that keeps calls in flight. */
zvx_status zvx_wait_host(zvx_ctx* ctx, int slot, const void** rows, int64_t* stride, int32_t* nrows, int64_t* valid);

/* Debug/parity taps: copy an intermediate of the last call to host fp32.
 * what: "encoder_out" [B][Tmax][hidden] (after the style add), "features" [B][Lmax][hidden],
 *       "mel" [B][Lmax][n_mels], "pitch_idx"/"energy_idx"/"duration" [B][Tmax] (as float). */
zvx_status zvx_fetch(zvx_ctx* ctx, const char* what, float* out, size_t out_floats);

/* ---- multi-GPU: utterances shard across ranks with no data-path exchange; the ONE collective is the gather of the
 * finished waveform rows to one rank, issued here directly on RCCL (grouped ncclSend / ncclRecv: every peer -> root
 * transfer rides its own xGMI link).  The reference has no distributed layer; one process per GPU,
 * one context per process.  librccl.so.1 is dlopen'ed by zvx_comm_unique_id / zvx_comm_init only; it must belong to the
 * same ROCm runtime as the HIP library already in the process (do not import a framework that bundles its own ROCm AFTER
 * this library has been loaded). ---- */
#define ZVX_COMM_ID_BYTES 128
/* rank 0 creates the communicator id (ncclGetUniqueId) and ships the 128 bytes to the other ranks out of band */
zvx_status zvx_comm_unique_id(void* id_out);
zvx_status zvx_comm_init(zvx_ctx* ctx, const void* id, int rank, int world);
/* Every rank contributes `bytes` bytes at device pointer `local`; rank `root` receives them in rank order at device
 * pointer `recv` (world * bytes; ignored elsewhere).  Enqueued on the context's communication stream behind everything
 * issued so far on its compute stream, so with ZVX_NO_SYNC it overlaps the next synthesis call; a later
 * ZVX_DEVICE_OUT synthesis into `local` waits on the device for this gather to have read it.  world == 1: a copy. */
zvx_status zvx_comm_gather(zvx_ctx* ctx, const void* local, size_t bytes, void* recv, int root, int flags);
/* all ranks: returns after every rank has drained both of its streams and arrived (ncclAllReduce of one word) */
zvx_status zvx_comm_barrier(zvx_ctx* ctx);
/* *value = max over ranks (bench.py: MAX-over-ranks elapsed time) */
zvx_status zvx_comm_max_f64(zvx_ctx* ctx, double* value);
/* Collective (every rank calls it): who is in the job as the communicator itself reports it -- out[0] = world, out[1] =
 * ncclCommCount, out[2] = ncclGetVersion code, out[3] = ranks that contributed to an all-reduce SUM of ones, out[4 + r] = PCI
 * address ((domain << 16) | (bus << 8) | (device << 3) | function) of rank r's device.  n_out >= 4 + world.  bench.py puts it
 * into the N > 1 JSON line ("rccl": {...}). */
zvx_status zvx_comm_info(zvx_ctx* ctx, int64_t* out, int n_out);
void       zvx_comm_destroy(zvx_ctx* ctx);

/* Device buffers for ZVX_DEVICE_OUT outputs / zvx_comm_gather without any other GPU runtime in the process. */
zvx_status zvx_dev_alloc(zvx_ctx* ctx, size_t bytes, void** out);
zvx_status zvx_dev_free(zvx_ctx* ctx, void* p);
zvx_status zvx_dev_from_host(zvx_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
zvx_status zvx_dev_to_host(zvx_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);

/* drains every stream of the context (compute, front end, communication) */
zvx_status zvx_sync(zvx_ctx* ctx);
zvx_status zvx_stage_times(zvx_ctx* ctx, float ms[ZVX_T_COUNT]);

/* Per-kernel-variant counters accumulated while "profile" == 2: launches, summed milliseconds,
 * algorithmic FLOPs and algorithmic bytes.  Returns the number of variants; name buffers are 64 bytes. */
typedef struct {
    char   name[64];
    int64_t launches;
    double ms;
    double flops;
    double bytes;
} zvx_kernel_stat;
int        zvx_kernel_stats(zvx_ctx* ctx, zvx_kernel_stat* out, int max_out);
/* The same counters grouped by pipeline stage ("encoder", "variance", "lenreg", "decoder", "decoder.norm", "voc.pre",
 * "voc.up1".."voc.res4", "voc.post", "spkemb"; name = stage): every launch of the stage, including the HBM-bound helper
 * kernels, while "profile" == 2 and "profile_only" == -1.  Feeds the per-stage roofline fractions of bench.py. */
int        zvx_tag_stats(zvx_ctx* ctx, zvx_kernel_stat* out, int max_out);
zvx_status zvx_reset_stats(zvx_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* ZVX_H */
