// zvx.hip -- libzvx context, weight loading, launch sequences and the C-ABI (include/zvx.h).
//
// Host-side C++ only sequences kernels: every contraction is a launch_gemm() (gemm.hip), everything
// else a kernel from ops.hip.  One HIP stream per context; workspaces are grown on demand and reused.
// Reference call structure being replaced: ZeroVox.inference_ex (model.py:308-347).
#include "../../include/zvx.h"
#include "zvx_kernels.h"

#include <hip/hip_ext.h>
#include <dlfcn.h>
#include <math.h>
#include <cmath>
#include <rccl/rccl.h>          // types and prototypes only: the library is dlopen'ed by zvx_comm_* (single-GPU use never loads it)
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

using namespace zvx;

namespace {

struct ZvxError : std::runtime_error {
    int code;
    ZvxError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
[[noreturn]] void fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    throw ZvxError(code, buf);
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) fail(ZVX_E_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

thread_local std::string g_create_error;

struct Tensor {
    char kind = 'p';
    std::vector<int> dims;
    size_t off = 0, numel = 0;
    void* dev = nullptr;
    int dtype = DT_F32;
    const float* host = nullptr;
    float alpha = 1.f;                       // split-plane weights in IEEE half: the planes carry w * 2^s, alpha = 2^-s (applied by the GEMM epilogue)
    int dim(int i) const { return dims.at(i); }
};

struct DevBuf { void* p = nullptr; void* base = nullptr; size_t cap = 0; };

struct GemmEvent { hipEvent_t a, b; int variant; double flops, bytes; long rows; int N, K, taps, res, fused; std::string tag; };

}  // namespace

struct zvx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;          // the stream launches are being issued on (swapped to a side stream and back by the schedules below)
    hipStream_t main0 = nullptr;           // the context's main stream (what `stream` is outside those swaps)
    // multi-GPU (zvx_comm_*): communicator, its own stream, and per-buffer "the gather has read this" events
    hipStream_t comm_stream = nullptr;
    hipStream_t voc_aux[2] = {nullptr, nullptr};   // single requests: the non-final pairs of the 2nd / 3rd ResBlock of a vocoder stage run beside the 1st
    hipEvent_t voc_ev[3] = {nullptr, nullptr, nullptr};
    int voc_overlap_maxb = 1 << 20;        // zvx_set_int("voc_overlap_maxb", n): batches of at most n utterances use them (0: never; A/B)
    long voc_overlap_frames = 28672;       // zvx_set_int("voc_overlap_frames", n): ... and only below n mel frames per call (B x Pmax)
    // Front end of call i+1 under the vocoder of call i (zvx_synthesize): encoder / variance adaptor / length regulator / mel decoder are
    // issued on front_stream, the vocoder on `stream`.  The two meet in ONE buffer, "mel": the vocoder waits for ev_front_done, the
    // next call's front end waits for ev_mel_free (recorded behind the vocoder's first kernel, which copies the mel into its padded
    // input).  A host that queues calls (ZVX_DEVICE_OUT | ZVX_NO_SYNC) gets the latency-paced front end (5-28 % of the matrix roof)
    // hidden under the previous call's vocoder; a host that waits for every call sees the serial schedule.  Results are bit-identical
    // (same kernels on the same data).  zvx_set_int("front_overlap", v): 1 (default) = queued calls (ZVX_DEVICE_OUT | ZVX_NO_SYNC) take the
    // two-stream schedule, 2 = every zvx_synthesize call does (tests), 0 = everything on `stream` (A/B).
    hipStream_t front_stream = nullptr;
    hipEvent_t ev_front_done = nullptr, ev_mel_free = nullptr, ev_main_join = nullptr;
    int front_overlap = 1;
    bool mel_free_pending = false;         // ev_mel_free is recorded and the front stream has not waited for it yet
    bool front_dirty_main = true;          // front-end buffers were touched on `stream` (zvx_encode / zvx_decode ...) since the front stream last joined it
    // Asynchronous host delivery (ZVX_HOST_ASYNC; synthesize.py:233-239 hands the caller host memory): the finished waveform rows go to one
    // of two PINNED host slots on a copy stream of their own, behind an event the vocoder's last kernel records -- the copy of call i runs
    // under the front end / vocoder of call i + 1, the host thread never waits inside a synthesis call (zvx_wait_host does).
    struct HostSlot { void* p = nullptr; size_t cap = 0; hipEvent_t ready = nullptr, done = nullptr; bool pending = false; int B = 0, pcm16 = 0; long stride = 0, need = 0; };
    HostSlot host_slot[2];
    int host_next = 0, host_last = -1;
    hipStream_t copy_stream = nullptr;
    bool copy_is_comm = false;
    hipStream_t aux_stream = nullptr;      // second compute stream: the duration predictor of a small batch beside the pitch predictor
    hipEvent_t ev_aux[2] = {nullptr, nullptr};
    int va_overlap_maxb = 1 << 20;             // zvx_set_int("va_overlap_maxb", n): batches of at most n utterances overlap the two predictors (0: never; A/B)
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipEvent_t ev_compute = nullptr;
    std::map<const void*, hipEvent_t> gather_fence;
    std::string err;
    std::map<std::string, std::string> cfg;
    std::map<std::string, Tensor> tensors;
    std::vector<float> host_blob;
    std::map<std::string, DevBuf> bufs;
    std::map<const void*, const void*> packed;   // bf16 weight tensor -> its MFMA-fragment-order copy (conv-slab kernel)
    std::map<const void*, const void*> pair_packed;   // StyleTTS residual blocks with a 1x1 shortcut: conv2 weight tensor -> combined fragment stream [conv2 | shortcut] per channel tile
    int dt = DT_BF16;               // activation / weight dtype of the bf16-able stages
    // config
    int H = 0, emb_dim = 0, punct_dim = 0, n_phone_rows = 0, n_punct_rows = 0, max_txt_len = 0, max_mel_len = 0;
    int enc_layers = 0, enc_heads = 0, ffn_dim = 0, ffn_k0 = 0, ffn_k1 = 0, vp_dim = 0, vp_k = 0, n_bins = 0;
    int dec_layers = 0, dec_heads = 0, dec_scln = 0, n_mels = 0, hop = 0, res_dim = 0, dec_kind = 0, rn_asp = 1;
    std::vector<int> rn_layers, rn_filters, voc_rates, voc_ksizes, voc_rb_k;
    std::vector<std::vector<int>> voc_rb_d;
    int voc_resblock = 1, voc_c0 = 0;
    // per-call state
    int B = 0, Tmax = 0, Lmax = 0;
    std::vector<int> T_host, mel_len_host;
    std::vector<int> in_stage;            // host staging of a call's integer inputs (one upload)
    bool have_features = false, have_mel = false;
    // profiling
    int profile = 0;
    int profile_only = -1;                 // >= 0: per-launch events only for this kernel variant (keeps the timed region lean)
    int rs_prof = 0;                       // zvx_set_int("rs_prof", 1): per-wave cycle counters of the streaming kernels (library built with -DRS_PROFILE)
    int enc_split = 0;                     // 16-bit mode: the f32 GEMMs of the phoneme encoder run as 3-plane 16-bit GEMMs: 2 = IEEE-half planes (default: 2^-24-class, f32
                                           // results), 1 = bf16 planes (rounds 2-3: 5e-5 on the encoder output; A/B), 0 = the exact-f32 MFMA
    const void* fft_xs_ready = nullptr;     // fft.xs holds the split planes of this buffer (written by the previous FFT block's last LayerNorm)
    int norm_fuse_maxb = 1 << 20;          // zvx_set_int("norm_fuse_maxb", n): batches of at most n utterances may take the one-launch InstanceNorm of the StyleTTS decoder (0: never; A/B)
    int dec_sc_fuse = 1;                   // zvx_set_int("dec_sc_fuse", 0): the 1x1 shortcut of a StyleTTS residual block as its own launch (A/B; the fused form skips one 16-bit rounding of the conv2 result)
    int dec_flat = 1;                      // zvx_set_int("dec_flat", 0): the StyleTTS decoder's convolutions per utterance instead of batch-flattened (A/B, bit-identical)
    int voc_f16_stages = -1;               // zvx_set_int("voc_f16_stages", mask): which domains of the generator (bit 0: mel / conv_pre, bit i: upsampling stage i) compute in IEEE half when voc_f16 is on, the others in bf16; -1 (default): all but a 128-channel ResBlock1 stage
    int voc_h16_ok = -1;                   // every contraction weight of the generator has an IEEE-half copy (decided on the first vocoder call)
    int voc_f16 = 1;                       // zvx_set_int("voc_f16", 0): the vocoder's activations / weights / running sum in bf16 instead of IEEE half (A/B; round 5)
    int dec_qkv = 1;                       // zvx_set_int("dec_qkv", 0): the half FFT-block decoder's Q | K and V projections as two launches (A/B; round 6)
    int dec_y16 = 1;                       // zvx_set_int("dec_y16", 0): the half FFT-block decoder's pre-norm sums in f32 instead of half (A/B; round 6)
    int dec_f16 = 1;                       // zvx_set_int("dec_f16", 0): StyleTTS decoder activations / weights in bf16 instead of IEEE half (A/B)
    int use_attn_f32 = 1;                  // zvx_set_int("attn_f32", 0): the encoder's attention as V^T / score / P.V GEMMs + softmax (A/B)
    int use_flash = 1;                     // zvx_set_int("flash", 0): the decoder's attention as score GEMM + softmax + PV GEMM (A/B)
    int voc_chunk = 0;                     // utterances per vocoder ResBlock sub-batch (0 = whole batch)
    int use_resstream = 1;                 // zvx_set_int("resstream", 0): ResBlocks of the narrow stages as per-pair launches (A/B, bit-equal)
    int rs_seg_min = 0;                    // zvx_set_int("rs_seg_min", -1): streaming ResBlock segments never shorter than 2048 rows (A/B of the single-request sizing)
    int rs_opt = 3;                        // zvx_set_int("rs_opt", v): StreamArgs.opt of the streaming ResBlock kernels (bit 0: staggered wave priorities)
    int spk_s2_fuse = 1;                   // zvx_set_int("spk_s2_fuse", 0): the level transitions as two launches of the gathered-row GEMM (A/B)
    int spk_pool_fuse = 1;                 // zvx_set_int("spk_pool_fuse", 0): the speaker encoder's SE pool as its own pass everywhere (A/B)
    int slab_small = 2, slab_flat = 1;     // zvx_set_int("slab_small" / "slab_flat", v): conv-slab tile choice for single requests / whole-grid XCD remap (A/B; per context)
    int poison_pads = 0;                   // zvx_set_int("poison_pads", 1): every work buffer of the mel decoders is filled with NaN bit patterns before a decode (tests: padding rows / stale rows must never reach a result -- ADVICE r4)
    // Saturation audit of the half mode (zvx_set_int("f16_sat_check", 1)): every 16-bit tensor of the vocoder and the mel
    // decoders goes through HBM (one launch per convolution: no LDS-resident intermediate) and is scanned for clamped values
    // (+-65504) behind the launch that wrote it; zvx_get_int("f16_sat_events") is the count since the switch was last set.  A debug mode:
    // same arithmetic per convolution, ~3x the time.
    int sat_check = 0;
    unsigned long long* sat_count_dev() { return (unsigned long long*)buf("sat.count", 64); }
    void sat_scan(const void* x, int dt, long bs, int ld, int B, int rows_max, const int* rows, int C) {
        if (sat_check && x && dt == DT_F16) launch_count_sat16(x, bs, ld, B, rows_max, rows, C, sat_count_dev(), stream);
    }
    int use_rb2fuse = 1;                   // zvx_set_int("rb2fuse", 0): every convolution of a ResBlock2 (HiFi-GAN V3 / resblock "2") as its own launch instead of one launch per block (A/B)
    int use_stagefuse = 1;                 // zvx_set_int("stagefuse", 0): narrow vocoder stages (C = 16 / 8) as per-pair launches instead of ONE launch per stage (narrowstage.hip; A/B)
    struct NsWeights { void* W = nullptr; float* bias = nullptr; int woff[18] = {0}; };
    std::map<std::string, NsWeights> ns_weights;   // narrowstage.hip fragment order, per (stage, dtype), built on first use
    int use_pairstream = 1;                // zvx_set_int("pairstream", v): C = 128 ResBlock pairs on pairstream.hip: 1 = every k (default; jobs under ~200 k rows run the bit-identical two-launch path), 3 = every k and every job size (tests), 4 = like 3 with 256-row segments for small jobs, <= 0 = two conv-slab launches per pair (the bit-equality reference)
    int shape_log = 0;                     // zvx_set_int("shape_log", 1): one stderr line per timed launch (profile 2)
    int max_frames = 1 << 18;              // hard cap on a predicted mel length (guards the allocation, fs2.py:678-681 has none)
    hipEvent_t stage_ev[ZVX_T_COUNT][2];
    bool stage_used[ZVX_T_COUNT];
    float stage_ms[ZVX_T_COUNT];
    std::vector<GemmEvent> pending;
    std::vector<zvx_kernel_stat> stats;
    std::string tag = "other";             // stage label of the launches being issued (per-stage roofline accounting, profile 2)
    std::map<std::string, zvx_kernel_stat> tagstats;
    std::vector<hipEvent_t> event_pool;

    // ------------------------------------------------------------------ helpers
    int cfg_int(const char* k) const {
        auto it = cfg.find(k);
        if (it == cfg.end()) fail(ZVX_E_MANIFEST, "manifest: missing cfg '%s'", k);
        return atoi(it->second.c_str());
    }
    std::vector<int> cfg_list(const char* k) const {
        auto it = cfg.find(k);
        if (it == cfg.end()) fail(ZVX_E_MANIFEST, "manifest: missing cfg '%s'", k);
        std::vector<int> v; std::stringstream ss(it->second); std::string tok;
        while (std::getline(ss, tok, ',')) if (!tok.empty()) v.push_back(atoi(tok.c_str()));
        return v;
    }
    const Tensor& t(const std::string& name) const {
        auto it = tensors.find(name);
        if (it == tensors.end()) fail(ZVX_E_MANIFEST, "weights: tensor '%s' missing", name.c_str());
        return it->second;
    }
    bool has(const std::string& name) const { return tensors.count(name) != 0; }
    const float* pf(const std::string& name) const { return (const float*)t(name).dev; }

    // A buffer about to be freed may still be read by work queued on ANY of the context's streams (the two-stream schedule: the front
    // end of call i + 1 regrows "mel" on the front stream while call i's launch_mel_pad is queued on the main stream) -- every stream
    // of the context is drained first; hipFree's own device-wide synchronisation is not a contract to lean on
    void drain_for_free() {
        HIPCHK(hipStreamSynchronize(stream));
        for (hipStream_t s2 : {main0, front_stream, aux_stream, voc_aux[0], voc_aux[1], comm_stream, copy_stream})
            if (s2 && s2 != stream) HIPCHK(hipStreamSynchronize(s2));
    }
    void* buf(const std::string& name, size_t bytes) {
        DevBuf& d = bufs[name];
        if (bytes > d.cap) {
            if (d.base) { drain_for_free(); HIPCHK(hipFree(d.base)); d.base = nullptr; d.p = nullptr; }
            size_t cap = bytes + bytes / 8 + 256;
            HIPCHK(hipMalloc(&d.base, cap));
            HIPCHK(hipMemsetAsync(d.base, 0, cap, stream));
            d.p = d.base;
            d.cap = cap;
        }
        return d.p;
    }
    // Several small named buffers as slices of ONE allocation, so that a call zero-fills / uploads them with one operation instead of
    // one each (a single request is paced by its launch count).  Lookups by name keep working; the slices are re-cut on every call.
    char* carve(const std::string& pool, const std::vector<std::string>& names, size_t bytes_each, size_t* stride_out) {
        const size_t stride = (bytes_each + 255) & ~(size_t)255;
        char* base = (char*)buf(pool, stride * names.size());
        for (size_t i = 0; i < names.size(); i++) {
            DevBuf& d = bufs[names[i]];
            if (d.base) { drain_for_free(); HIPCHK(hipFree(d.base)); d.base = nullptr; }
            d.p = base + i * stride; d.cap = stride;
        }
        *stride_out = stride;
        return base;
    }
    int* ibuf(const std::string& name, size_t n) { return (int*)buf(name, n * sizeof(int)); }
    float* fbuf(const std::string& name, size_t n) { return (float*)buf(name, n * sizeof(float)); }
    // Small host -> device copies go through PINNED memory owned by the context.  A copy from pageable memory makes the runtime
    // wait for the stream before it returns; from pinned memory it is queued like a launch.  With forced durations and a device
    // output (ZVX_NO_SYNC) a whole zvx_synthesize call then only QUEUES work: the host runs ahead of the GPU by a call or more and
    // a slow or preempted host thread no longer shows up as GPU idle time.  Two arenas alternate between API calls; an arena is
    // reused only after the event recorded behind its last copy has passed.
    struct Arena { char* p = nullptr; size_t cap = 0, cur = 0; hipEvent_t ev = nullptr; bool pending = false; };
    Arena arena[2];
    int arena_i = 0;
    int api_depth = 0;
    static constexpr size_t ARENA_BYTES = 2u << 20;
    void arena_begin() {
        Arena& a = arena[arena_i];
        if (a.pending) { HIPCHK(hipEventSynchronize(a.ev)); a.pending = false; }
        a.cur = 0;
    }
    void arena_end() {                      // only a call that uploaded something rotates the arenas: a gather or a query between two
        Arena& a = arena[arena_i];          // synthesis calls must not send them both to the same arena (the second would wait for the first)
        if (!a.cur || !stream) return;
        if (!a.ev) HIPCHK(hipEventCreateWithFlags(&a.ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(a.ev, stream));
        a.pending = true;
        arena_i ^= 1;
    }
    void upload(void* dst, const void* src, size_t bytes) {
        Arena& a = arena[arena_i];
        const size_t need = (bytes + 63) & ~(size_t)63;
        if (!a.p) { HIPCHK(hipHostMalloc((void**)&a.p, ARENA_BYTES, hipHostMallocDefault)); a.cap = ARENA_BYTES; }
        if (a.cur + need > a.cap) {
            // larger than what is left of the arena: a plain copy, COMPLETED before this returns (the caller's / the context's staging
            // memory may be reused right after the call; the runtime's own staging of pageable copies is not a contract to lean on)
            HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
            HIPCHK(hipStreamSynchronize(stream));
            return;
        }
        memcpy(a.p + a.cur, src, bytes);
        HIPCHK(hipMemcpyAsync(dst, a.p + a.cur, bytes, hipMemcpyHostToDevice, stream));
        a.cur += need;
    }
    int* upload_ints(const std::string& name, const int* v, size_t n) {
        int* d = ibuf(name, n);
        upload(d, v, n * sizeof(int));
        return d;
    }
    size_t es() const { return dtype_size(dt); }

    hipEvent_t new_event() {
        hipEvent_t e;
        if (!event_pool.empty()) { e = event_pool.back(); event_pool.pop_back(); return e; }
        HIPCHK(hipEventCreate(&e));
        return e;
    }
    void stage_begin(int s) { if (profile) { HIPCHK(hipEventRecord(stage_ev[s][0], stream)); } }
    void stage_end(int s) { if (profile) { HIPCHK(hipEventRecord(stage_ev[s][1], stream)); stage_used[s] = true; } }

    void gemm(GemmArgs& a) {
        if (a.flops <= 0) a.flops = 2.0 * (double)a.M * a.nbatch * a.nheads * (double)a.N * ((double)a.K * a.ntaps + (double)a.K2);
        if (!a.Wp && a.dtype != DT_F32) { auto it = packed.find(a.W); if (it != packed.end()) a.Wp = it->second; }
        a.slab_small = slab_small | (spk_s2_fuse ? 0 : 32); a.xcd_flat = slab_flat;      // (bit 5: no conv2d_s2_kernel -- spk_s2_fuse 0 is the gathered-row launch set for BOTH level transitions)
        GemmEvent ev{};
        const bool prof = profile >= 2 && (profile_only < 0 || gemm_variant_of(a) == profile_only);
        if (prof) { ev.a = new_event(); ev.b = new_event(); gemm_profile_events(ev.a, ev.b); }
        int id = a.fused == 2 ? launch_rb2fuse(a, stream) : (a.fused ? launch_resfuse(a, stream) : launch_gemm(a, stream));
        if (prof) gemm_profile_events(nullptr, nullptr);
        if (id < 0) fail(ZVX_E_INVALID, "launch_gemm rejected shape M=%d N=%d K=%d taps=%d", a.M, a.N, a.K, a.ntaps);
        if (sat_check && a.nheads == 1) {                   // (audit mode: the attention products run fused, their output is scanned by fft_block)
            sat_scan(a.out, a.out_dtype, a.o_bs, a.ldo, a.nbatch, a.M, a.out_len, a.N);
            if (a.accum && (a.accum_mode & 2)) sat_scan(a.accum, a.accum_dtype, a.a_bs, a.lda, a.nbatch, a.M, a.out_len, a.N);
        }
        if (prof) {
            ev.variant = id; ev.flops = a.flops; ev.rows = (long)a.M * a.nbatch * a.nheads; ev.N = a.N; ev.K = a.K; ev.taps = a.ntaps; ev.res = a.res_mode; ev.fused = a.fused;
            const double esz = dtype_size(a.dtype);
            // algorithmic bytes: input rows + weights + everything the epilogue reads and writes (output, residual, running sum)
            const double cells = (double)a.M * a.nbatch * a.nheads * a.N;
            ev.bytes = ((double)a.M * a.nbatch * a.nheads) * (double)(a.K + a.K2) * esz + (double)a.N * ((double)a.K * a.ntaps + a.K2) * esz * (a.fused ? 2 : 1) +
                       (a.out ? cells * dtype_size(a.out_dtype) : 0.0) + ((a.res_mode && !a.fused) ? cells * dtype_size(a.res_dtype) : 0.0) +
                       ((a.accum && (a.accum_mode & 1)) ? cells * dtype_size(a.accum_dtype) : 0.0) + ((a.accum && (a.accum_mode & 2)) ? cells * dtype_size(a.accum_dtype) : 0.0);
            ev.tag = tag;
            pending.push_back(ev);
        }
    }
    // streaming ResBlock chain (resstream.hip); returns false (nothing launched) when the shape is not covered
    bool run_stream(StreamArgs& a) {
        const int id = launch_resstream(a, this->stream, true);
        if (id < 0) return false;
        GemmEvent ev{};
        const bool prof = profile >= 2 && (profile_only < 0 || id == profile_only);
        if (prof) { ev.a = new_event(); ev.b = new_event(); resstream_profile_events(ev.a, ev.b); }
        const int id2 = launch_resstream(a, this->stream, false);
        if (prof) resstream_profile_events(nullptr, nullptr);
        if (id2 < 0) fail(ZVX_E_INVALID, "launch_resstream rejected a shape its dry run accepted (C=%d k=%d pairs=%d)", a.C, a.ntaps, a.npair);
        if (prof) {
            const double rows = (double)a.M * a.nbatch;
            ev.variant = id2; ev.flops = 2.0 * 2.0 * a.npair * rows * a.C * a.C * a.ntaps; ev.rows = (long)rows; ev.N = a.C; ev.K = a.C; ev.taps = a.ntaps;
            ev.res = a.accum_mode; ev.fused = 10 + a.npair;
            ev.bytes = rows * a.C * 2.0 * (1 + (a.out ? 1 : 0) + ((a.accum && (a.accum_mode & 1)) ? 1 : 0) + ((a.accum && (a.accum_mode & 2)) ? 1 : 0));
            ev.tag = tag;
            pending.push_back(ev);
        }
        return true;
    }
    // the HBM-bound helpers (norms, gathers, conv_post ...): event-timed as a group when every launch is being profiled
    template <typename F>
    void timed(double flops, double bytes, F&& f) {
        const bool prof = profile >= 2 && profile_only < 0;
        GemmEvent ev{};
        if (prof) { ev.a = new_event(); ev.b = new_event(); HIPCHK(hipEventRecord(ev.a, stream)); }
        f();
        if (prof) { HIPCHK(hipEventRecord(ev.b, stream)); ev.variant = -1; ev.flops = flops; ev.bytes = bytes; ev.tag = tag; pending.push_back(ev); }
    }
    void resolve_events() {
        if (stats.empty()) {
            stats.resize(gemm_num_variants());
            for (int i = 0; i < gemm_num_variants(); i++) { memset(&stats[i], 0, sizeof stats[i]); snprintf(stats[i].name, 64, "%s", gemm_variant_name(i)); }
        }
        for (auto& e : pending) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
            if (shape_log && e.variant >= 0) fprintf(stderr, "launch %-24s rows=%-8ld N=%-4d K=%-4d taps=%-2d res=%d fused=%d  %8.3f ms %8.1f TF/s %8.1f GB/s(alg)\n",
                                   gemm_variant_name(e.variant), e.rows, e.N, e.K, e.taps, e.res, e.fused, ms, e.flops / ms / 1e9, e.bytes / ms / 1e6);
            else if (shape_log) fprintf(stderr, "launch %-24s [%s]  %8.3f ms %8.1f GB/s(alg)\n", "(helper)", e.tag.c_str(), ms, e.bytes / ms / 1e6);
            if (e.variant >= 0) { auto& s = stats[e.variant]; s.launches++; s.ms += ms; s.flops += e.flops; s.bytes += e.bytes; }
            auto& ts = tagstats[e.tag];
            if (!ts.name[0]) snprintf(ts.name, 64, "%s", e.tag.c_str());
            ts.launches++; ts.ms += ms; ts.flops += e.flops; ts.bytes += e.bytes;
            event_pool.push_back(e.a); event_pool.push_back(e.b);
        }
        pending.clear();
        for (int s = 0; s < ZVX_T_COUNT; s++)
            if (stage_used[s]) { float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, stage_ev[s][0], stage_ev[s][1])); stage_ms[s] = ms; stage_used[s] = false; }
    }
    void sync() {
        if (front_stream) HIPCHK(hipStreamSynchronize(front_stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (comm_stream) HIPCHK(hipStreamSynchronize(comm_stream));
        if (copy_stream && !copy_is_comm) HIPCHK(hipStreamSynchronize(copy_stream));
        if (profile) resolve_events();
    }
    // error path of an API call: whatever it queued on a side stream before it threw is drained, so that the next call's main-stream
    // work cannot race with it (the joins that normally order the streams were never issued)
    void quiesce_side_streams() noexcept {
        if (front_stream) (void)hipStreamSynchronize(front_stream);
        if (aux_stream) (void)hipStreamSynchronize(aux_stream);
        for (int i = 0; i < 2; i++) if (voc_aux[i]) (void)hipStreamSynchronize(voc_aux[i]);
        front_dirty_main = true; mel_free_pending = false;
    }
    int front_prio = 1;                    // zvx_set_int("front_prio", v): priority of the front stream: 1 = highest the device offers, -1 = lowest, 0 = default
    void front_setup() {
        if (front_stream) return;
        if (front_prio) {
            int least = 0, greatest = 0;                                     // (numerically: greatest priority = the smaller value)
            HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIPCHK(hipStreamCreateWithPriority(&front_stream, hipStreamNonBlocking, front_prio > 0 ? greatest : least));
        } else
        HIPCHK(hipStreamCreateWithFlags(&front_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&ev_front_done, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_mel_free, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_main_join, hipEventDisableTiming));
    }
};

namespace {

GemmArgs gemm_base(int dtype) {
    GemmArgs a;
    memset(&a, 0, sizeof a);
    a.dtype = dtype; a.nbatch = 1; a.nheads = 1; a.ntaps = 1; a.stride = 1; a.wout = 0; a.hin = 1; a.win = 0;
    a.alpha = 1.f; a.out_scale = 1.f; a.out_dtype = dtype; a.res_dtype = dtype;
    return a;
}
void set_taps_1d(GemmArgs& a, int k, int dilation) {
    a.ntaps = k;
    for (int i = 0; i < k; i++) { a.du[i] = 0; a.dv[i] = (int)((i - (k - 1) / 2) * dilation); }
}

// ------------------------------------------------------------------------------------------------
// manifest
// ------------------------------------------------------------------------------------------------
void parse_manifest(zvx_ctx* c, const char* manifest, const void* weights, size_t nbytes) {
    if (!manifest || !weights) fail(ZVX_E_MANIFEST, "manifest/weights pointer is NULL");
    const size_t nfloats = nbytes / 4;
    c->host_blob.assign((const float*)weights, (const float*)weights + nfloats);
    std::stringstream ss(manifest);
    std::string line;
    bool first = true;
    while (std::getline(ss, line)) {
        if (line.empty()) continue;
        std::stringstream ls(line);
        std::string kw; ls >> kw;
        if (first) { if (kw != "zvx_manifest") fail(ZVX_E_MANIFEST, "manifest: bad magic '%s'", kw.c_str()); first = false; continue; }
        if (kw == "cfg") {
            std::string k, v; ls >> k; std::getline(ls, v);
            size_t p = v.find_first_not_of(' ');
            c->cfg[k] = p == std::string::npos ? "" : v.substr(p);
        } else if (kw == "tensor") {
            std::string name, kind; int nd = 0; ls >> name >> kind >> nd;
            Tensor t; t.kind = kind.empty() ? 'p' : kind[0]; t.numel = 1;
            for (int i = 0; i < nd; i++) { int d; ls >> d; t.dims.push_back(d); t.numel *= (size_t)d; }
            ls >> t.off;
            if (ls.fail() || t.off + t.numel > nfloats) fail(ZVX_E_MANIFEST, "manifest: tensor '%s' out of blob bounds", name.c_str());
            c->tensors[name] = t;
        } else {
            fail(ZVX_E_MANIFEST, "manifest: unknown line '%s'", line.c_str());
        }
    }
    if (first) fail(ZVX_E_MANIFEST, "manifest: empty");
}

// phoneme encoder (kept in f32 because it feeds discrete decisions): in the 16-bit mode its static-weight GEMMs run on the 16-bit
// MFMA as 3-plane split products (ops.hip: k_split3), f32-class accuracy at ~5x the f32 MFMA rate.
// Weights [taps][N][K] f32 -> 16-bit [taps][N][wh | wl | wh], fragment-packed like every other slab-kernel weight.  f16: IEEE-half
// planes scaled by 2^s per tensor (".s3h", Tensor.alpha = 2^-s), else bf16 planes (".s3").  Built from the f32 copies on the
// device, at load for the default mode and on the first zvx_set_int("enc_split", other mode).
void build_split_weights(zvx_ctx* c, bool f16) {
    const char* sfx = f16 ? ".s3h" : ".s3";
    if (c->has(std::string("enc.0.wqk") + sfx)) return;
    std::vector<std::pair<std::string, Tensor>> add;
    size_t stotal = 0;
    for (auto& kv : c->tensors) {
        const Tensor& t = kv.second;
        const bool mine = kv.first.rfind("enc.", 0) == 0;      // the variance predictors stay on the exact-f32 MFMA: N = 256 gives the
                                                                 // slab tiling too few workgroups to win, and their outputs are the decisions
        if (!mine || t.kind != 'f' || t.dtype != DT_F32 || t.dims.size() != 3 || (3 * t.dim(2)) % 16 || t.dim(1) % 8) continue;
        Tensor s3 = t;
        s3.kind = 's'; s3.dtype = f16 ? DT_F16 : DT_BF16; s3.dims = {t.dim(0), t.dim(1), 3 * t.dim(2)}; s3.numel = t.numel * 3; s3.host = nullptr;
        stotal += ((s3.numel * 2 + 255) & ~(size_t)255) + ((packed_weight_elems(s3.dim(0), s3.dim(1), s3.dim(2)) * 2 + 255) & ~(size_t)255);
        add.emplace_back(kv.first + sfx, s3);
    }
    char* sarena = stotal ? (char*)c->buf(f16 ? "weights_split_h" : "weights_split", stotal) : nullptr;
    float* mx_d = f16 ? c->fbuf("weights_split_absmax", 64) : nullptr;
    size_t soff = 0;
    for (auto& kv : add) {
        Tensor& s3 = kv.second;
        const Tensor& src = c->tensors[kv.first.substr(0, kv.first.size() - strlen(sfx))];
        float scale = 1.f;
        if (f16) {
            // s: max |w| 2^s in [2^14, 2^15) -- the hi plane uses the top of half's range, wl = w 2^s - wh stays normal down to
            // |w| ~ 2^-9 max |w| and the third plane (w 2^(s-11)) down to 2^-17 max |w|; below that the planes lose bits of an
            // element whose contribution is already < 2^-24 of the row's largest
            float mx = 0.f;
            launch_absmax((const float*)src.dev, src.numel, mx_d, c->stream);
            HIPCHK(hipMemcpyAsync(&mx, mx_d, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            if (!(mx > 0.f) || !std::isfinite(mx)) mx = 1.f;
            int e = 0; (void)frexpf(mx, &e);                 // mx = m 2^e, m in [0.5, 1)
            int sh = 15 - e; sh = std::max(-24, std::min(sh, 40));
            scale = ldexpf(1.f, sh);
            s3.alpha = ldexpf(1.f, -sh);
        }
        s3.dev = sarena + soff; soff += (s3.numel * 2 + 255) & ~(size_t)255;
        launch_split3_weights((const float*)src.dev, s3.dev, (long)src.dim(0) * src.dim(1), src.dim(2), c->stream, f16 ? 1 : 0, scale);
        void* pk = sarena + soff; soff += (packed_weight_elems(s3.dim(0), s3.dim(1), s3.dim(2)) * 2 + 255) & ~(size_t)255;
        launch_pack_weights(s3.dev, s3.dim(0), s3.dim(1), s3.dim(2), pk, c->stream);
        c->packed[s3.dev] = pk;
        c->tensors[kv.first] = s3;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
}

void upload_weights(zvx_ctx* c) {
    // one device arena for all tensors; 'w' tensors are converted to the context precision on device
    size_t total = 0;
    for (auto& kv : c->tensors) {
        Tensor& t = kv.second;
        t.dtype = (t.kind == 'w') ? c->dt : DT_F32;
        total += (t.numel * dtype_size(t.dtype) + 255) & ~(size_t)255;
    }
    char* arena = (char*)c->buf("weights", total);
    float* staging = c->fbuf("weights_staging", c->host_blob.size());
    HIPCHK(hipMemcpyAsync(staging, c->host_blob.data(), c->host_blob.size() * 4, hipMemcpyHostToDevice, c->stream));
    size_t off = 0;
    for (auto& kv : c->tensors) {
        Tensor& t = kv.second;
        t.dev = arena + off;
        t.host = c->host_blob.data() + t.off;
        launch_cast(staging + t.off, DT_F32, t.dev, t.dtype, t.numel, c->stream);
        off += (t.numel * dtype_size(t.dtype) + 255) & ~(size_t)255;
    }
    // fragment-order copies of every bf16 contraction weight [taps][N][K] with K % 16 == 0
    size_t ptotal = 0;
    for (auto& kv : c->tensors) {
        Tensor& t = kv.second;
        if (t.kind == 'w' && t.dtype == DT_BF16 && t.dims.size() == 3 && t.dim(2) % 8 == 0)
            ptotal += (packed_weight_elems(t.dim(0), t.dim(1), t.dim(2)) * 2 + 255) & ~(size_t)255;
    }
    if (ptotal) {
        char* parena = (char*)c->buf("weights_packed", ptotal);
        size_t poff = 0;
        for (auto& kv : c->tensors) {
            Tensor& t = kv.second;
            if (t.kind == 'w' && t.dtype == DT_BF16 && t.dims.size() == 3 && t.dim(2) % 8 == 0) {
                launch_pack_weights(t.dev, t.dim(0), t.dim(1), t.dim(2), parena + poff, c->stream);
                c->packed[t.dev] = parena + poff;
                poff += (packed_weight_elems(t.dim(0), t.dim(1), t.dim(2)) * 2 + 255) & ~(size_t)255;
            }
        }
    }
    // The mel decoders in IEEE half (bf16 mode): a second, f16 copy of their convolution / projection weights (cast from the f32 blob, fragment-packed
    // like the bf16 ones).  Same MFMA rate, 8x smaller rounding error on weights and activations (every tensor there sits behind a
    // norm: O(1..100), nowhere near 65504; stores saturate).  zvx_set_int("dec_f16", 0) runs the bf16 copies (A/B).
    if (c->dt == DT_BF16) {
        std::vector<std::pair<std::string, Tensor>> add16;
        size_t htotal = 0;
        for (auto& kv : c->tensors) {
            const Tensor& t = kv.second;
            // (round 5: the HiFi-GAN vocoder's weights too -- "voc.": its 16-bit tensors are IEEE half unless zvx_set_int("voc_f16", 0))
            if ((kv.first.rfind("sty.", 0) != 0 && kv.first.rfind("dec.", 0) != 0 && kv.first.rfind("voc.", 0) != 0) || t.kind != 'w' || t.dtype != DT_BF16 || t.dims.size() != 3 || t.dim(2) % 8) continue;
            Tensor hcopy = t; hcopy.dtype = DT_F16; hcopy.kind = 'h';
            htotal += ((t.numel * 2 + 255) & ~(size_t)255) + ((packed_weight_elems(t.dim(0), t.dim(1), t.dim(2)) * 2 + 255) & ~(size_t)255);
            add16.emplace_back(kv.first + ".h16", hcopy);
        }
        char* harena = htotal ? (char*)c->buf("weights_f16", htotal) : nullptr;
        size_t hoff = 0;
        for (auto& kv : add16) {
            Tensor& t = kv.second;
            t.dev = harena + hoff; hoff += (t.numel * 2 + 255) & ~(size_t)255;
            launch_cast(staging + t.off, DT_F32, t.dev, DT_F16, t.numel, c->stream);
            void* pk = harena + hoff; hoff += (packed_weight_elems(t.dim(0), t.dim(1), t.dim(2)) * 2 + 255) & ~(size_t)255;
            launch_pack_weights(t.dev, t.dim(0), t.dim(1), t.dim(2), pk, c->stream);
            c->packed[t.dev] = pk;
            c->tensors[kv.first] = t;
        }
    }
    // StyleTTS residual blocks with a learned shortcut (styletts.py:60-68, 130-139): out = (conv2(t) + conv1x1(x)) / sqrt(2).  The
    // shortcut runs INSIDE the k = 3 convolution as a second source of its K loop (GemmArgs::X2), which needs the two fragment
    // streams interleaved per 32-channel tile -- built once here, for the bf16 and the half copies
    for (const char* blk : {"sty.enc0", "sty.dec0", "sty.dec1", "sty.dec2"})
        for (const char* sfx : {"", ".h16"}) {
            const std::string n2 = std::string(blk) + ".c2" + sfx, ns = std::string(blk) + ".sc" + sfx;
            if (!c->has(n2) || !c->has(ns) || c->has(std::string(blk) + ".sc_b")) continue;
            const Tensor& t2 = c->t(n2); const Tensor& ts = c->t(ns);
            if (t2.dims.size() != 3 || ts.dims.size() != 3 || ts.dim(0) != 1 || ts.dim(1) != t2.dim(1) || !c->packed.count(t2.dev) || !c->packed.count(ts.dev)) continue;
            const size_t bytes = (packed_weight_elems(t2.dim(0), t2.dim(1), t2.dim(2)) + packed_weight_elems(1, ts.dim(1), ts.dim(2))) * 2;
            void* comb = c->buf("weights_pair." + n2, bytes);
            launch_pack_pair(c->packed[t2.dev], t2.dim(0), t2.dim(2), c->packed[ts.dev], 1, ts.dim(2), t2.dim(1), comb, c->stream);
            c->pair_packed[t2.dev] = comb;
        }
    // f32 FFT blocks: Q, K and V projections as ONE GEMM (fs2.py:143-145) -- [Wq; Wk; Wv] and the biases concatenated once here
    {
        std::vector<std::pair<std::string, Tensor>> add;
        for (auto& kv : c->tensors) {
            const std::string& nm = kv.first;
            if (nm.size() < 5 || nm.compare(nm.size() - 4, 4, ".wqk") != 0) continue;          // every f32 FFT block: the encoder's, and the FS2 decoder's in f32 mode
            const std::string pre = nm.substr(0, nm.size() - 4);
            const Tensor& wqk = kv.second;
            if (!c->has(pre + ".wv") || !c->has(pre + ".bqk") || !c->has(pre + ".bv") || wqk.dtype != DT_F32 || wqk.dims.size() != 3) continue;
            const Tensor& wv = c->t(pre + ".wv");
            if (wv.dtype != DT_F32 || wv.dims.size() != 3 || wv.dim(2) != wqk.dim(2)) continue;
            Tensor w = wqk; w.dims = {1, wqk.dim(1) + wv.dim(1), wqk.dim(2)}; w.numel = wqk.numel + wv.numel; w.host = nullptr;
            w.dev = c->buf(pre + ".wqkv", w.numel * 4);
            HIPCHK(hipMemcpyAsync(w.dev, wqk.dev, wqk.numel * 4, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(hipMemcpyAsync((float*)w.dev + wqk.numel, wv.dev, wv.numel * 4, hipMemcpyDeviceToDevice, c->stream));
            const Tensor& bqk = c->t(pre + ".bqk"); const Tensor& bv = c->t(pre + ".bv");
            Tensor bb = bqk; bb.dims = {(int)(bqk.numel + bv.numel)}; bb.numel = bqk.numel + bv.numel; bb.host = nullptr;
            bb.dev = c->buf(pre + ".bqkv", bb.numel * 4);
            HIPCHK(hipMemcpyAsync(bb.dev, bqk.dev, bqk.numel * 4, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(hipMemcpyAsync((float*)bb.dev + bqk.numel, bv.dev, bv.numel * 4, hipMemcpyDeviceToDevice, c->stream));
            add.emplace_back(pre + ".wqkv", w); add.emplace_back(pre + ".bqkv", bb);
        }
        for (auto& kv : add) c->tensors[kv.first] = kv.second;
    }
    // half FFT blocks (the FS2 / SCLN decoder in the 16-bit mode): Q, K and V projections as ONE GEMM too (round 6) -- [Wq; Wk; Wv] in IEEE half,
    // fragment-packed, and the concatenated f32 biases
    {
        std::vector<std::pair<std::string, Tensor>> add;
        for (auto& kv : c->tensors) {
            const std::string& nm = kv.first;
            const std::string sfx = ".wqk.h16";
            if (nm.size() <= sfx.size() || nm.compare(nm.size() - sfx.size(), sfx.size(), sfx) != 0) continue;
            const std::string pre = nm.substr(0, nm.size() - sfx.size());
            if (!c->has(pre + ".wv.h16") || !c->has(pre + ".bqk") || !c->has(pre + ".bv")) continue;
            const Tensor& wqk = kv.second; const Tensor& wv = c->t(pre + ".wv.h16");
            if (wqk.dims.size() != 3 || wv.dims.size() != 3 || wqk.dim(0) != 1 || wv.dim(0) != 1 || wv.dim(2) != wqk.dim(2) || wqk.dim(2) % 8) continue;
            Tensor w = wqk; w.dims = {1, wqk.dim(1) + wv.dim(1), wqk.dim(2)}; w.numel = wqk.numel + wv.numel; w.host = nullptr;
            w.dev = c->buf(pre + ".wqkv.h16", w.numel * 2);
            HIPCHK(hipMemcpyAsync(w.dev, wqk.dev, wqk.numel * 2, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(hipMemcpyAsync((char*)w.dev + wqk.numel * 2, wv.dev, wv.numel * 2, hipMemcpyDeviceToDevice, c->stream));
            void* pk = c->buf(pre + ".wqkv.h16.packed", packed_weight_elems(1, w.dim(1), w.dim(2)) * 2);
            launch_pack_weights(w.dev, 1, w.dim(1), w.dim(2), pk, c->stream);
            c->packed[w.dev] = pk;
            add.emplace_back(pre + ".wqkv.h16", w);
            if (!c->has(pre + ".bqkv")) {
                const Tensor& bqk = c->t(pre + ".bqk"); const Tensor& bv = c->t(pre + ".bv");
                Tensor bb = bqk; bb.dims = {(int)(bqk.numel + bv.numel)}; bb.numel = bqk.numel + bv.numel; bb.host = nullptr;
                bb.dev = c->buf(pre + ".bqkv", bb.numel * 4);
                HIPCHK(hipMemcpyAsync(bb.dev, bqk.dev, bqk.numel * 4, hipMemcpyDeviceToDevice, c->stream));
                HIPCHK(hipMemcpyAsync((float*)bb.dev + bqk.numel, bv.dev, bv.numel * 4, hipMemcpyDeviceToDevice, c->stream));
                add.emplace_back(pre + ".bqkv", bb);
            }
        }
        for (auto& kv : add) c->tensors[kv.first] = kv.second;
    }
    if (c->enc_split) build_split_weights(c, c->enc_split == 2);
    HIPCHK(hipStreamSynchronize(c->stream));
    DevBuf& st = c->bufs["weights_staging"];
    HIPCHK(hipFree(st.base)); st.base = nullptr; st.p = nullptr; st.cap = 0;
}

void read_config(zvx_ctx* c) {
    auto it = c->cfg.find("precision");
    if (it == c->cfg.end()) fail(ZVX_E_MANIFEST, "manifest: missing cfg 'precision'");
    if (it->second == "bf16") c->dt = DT_BF16; else if (it->second == "f32") c->dt = DT_F32;
    else fail(ZVX_E_MANIFEST, "manifest: unknown precision '%s'", it->second.c_str());
    c->enc_split = c->dt == DT_BF16 ? (c->cfg.count("enc_split") ? std::max(0, std::min(2, c->cfg_int("enc_split"))) : 2) : 0;
    c->H = c->cfg_int("hidden"); c->emb_dim = c->cfg_int("emb_dim"); c->punct_dim = c->cfg_int("punct_dim");
    c->n_phone_rows = c->cfg_int("n_phone_rows"); c->n_punct_rows = c->cfg_int("n_punct_rows");
    c->max_txt_len = c->cfg_int("max_txt_len"); c->max_mel_len = c->cfg_int("max_mel_len");
    c->enc_layers = c->cfg_int("enc_layers"); c->enc_heads = c->cfg_int("enc_heads"); c->ffn_dim = c->cfg_int("ffn_dim");
    auto fk = c->cfg_list("ffn_k"); if (fk.size() != 2) fail(ZVX_E_MANIFEST, "manifest: ffn_k needs 2 entries");
    c->ffn_k0 = fk[0]; c->ffn_k1 = fk[1];
    c->vp_dim = c->cfg_int("vp_dim"); c->vp_k = c->cfg_int("vp_k"); c->n_bins = c->cfg_int("n_bins");
    c->dec_layers = c->cfg_int("dec_layers"); c->dec_heads = c->cfg_int("dec_heads"); c->dec_scln = c->cfg_int("dec_scln");
    c->n_mels = c->cfg_int("n_mels"); c->hop = c->cfg_int("hop"); c->res_dim = c->cfg_int("res_dim");
    c->rn_layers = c->cfg_list("rn_layers"); c->rn_filters = c->cfg_list("rn_filters"); c->rn_asp = c->cfg_int("rn_asp");
    const std::string dk = c->cfg["dec_kind"];
    if (dk == "fastspeech2") c->dec_kind = 0; else if (dk == "styletts") c->dec_kind = 1;
    else fail(ZVX_E_MANIFEST, "unknown decoder kind: '%s'", dk.c_str());                 // model.py:244
    c->voc_resblock = c->cfg_int("voc_resblock"); c->voc_c0 = c->cfg_int("voc_c0");
    c->voc_rates = c->cfg_list("voc_rates"); c->voc_ksizes = c->cfg_list("voc_ksizes"); c->voc_rb_k = c->cfg_list("voc_rb_k");
    {
        std::stringstream ss(c->cfg["voc_rb_d"]); std::string grp;
        while (std::getline(ss, grp, ';')) {
            std::vector<int> v; std::stringstream gs(grp); std::string tok;
            while (std::getline(gs, tok, ',')) if (!tok.empty()) v.push_back(atoi(tok.c_str()));
            c->voc_rb_d.push_back(v);
        }
    }
    int prod = 1; for (int r : c->voc_rates) prod *= r;
    if (prod != c->hop) fail(ZVX_E_MANIFEST, "prod(voc_rates)=%d != hop=%d", prod, c->hop);
    if (c->voc_rb_d.size() != c->voc_rb_k.size()) fail(ZVX_E_MANIFEST, "voc_rb_d / voc_rb_k size mismatch");
    if (c->H % 8 || (c->H / c->enc_heads) % 8 || (c->H / c->dec_heads) % 8) fail(ZVX_E_UNSUPPORTED, "hidden/head dims must be multiples of 8");
}

// ------------------------------------------------------------------------------------------------
// FFT block (fs2.py:221-230): x [B][Lmax][H] (dtype dt) -> x, in place.
// ------------------------------------------------------------------------------------------------
struct FftWeights { std::string p; bool scln; const float* bg; long bg_bs; const float* post_add; };

// Ls: rows per utterance of x and of every row-major buffer below (>= Lmax).  Ls > Lmax + the k = 9 convolution's halo lets the
// static-weight GEMMs run batch-flattened (GemmArgs::bflat; `flat`): the FS2 decoder's 896 frames are 3.5 tiles of 256 rows.
void fft_block(zvx_ctx* c, void* x, int dt, int B, int Lmax, const int* len_dev, int nheads, const FftWeights& w, int Ls = 0, bool flat = false) {
    if (Ls <= 0) Ls = Lmax;
    const int H = c->H, d = H / nheads, Lp = (Lmax + 7) & ~7, F = c->ffn_dim;
    const size_t es = dtype_size(dt);
    const bool h16 = dt == DT_F16;                                               // the 16-bit tensors of this block are IEEE half (weights: the ".h16" copies)
    auto wdev = [&](const char* n) { return c->t(w.p + n + (h16 ? ".h16" : "")).dev; };
    void* qk = c->buf("fft.qk", (size_t)B * Ls * 2 * H * es);
    void* vt = c->buf("fft.vt", (size_t)B * H * Lp * es);
    float* sc = c->fbuf("fft.scores", (size_t)B * nheads * Lmax * Lp);
    void* P = c->buf("fft.P", (size_t)B * nheads * Lmax * Lp * es);
    void* o = c->buf("fft.o", (size_t)B * Ls * H * es);
    float* y = c->fbuf("fft.y", (size_t)B * Ls * H);
    // half blocks (the FS2 / SCLN decoder in the 16-bit mode; round 6): the pre-norm sums y = fc(O) + x and y = conv_k1(h) + x leave their GEMMs
    // as IEEE half through the compile-time decoder epilogue (raw 16-bit residual, ZVX_EPI_DEC) instead of f32 through the run-time one, and the
    // LayerNorm / SCLN pass reads 2 bytes per element instead of 4: one more rounding at 2^-11 of a value the norm then rescales
    // (zvx_set_int("dec_y16", 0): f32, A/B)
    const bool y16 = h16 && c->dec_y16;
    void* const yv = y16 ? c->buf("fft.y16", (size_t)B * Ls * H * 2) : (void*)y;
    const int ydt = y16 ? (int)DT_F16 : (int)DT_F32;
    void* hbuf = c->buf("fft.h", (size_t)B * Ls * F * es);

    // f32 blocks (phoneme encoder) in bf16 mode: the static-weight GEMMs take bf16 split planes [hi | hi | lo] of their f32 input
    // against [wh | wl | wh] weights (K = 3 x the logical K): f32-class results from the bf16 MFMA
    const bool sp16 = c->enc_split == 2;                                          // IEEE-half planes (f32-class to 2^-24) / bf16 planes (2^-17)
    const char* const s3 = sp16 ? ".s3h" : ".s3";
    const bool split = dt == DT_F32 && c->enc_split && c->has(w.p + ".wqk" + s3);
    void* xs = split ? c->buf("fft.xs", (size_t)B * Ls * 3 * H * 2) : nullptr;
    auto split_of = [&](const float* src, int C, void* dst) { launch_split3(src, C, dst, B, Ls, len_dev, C, c->stream, sp16); };
    auto as_split = [&](GemmArgs& a, const void* planes, int C, const std::string& wname) {      // operand swap: same GEMM, 3-plane K axis
        const Tensor& ws = c->t(wname + s3);
        a.dtype = sp16 ? DT_F16 : DT_BF16; a.X = planes; a.x_bs = (long)Ls * 3 * C; a.ldx = 3 * C; a.K = 3 * C;
        a.alpha = ws.alpha;                                                                        // half planes carry w 2^s
        a.W = ws.dev; a.ldw = 3 * C; a.w_ts = (long)ws.dim(1) * 3 * C;
        a.flops = 2.0 * (double)a.M * a.nbatch * a.N * C * a.ntaps;                                 // algorithmic (f32) work, not the 3x issued
    };
    // exact-f32 blocks with merged projection weights: one Q | K | V GEMM + one fused attention launch (attention.hip)
    AttnF32Args af;
    memset(&af, 0, sizeof af);
    af.ld = 3 * H; af.bs = (long)Ls * 3 * H; af.q_off = 0; af.k_off = H; af.v_off = 2 * H;
    af.out = (float*)o; af.o_bs = (long)Ls * H; af.ldo = H; af.len = len_dev; af.L = Lmax; af.D = d; af.nheads = nheads; af.nbatch = B;
    af.scale = (float)(1.0 / pow((double)d, 0.5));
    const bool fused_f32 = dt == DT_F32 && c->use_attn_f32 && c->has(w.p + ".wqkv") && launch_attention_f32(af, c->stream, true);
    // split planes of a GEMM input come from its PRODUCER where that is one of ours (LayerNorm, the fused attention, the k = 9
    // convolution's epilogue); k_split3 runs only for the block input of the first layer and on the unfused attention path
    const bool xs_from_producer = split && c->fft_xs_ready == x;
    if (split && !xs_from_producer) split_of((const float*)x, H, xs);
    c->fft_xs_ready = nullptr;
    if (fused_f32) {
        float* qkv = c->fbuf("fft.qkv", (size_t)B * Ls * 3 * H);
        GemmArgs a = gemm_base(dt);
        a.X = x; a.x_bs = (long)Ls * H; a.ldx = H; a.W = c->t(w.p + ".wqkv").dev; a.ldw = H;
        a.M = Lmax; a.N = 3 * H; a.K = H; a.nbatch = B; a.in_len = len_dev; a.out_len = len_dev; if (flat) a.bflat = Ls;
        if (split) { as_split(a, xs, H, w.p + ".wqkv"); a.out_dtype = DT_F32; }
        a.bias = c->pf(w.p + ".bqkv"); a.bias_mode = 1;
        a.out = qkv; a.o_bs = (long)Ls * 3 * H; a.ldo = 3 * H;
        c->gemm(a);
        af.qkv = qkv;
        if (split) { af.planes = (unsigned short*)xs; af.planes_C = H; af.planes_f16 = sp16; }   // o as split planes for the output projection
        c->timed(4.0 * B * nheads * (double)Lmax * Lmax * d, (double)B * Lmax * 4.0 * H * 4, [&] { launch_attention_f32(af, c->stream, false); });
    } else {
    if (split) split_of((const float*)x, H, xs);
    const bool qkv16 = h16 && c->dec_qkv && c->has(w.p + ".wqkv.h16") && c->has(w.p + ".bqkv");
    void* qkvbuf = qkv16 ? c->buf("fft.qkv16", (size_t)B * Ls * 3 * H * es) : nullptr;
    if (qkv16) {   // half blocks: [Q | K | V] = x [Wq; Wk; Wv]^T + b in ONE launch (round 6), then the 16-bit transpose of its V columns
        GemmArgs a = gemm_base(dt);
        a.X = x; a.x_bs = (long)Ls * H; a.ldx = H; a.W = c->t(w.p + ".wqkv.h16").dev; a.ldw = H;
        a.M = Lmax; a.N = 3 * H; a.K = H; a.nbatch = B; a.in_len = len_dev; a.out_len = len_dev; if (flat) a.bflat = Ls;
        a.bias = c->pf(w.p + ".bqkv"); a.bias_mode = 1;
        a.out = qkvbuf; a.o_bs = (long)Ls * 3 * H; a.ldo = 3 * H;
        c->gemm(a);
        c->timed(0, (double)B * Lmax * H * 4.0, [&] { launch_transpose16((const char*)qkvbuf + (size_t)2 * H * es, 3 * H, vt, Lp, B, Ls, H, c->stream, len_dev); });
    } else
    {   // [Q | K] = x Wqk^T + b                                       fs2.py:143-144
        GemmArgs a = gemm_base(dt);
        a.X = x; a.x_bs = (long)Ls * H; a.ldx = H; a.W = wdev(".wqk"); a.ldw = H;
        a.M = Lmax; a.N = 2 * H; a.K = H; a.nbatch = B; a.in_len = len_dev; a.out_len = len_dev; if (flat) a.bflat = Ls;
        if (split) { as_split(a, xs, H, w.p + ".wqk"); a.out_dtype = DT_F32; }
        a.bias = c->pf(w.p + ".bqk"); a.bias_mode = 1;
        a.out = qk; a.o_bs = (long)Ls * 2 * H; a.ldo = 2 * H;
        c->gemm(a);
    }
    if (qkv16) {
    } else if (h16) {   // half: V = x Wv^T + b on the conv-slab kernel (static weights), then one 16-bit transpose to the key-contiguous layout
        void* vrow = hbuf;                                                       // [B][Lmax][H]: the FFN buffer is idle here
        GemmArgs a = gemm_base(dt);
        a.X = x; a.x_bs = (long)Ls * H; a.ldx = H; a.W = wdev(".wv"); a.ldw = H;
        a.M = Lmax; a.N = H; a.K = H; a.nbatch = B; a.in_len = len_dev; a.out_len = len_dev; if (flat) a.bflat = Ls;
        a.bias = c->pf(w.p + ".bv"); a.bias_mode = 1;
        a.out = vrow; a.o_bs = (long)Ls * H; a.ldo = H;
        c->gemm(a);
        c->timed(0, (double)B * Lmax * H * 4.0, [&] { launch_transpose16(vrow, H, vt, Lp, B, Ls, H, c->stream, len_dev); });   // V^T columns >= len[b]: zeros (stale rows of the shared FFN buffer)
    } else
    {   // V^T[h*d + j][l] = Wv x^T + b  (stored transposed so that P.V is K-contiguous)   fs2.py:145
        GemmArgs a = gemm_base(dt);
        a.X = c->t(w.p + ".wv").dev; a.x_bs = 0; a.ldx = H; a.W = x; a.w_bs = (long)Ls * H; a.ldw = H;
        a.M = H; a.N = Lmax; a.K = H; a.nbatch = B; a.in_len_static = H;
        a.bias = c->pf(w.p + ".bv"); a.bias_mode = 2;
        a.out = vt; a.o_bs = (long)H * Lp; a.ldo = Lp;
        // columns in [len, Lmax) hold finite junk (x rows beyond len are never NaN: buffers start zeroed and
        // only ever receive finite values); P is exactly zero there, so they never contribute.  The pad columns [Lmax, Lp) are not
        // written by this GEMM and the buffer is shared between the f32 encoder and the bf16 decoder (stale bits of one can read
        // as Inf / NaN in the other, and 0 * NaN = NaN in the P.V product): zero them
        if (Lp > Lmax) HIPCHK(hipMemset2DAsync((char*)vt + (size_t)Lmax * es, (size_t)Lp * es, 0, (size_t)(Lp - Lmax) * es, (size_t)B * H, c->stream));
        c->gemm(a);
        // ... and round 5: the columns [len, Lmax) as well -- "finite junk" held only as long as every row of x past an
        // utterance's length had been defined by something; nothing guarantees that (tests: zvx_set_int("poison_pads", 1))
        launch_zero_tail_cols(vt, (int)es, Lp, (long)H * Lp, B, H, Lmax, len_dev, c->stream);
    }
    FlashArgs fa;
    memset(&fa, 0, sizeof fa);
    fa.qk = qk; fa.qk_bs = (long)Ls * 2 * H; fa.ldq = 2 * H; fa.k_off = H; fa.vt = vt;
    if (qkv16) { fa.qk = qkvbuf; fa.qk_bs = (long)Ls * 3 * H; fa.ldq = 3 * H; } fa.vt_bs = (long)H * Lp; fa.ldv = Lp;
    fa.out = o; fa.o_bs = (long)Ls * H; fa.ldo = H; fa.len = len_dev; fa.L = Lmax; fa.D = d; fa.nheads = nheads; fa.nbatch = B;
    fa.scale = (float)(1.0 / pow((double)d, 0.5));
    fa.f16 = h16;
    const bool flash = (dt == DT_BF16 || h16) && c->use_flash && launch_flash_attention(fa, c->stream, true);
    if (h16 && !flash) fail(ZVX_E_UNSUPPORTED, "half-precision FFT block without the fused attention (head depth %d)", d);
    if (flash) {
        // softmax(Q K^T / sqrt(d)) V in one launch, scores and probabilities stay on chip            fs2.py:47-58
        c->timed(4.0 * B * nheads * (double)Lmax * Lmax * d, (double)B * Lmax * (3.0 * H + H) * es, [&] { launch_flash_attention(fa, c->stream, false); });
        c->sat_scan(o, dt, (long)Ls * H, H, B, Lmax, len_dev, H);
    } else {
    {   // scores = Q K^T / sqrt(d)                                    fs2.py:49-50
        GemmArgs a = gemm_base(dt);
        a.X = qk; a.x_bs = (long)Ls * 2 * H; a.x_hs = d; a.ldx = 2 * H;
        a.W = (const char*)qk + (size_t)H * es; a.w_bs = (long)Ls * 2 * H; a.w_hs = d; a.ldw = 2 * H;
        a.M = Lmax; a.N = Lmax; a.K = d; a.nbatch = B; a.nheads = nheads; a.in_len = len_dev; a.out_len = len_dev;
        a.alpha = (float)(1.0 / pow((double)d, 0.5));
        a.out = sc; a.out_dtype = DT_F32; a.o_bs = (long)nheads * Lmax * Lp; a.o_hs = (long)Lmax * Lp; a.ldo = Lp;
        a.flops = 2.0 * B * nheads * (double)Lmax * Lmax * d;
        c->gemm(a);
    }
    c->timed(0, (double)B * nheads * Lmax * Lp * (4.0 + es), [&] { launch_softmax_rows(sc, Lp, P, dt, Lp, B, nheads, Lmax, len_dev, c->stream); });      // fs2.py:52-55
    {   // O = P V                                                      fs2.py:56
        GemmArgs a = gemm_base(dt);
        a.X = P; a.x_bs = (long)nheads * Lmax * Lp; a.x_hs = (long)Lmax * Lp; a.ldx = Lp;
        a.W = vt; a.w_bs = (long)H * Lp; a.w_hs = (long)d * Lp; a.ldw = Lp;
        a.M = Lmax; a.N = d; a.K = Lp; a.k_len = len_dev; a.nbatch = B; a.nheads = nheads; a.in_len = len_dev; a.out_len = len_dev;
        a.out = o; a.o_bs = (long)Ls * H; a.o_hs = d; a.ldo = H;
        a.flops = 2.0 * B * nheads * (double)Lmax * Lmax * d;
        c->gemm(a);
    }
    }
    }
    {   // y = fc(O) + residual                                         fs2.py:158-162
        GemmArgs a = gemm_base(dt);
        a.X = o; a.x_bs = (long)Ls * H; a.ldx = H; a.W = wdev(".wo"); a.ldw = H;
        a.M = Lmax; a.N = H; a.K = H; a.nbatch = B; a.in_len = len_dev; a.out_len = len_dev; if (flat) a.bflat = Ls;
        a.bias = c->pf(w.p + ".bo"); a.bias_mode = 1;
        a.res = x; a.r_bs = (long)Ls * H; a.ldr = H; a.res_mode = 1; a.res_dtype = dt;
        a.out = yv; a.out_dtype = ydt; a.o_bs = (long)Ls * H; a.ldo = H;
        if (split) { if (!fused_f32) split_of((const float*)o, H, xs); as_split(a, xs, H, w.p + ".wo"); }
        c->gemm(a);
    }
    const double ln_bytes = (double)B * Lmax * H * ((y16 ? 2.0 : 4.0) + es);
    c->timed(0, ln_bytes, [&] {
        if (w.scln) launch_layernorm(yv, ydt, H, x, dt, H, B, Ls, len_dev, H, 1, 1e-8f, nullptr, nullptr, w.bg, w.bg_bs, nullptr, c->stream);
        else launch_layernorm(yv, ydt, H, x, dt, H, B, Ls, len_dev, H, 0, 1e-5f, c->pf(w.p + ".ln1_g"), c->pf(w.p + ".ln1_b"), nullptr, 0, nullptr, c->stream, split ? xs : nullptr, sp16);
    });
    c->sat_scan(x, dt, (long)Ls * H, H, B, Lmax, len_dev, H);
    {   // h = relu(conv_k9(x))                                         fs2.py:198-200
        GemmArgs a = gemm_base(dt);
        a.X = x; a.x_bs = (long)Ls * H; a.ldx = H; a.W = wdev(".w1"); a.ldw = H; a.w_ts = (long)F * H;
        a.M = Lmax; a.N = F; a.K = H; a.nbatch = B; a.in_len = len_dev; a.out_len = len_dev; if (flat) a.bflat = Ls;
        set_taps_1d(a, c->ffn_k0, 1);
        a.bias = c->pf(w.p + ".b1"); a.bias_mode = 1; a.act = ACT_RELU;
        a.out = hbuf; a.o_bs = (long)Ls * F; a.ldo = F;
        void* hs = split ? c->buf("fft.hs", (size_t)B * Ls * 3 * F * 2) : nullptr;
        if (split) {
            if (w.scln) split_of((const float*)x, H, xs);                           // (SCLN blocks are never f32 + split today; kept correct)
            as_split(a, xs, H, w.p + ".w1");
            a.out_dtype = DT_F32; a.out_split3 = sp16 ? 2 : 1; a.out = hs; a.o_bs = (long)Ls * 3 * F; a.ldo = 3 * F;   // h straight into the planes of the k = 1 convolution
        }
        c->gemm(a);
    }
    {   // y = conv_k1(h) + residual                                    fs2.py:201-207
        GemmArgs a = gemm_base(dt);
        a.X = hbuf; a.x_bs = (long)Ls * F; a.ldx = F; a.W = wdev(".w2"); a.ldw = F; a.w_ts = (long)H * F;
        a.M = Lmax; a.N = H; a.K = F; a.nbatch = B; a.in_len = len_dev; a.out_len = len_dev; if (flat) a.bflat = Ls;
        set_taps_1d(a, c->ffn_k1, 1);
        a.bias = c->pf(w.p + ".b2"); a.bias_mode = 1;
        a.res = x; a.r_bs = (long)Ls * H; a.ldr = H; a.res_mode = 1; a.res_dtype = dt;
        a.out = yv; a.out_dtype = ydt; a.o_bs = (long)Ls * H; a.ldo = H;
        if (split) as_split(a, c->buf("fft.hs", (size_t)B * Ls * 3 * F * 2), F, w.p + ".w2");
        c->gemm(a);
    }
    c->timed(0, ln_bytes, [&] {
        if (w.scln) launch_layernorm(yv, ydt, H, x, dt, H, B, Ls, len_dev, H, 1, 1e-8f, nullptr, nullptr, w.bg + 2 * H, w.bg_bs, w.post_add, c->stream);
        else launch_layernorm(yv, ydt, H, x, dt, H, B, Ls, len_dev, H, 0, 1e-5f, c->pf(w.p + ".ln2_g"), c->pf(w.p + ".ln2_b"), nullptr, 0, w.post_add, c->stream, split ? xs : nullptr, sp16);
    });
    c->sat_scan(x, dt, (long)Ls * H, H, B, Lmax, len_dev, H);
    if (split && !w.scln) c->fft_xs_ready = x;                                       // the next block on the same buffer finds its input planes in fft.xs
}

// ------------------------------------------------------------------------------------------------
// encoder + variance adaptor + length regulator   (fs2.py:732-775)
// ------------------------------------------------------------------------------------------------
void variance_predictor(zvx_ctx* c, const char* nm, const float* x, int B, int Tmax, const int* T_dev, float* pred, const char* bufsfx = "") {
    const int H = c->H, Fv = c->vp_dim;
    const std::string p = std::string("va.") + nm;
    float* h1 = c->fbuf(std::string("va.h1") + bufsfx, (size_t)B * Tmax * Fv);      // (own buffers when it runs beside another predictor)
    float* h2 = c->fbuf(std::string("va.h2") + bufsfx, (size_t)B * Tmax * Fv);
    {
        GemmArgs a = gemm_base(DT_F32);
        a.X = x; a.x_bs = (long)Tmax * H; a.ldx = H; a.W = c->t(p + ".c1").dev; a.ldw = H; a.w_ts = (long)Fv * H;
        a.M = Tmax; a.N = Fv; a.K = H; a.nbatch = B; a.in_len = T_dev; a.out_len = T_dev;
        set_taps_1d(a, c->vp_k, 1);
        a.bias = c->pf(p + ".b1"); a.bias_mode = 1; a.act = ACT_RELU;
        a.out = h1; a.o_bs = (long)Tmax * Fv; a.ldo = Fv;
        c->gemm(a);
    }
    launch_layernorm(h1, DT_F32, Fv, h1, DT_F32, Fv, B, Tmax, T_dev, Fv, 0, 1e-5f, c->pf(p + ".ln1_g"), c->pf(p + ".ln1_b"), nullptr, 0, nullptr, c->stream);
    {
        GemmArgs a = gemm_base(DT_F32);
        a.X = h1; a.x_bs = (long)Tmax * Fv; a.ldx = Fv; a.W = c->t(p + ".c2").dev; a.ldw = Fv; a.w_ts = (long)Fv * Fv;
        a.M = Tmax; a.N = Fv; a.K = Fv; a.nbatch = B; a.in_len = T_dev; a.out_len = T_dev;
        // fs2.py:543 hard-codes padding=1: taps are k - 1 for k in [0, vp_k)
        a.ntaps = c->vp_k;
        for (int k = 0; k < c->vp_k; k++) { a.du[k] = 0; a.dv[k] = (int)(k - 1); }
        a.bias = c->pf(p + ".b2"); a.bias_mode = 1; a.act = ACT_RELU;
        a.out = h2; a.o_bs = (long)Tmax * Fv; a.ldo = Fv;
        c->gemm(a);
    }
    launch_layernorm(h2, DT_F32, Fv, h2, DT_F32, Fv, B, Tmax, T_dev, Fv, 0, 1e-5f, c->pf(p + ".ln2_g"), c->pf(p + ".ln2_b"), nullptr, 0, nullptr, c->stream);
    launch_rowdot(h2, Fv, c->pf(p + ".lw"), c->t(p + ".lb").host[0], pred, B, Tmax, T_dev, Fv, c->stream);
}

void run_encode(zvx_ctx* c, const int32_t* phoneme, const int32_t* puncts, const int32_t* duration, const int32_t* T,
                int B, int Tmax, const float* spk, int32_t* mel_len_out, int Lmax_cap) {
    const int H = c->H;
    c->have_features = false; c->have_mel = false;
    if (c->stream != c->front_stream) c->front_dirty_main = true;
    if (B <= 0 || Tmax <= 0) fail(ZVX_E_INVALID, "B and Tmax must be positive");
    if (c->vp_k != 3) fail(ZVX_E_UNSUPPORTED, "vp_kernel_size != 3 changes the sequence length (fs2.py:543 padding=1)");
    c->T_host.assign(T, T + B);
    for (int b = 0; b < B; b++) {
        if (T[b] <= 0 || T[b] > Tmax) fail(ZVX_E_INVALID, "T[%d]=%d out of range (1..%d)", b, T[b], Tmax);
        for (int t = 0; t < T[b]; t++) {
            const int ph = phoneme[b * Tmax + t], pu = puncts[b * Tmax + t];
            if (ph < 0 || ph >= c->n_phone_rows) fail(ZVX_E_INVALID, "phoneme id %d out of range at [%d][%d]", ph, b, t);   // nn.Embedding IndexError
            if (pu < 0 || pu >= c->n_punct_rows) fail(ZVX_E_INVALID, "punct id %d out of range at [%d][%d]", pu, b, t);
        }
    }
    c->B = B; c->Tmax = Tmax; c->have_features = false; c->have_mel = false;
    const size_t nid = (size_t)B * Tmax;
    // the call's small inputs (ids, lengths, forced durations, speaker embeddings) travel as ONE upload
    size_t in_stride = 0;
    char* in_base = c->carve("in.all", {"in.phoneme", "in.puncts", "in.duration", "in.T", "in.spk"},
                             std::max(std::max(nid, (size_t)B), (size_t)B * H) * 4, &in_stride);
    c->in_stage.resize(in_stride * 5 / sizeof(int));
    {
        int* hs = c->in_stage.data();
        const size_t st = in_stride / sizeof(int);
        memcpy(hs, phoneme, nid * 4); memcpy(hs + st, puncts, nid * 4);
        if (duration) memcpy(hs + 2 * st, duration, nid * 4);
        memcpy(hs + 3 * st, T, (size_t)B * 4);
        memcpy(hs + 4 * st, spk, (size_t)B * H * 4);
        c->upload(in_base, hs, in_stride * 5);
    }
    int* ph_d = (int*)in_base; int* pu_d = (int*)(in_base + in_stride);
    int* dur_in = duration ? (int*)(in_base + 2 * in_stride) : nullptr;
    int* T_d = (int*)(in_base + 3 * in_stride);
    float* spk_d = (float*)(in_base + 4 * in_stride);

    c->stage_begin(ZVX_T_ENCODER);
    c->tag = "encoder";
    // positional table: stored rows cover max_txt_len; longer inputs recompute it (fs2.py:383-388)
    const float* pe = c->pf("enc.pe");
    if (Tmax > c->t("enc.pe").dim(0)) {
        std::vector<float> tab((size_t)Tmax * H);
        for (int pos = 0; pos < Tmax; pos++)
            for (int j = 0; j < H; j++) {
                const double ang = pos / pow(10000.0, 2.0 * (j / 2) / H);
                tab[(size_t)pos * H + j] = (float)((j & 1) ? cos(ang) : sin(ang));
            }
        float* d = c->fbuf("enc.pe_ext", tab.size());
        HIPCHK(hipMemcpyAsync(d, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        pe = d;
    }
    float* x = c->fbuf("enc.x", nid * H);
    launch_embed(ph_d, pu_d, c->pf("enc.emb"), c->emb_dim, c->pf("enc.pemb"), c->punct_dim, pe, x, B, Tmax, T_d, c->stream);
    c->fft_xs_ready = nullptr;
    for (int i = 0; i < c->enc_layers; i++) {
        FftWeights w{"enc." + std::to_string(i), false, nullptr, 0, (i == c->enc_layers - 1) ? spk_d : nullptr};   // + style (fs2.py:740-741)
        fft_block(c, x, DT_F32, B, Tmax, T_d, c->enc_heads, w);
    }
    c->fft_xs_ready = nullptr;                                                                       // valid only between consecutive blocks of one stack
    c->stage_end(ZVX_T_ENCODER);

    c->stage_begin(ZVX_T_VARIANCE);
    c->tag = "variance";
    HIPCHK(hipMemcpyAsync(c->fbuf("enc.out", nid * H), x, nid * H * 4, hipMemcpyDeviceToDevice, c->stream));
    // the per-phoneme outputs that padding positions must read as zero: one zero-fill for all six
    size_t va_stride = 0;
    char* va_base = c->carve("va.zeroed", {"va.logd", "va.pitch", "va.energy", "va.pitch_idx", "va.energy_idx", "va.dur"}, nid * 4, &va_stride);
    HIPCHK(hipMemsetAsync(va_base, 0, va_stride * 6, c->stream));
    float* logd = (float*)va_base; float* pitch = (float*)(va_base + va_stride); float* energy = (float*)(va_base + 2 * va_stride);
    int* pidx = (int*)(va_base + 3 * va_stride); int* eidx = (int*)(va_base + 4 * va_stride);
    // The duration and the pitch predictor read the same x and are independent (fs2.py:663-668): the duration predictor runs on a
    // second stream beside the pitch predictor.  Their exact-f32 convolutions are chains of dependent matrix instructions on a
    // fraction of the CUs at any batch size (64 workgroups at B = 32 x 128), so this pays everywhere: the stage 0.43 -> 0.32-0.36 ms
    // for B = 1 ... 32 (tools/ab_va_overlap.py).  Same kernels, same arithmetic; x is only modified (pitch embedding) after both
    // have read it.
    if (B <= c->va_overlap_maxb) {
        if (!c->aux_stream) {
            HIPCHK(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&c->ev_aux[0], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_aux[1], hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(c->ev_aux[0], c->stream));
        HIPCHK(hipStreamWaitEvent(c->aux_stream, c->ev_aux[0], 0));
        {
            struct Swap { zvx_ctx* c; hipStream_t keep; ~Swap() { c->stream = keep; } } sw{c, c->stream};   // restored on every path out
            c->stream = c->aux_stream;
            variance_predictor(c, "dur", x, B, Tmax, T_d, logd, ".dur");
            HIPCHK(hipEventRecord(c->ev_aux[1], c->aux_stream));
        }
        variance_predictor(c, "pitch", x, B, Tmax, T_d, pitch);
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_aux[1], 0));
    } else {
        variance_predictor(c, "dur", x, B, Tmax, T_d, logd);                                           // fs2.py:663
        variance_predictor(c, "pitch", x, B, Tmax, T_d, pitch);                                        // fs2.py:665-668
    }
    launch_bucket_embed_add(pitch, c->pf("va.pitch_emb"), c->n_bins, x, H, H, pidx, B, Tmax, T_d, c->stream);
    variance_predictor(c, "energy", x, B, Tmax, T_d, energy);                                          // fs2.py:669-672
    launch_bucket_embed_add(energy, c->pf("va.energy_emb"), c->n_bins, x, H, H, eidx, B, Tmax, T_d, c->stream);
    c->stage_end(ZVX_T_VARIANCE);

    c->stage_begin(ZVX_T_LENREG);
    c->tag = "lenreg";
    int* dur = (int*)(va_base + 5 * va_stride); int* cum = c->ibuf("va.cum", nid); int* ml = c->ibuf("va.mel_len", B);
    launch_durations(dur_in, logd, dur, cum, ml, B, Tmax, T_d, c->stream);
    c->mel_len_host.resize(B);
    if (duration) {
        // forced durations: the mel lengths are their sums (same clamps as k_durations) -- nothing to wait for
        for (int b = 0; b < B; b++) {
            long long sum = 0;
            for (int t = 0; t < T[b]; t++) sum += std::min(std::max(duration[(size_t)b * Tmax + t], 0), 65536);
            c->mel_len_host[b] = (int)std::min(sum, 0x7fffffffLL);
        }
    } else {
        HIPCHK(hipMemcpyAsync(c->mel_len_host.data(), ml, B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));        // predicted durations: the one data-dependent host sync (model.py:325)
    }
    if (mel_len_out) memcpy(mel_len_out, c->mel_len_host.data(), B * sizeof(int));
    int Lmax = 0; for (int b = 0; b < B; b++) Lmax = std::max(Lmax, c->mel_len_host[b]);
    // checked BEFORE anything is sized by it: a garbage log-duration must not drive an allocation
    const int cap = Lmax_cap > 0 ? std::min(Lmax_cap, c->max_frames) : c->max_frames;
    if (Lmax > cap) fail(ZVX_E_BUFFER, "predicted mel length %d exceeds %s %d", Lmax, Lmax_cap > 0 && Lmax_cap <= c->max_frames ? "Lmax_cap" : "the context's max_frames", cap);
    c->Lmax = Lmax;
    if (Lmax > 0) {
        float* feats = c->fbuf("features", (size_t)B * Lmax * H);
        double rows_out = 0; for (int b = 0; b < B; b++) rows_out += c->mel_len_host[b];
        c->timed(0, (rows_out + (double)B * Tmax) * H * 4.0, [&] { launch_length_regulate(x, H, cum, T_d, ml, feats, B, Tmax, Lmax, H, c->stream); });   // fs2.py:447-455: pure copy
    }
    c->stage_end(ZVX_T_LENREG);
    c->have_features = true;
}

// ------------------------------------------------------------------------------------------------
// mel decoders
// ------------------------------------------------------------------------------------------------
void decoder_fs2(zvx_ctx* c, const float* feats, const float* spk_d, const int* L_d, int B, int Lmax, float* mel) {
    const int H = c->H;
    // 16-bit mode: IEEE half like the StyleTTS decoder (needs the fused attention: head depth 264), zvx_set_int("dec_f16", 0): bf16
    const int dt = (c->dt == DT_BF16 && c->dec_f16 && c->use_flash && H / c->dec_heads == 264 && c->has("dec.mel_w.h16")) ? DT_F16 : c->dt;
    const float* pe = c->pf("dec.pe");
    if (Lmax > c->t("dec.pe").dim(0)) {                   // fs2.py:287-294: table recomputed for longer inputs
        std::vector<float> tab((size_t)Lmax * H);
        for (int pos = 0; pos < Lmax; pos++)
            for (int j = 0; j < H; j++) {
                const double ang = pos / pow(10000.0, 2.0 * (j / 2) / H);
                tab[(size_t)pos * H + j] = (float)((j & 1) ? cos(ang) : sin(ang));
            }
        float* d = c->fbuf("dec.pe_ext", tab.size());
        HIPCHK(hipMemcpyAsync(d, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        pe = d;
    }
    // 16-bit mode: every utterance gets ffn_k / 2 padding rows so that the FFT blocks' static-weight GEMMs run batch-flattened (fft_block)
    const bool flat = dt != DT_F32 && c->dec_flat && B > 1;
    const int Ls = flat ? Lmax + std::max(c->ffn_k0, c->ffn_k1) / 2 : Lmax;
    void* x = c->buf("dec.x", (size_t)B * Ls * H * dtype_size(dt));
    launch_add_pe_cast(feats, pe, x, dt, H, B, Lmax, L_d, H, c->stream, Ls);
    c->sat_scan(x, dt, (long)Ls * H, H, B, Lmax, L_d, H);
    float* bg = nullptr; long bg_bs = 0;
    if (c->dec_scln) {      // all 2*layers SCLN affine vectors of the call in one GEMM: [b | g] = W s   (fs2.py:85)
        const int NA = 2 * c->dec_layers * 2 * H;
        bg = c->fbuf("dec.bg", (size_t)B * NA); bg_bs = NA;
        GemmArgs a = gemm_base(DT_F32);
        a.X = spk_d; a.ldx = H; a.W = c->t("dec.scln_all").dev; a.ldw = H; a.M = B; a.N = NA; a.K = H; a.in_len_static = B;
        a.out = bg; a.ldo = NA;
        c->gemm(a);
    }
    for (int i = 0; i < c->dec_layers; i++) {
        FftWeights w{"dec." + std::to_string(i), c->dec_scln != 0, bg ? bg + (long)i * 4 * H : nullptr, bg_bs, nullptr};
        fft_block(c, x, dt, B, Lmax, L_d, c->dec_heads, w, Ls, flat);
    }
    GemmArgs a = gemm_base(dt);                             // mel_linear   fs2.py:313
    a.X = x; a.x_bs = (long)Ls * H; a.ldx = H; a.W = c->t(dt == DT_F16 ? "dec.mel_w.h16" : "dec.mel_w").dev; a.ldw = H;
    a.M = Lmax; a.N = c->n_mels; a.K = H; a.nbatch = B; a.in_len = L_d; a.out_len = L_d;
    a.bias = c->pf("dec.mel_b"); a.bias_mode = 1;
    a.out = mel; a.out_dtype = DT_F32; a.o_bs = (long)Lmax * c->n_mels; a.ldo = c->n_mels;
    c->gemm(a);
}

// Lmax: rows per utterance of every buffer of the decoder = longest utterance + one padding row (never a valid position), so that
// its k = 3 / 1x1 convolutions may run BATCH-FLATTENED (GemmArgs::bflat): row tiles cross utterance boundaries, no utterance ends in
// a half-empty tile.  Lrows: the longest utterance.  dt: the decoder's activation dtype (bf16 / f16 / f32)
struct StyCtx { zvx_ctx* c; int B, Lmax; const int* L_d; float* mean; float* rstd; int dt; int Lrows; int flat; const void* mel; };

// x2 / ldx2 / Cin2: the block's 1x1 shortcut as a second source of this convolution (the caller has checked can_fuse_shortcut)
void sty_conv(const StyCtx& s, const std::string& wname, const void* x, int ldx, int Cin, void* out, int ldo, int out_dt,
              int Cout, const void* res, int ldr, float out_scale, const void* x2 = nullptr, int ldx2 = 0, int Cin2 = 0) {
    zvx_ctx* c = s.c;
    const Tensor& w = c->t(s.dt == DT_F16 ? wname + ".h16" : wname);
    GemmArgs a = gemm_base(s.dt);
    if (x2) { a.X2 = x2; a.x2_bs = (long)s.Lmax * ldx2; a.ldx2 = ldx2; a.K2 = Cin2; a.Wp = c->pair_packed.at(w.dev); }
    a.X = x; a.x_bs = (long)s.Lmax * ldx; a.ldx = ldx; a.W = w.dev; a.ldw = Cin; a.w_ts = (long)Cout * Cin;
    a.M = s.Lrows; a.N = Cout; a.K = Cin; a.nbatch = s.B; a.in_len = s.L_d; a.out_len = s.L_d;
    const bool to_mel = out == s.mel;                              // the mel projection writes the caller-visible [B][Lrows][n_mels] layout: per utterance
    a.bflat = (s.flat && !to_mel) ? s.Lmax : 0;
    set_taps_1d(a, w.dim(0), 1);
    if (c->has(wname + "_b")) { a.bias = c->pf(wname + "_b"); a.bias_mode = 1; }
    if (res) { a.res = res; a.r_bs = (long)s.Lmax * ldr; a.ldr = ldr; a.res_mode = 1; a.res_dtype = s.dt; }
    a.out_scale = out_scale;
    a.out = out; a.o_bs = (long)(to_mel ? s.Lrows : s.Lmax) * ldo; a.ldo = ldo; a.out_dtype = out_dt;
    c->gemm(a);
}
// y = lrelu_0.2(affine(IN(x)))  (InstanceNorm over each utterance's true length only)
void sty_norm(const StyCtx& s, const void* x, int ldx, int C, void* y, int ldy, const float* gamma, const float* beta,
              long g_bs, int one_plus, int act) {
    const std::string keep = s.c->tag;
    s.c->tag = "decoder.norm";
    s.c->timed(0, (double)s.B * s.Lmax * C * s.c->es() * 3.0, [&] {          // statistics pass (read) + normalise pass (read + write)
        // One launch (statistics, then the same workgroup walks its rows again: bit-identical) where it measures faster
        // (tools/ab_norm_fuse.py): short utterances at any batch size, and any length once B x C / 64 workgroups fill a fair part
        // of the chip.  A few workgroups walking many rows twice (B <= 4 at 896 frames) lose to the wide second launch.
        const bool fuse_pays = s.Lmax <= 512 || (long)s.B * ((C + 63) / 64) >= 64;
        if (s.dt != DT_F32 && s.B <= s.c->norm_fuse_maxb && fuse_pays) {
            launch_instnorm_fused(x, s.dt, ldx, y, s.dt, ldy, s.B, s.Lmax, s.L_d, C, 1e-5f, s.mean, s.rstd, gamma, beta, g_bs, one_plus, act, 0.2f, s.c->stream);
            return;
        }
        launch_instnorm_stats(x, s.dt, ldx, s.B, s.Lmax, s.L_d, C, 1e-5f, s.mean, s.rstd, s.c->stream);
        launch_norm_affine_act(x, s.dt, ldx, y, s.dt, ldy, s.B, s.Lmax, s.L_d, C, s.mean, s.rstd, gamma, beta, g_bs, one_plus,
                               act, 0.2f, s.c->stream);
    });
    s.c->sat_scan(y, s.dt, (long)s.Lmax * ldy, ldy, s.B, s.Lmax, s.L_d, C);
    s.c->tag = keep;
}

void decoder_styletts(zvx_ctx* c, const float* feats, const float* spk_d, const int* L_d, int B, int Lrows, float* mel) {
    const int H = c->H, H2 = 2 * H, R = c->res_dim, CW = H2 + R;
    const int dt = (c->dt == DT_BF16 && c->dec_f16 && c->has("sty.out.h16")) ? DT_F16 : c->dt;      // 16-bit mode: IEEE half unless switched off
    const int Lmax = Lrows + 1;                           // row stride of every buffer below (see StyCtx)
    const size_t es = dtype_size(dt), rows = (size_t)B * Lmax;
    const float inv_sqrt2 = (float)(1.0 / sqrt(2.0));
    StyCtx s{c, B, Lmax, L_d, c->fbuf("sty.mean", (size_t)B * CW), c->fbuf("sty.rstd", (size_t)B * CW), dt, Lrows, c->dec_flat, mel};
    void* e = c->buf("sty.e", rows * H * es);
    void* t0 = c->buf("sty.t0", rows * CW * es);
    void* t1 = c->buf("sty.t1", rows * H2 * es);
    void* r = c->buf("sty.r", rows * H2 * es);
    void* catA = c->buf("sty.catA", rows * CW * es);
    void* catB = c->buf("sty.catB", rows * CW * es);
    launch_add_pe_cast(feats, nullptr, e, dt, H, B, Lrows, L_d, H, c->stream, Lmax);
    c->sat_scan(e, dt, (long)Lmax * H, H, B, Lmax, L_d, H);

    // AdaIN affine vectors for all 10 norms: h = fc(s)          styletts.py:89-91
    const Tensor& aw = c->t("sty.adain_w");
    const int NA = aw.dim(1);
    float* hall = c->fbuf("sty.adain_h", (size_t)B * NA);
    {
        GemmArgs a = gemm_base(DT_F32);
        a.X = spk_d; a.ldx = H; a.W = aw.dev; a.ldw = H; a.M = B; a.N = NA; a.K = H; a.in_len_static = B;
        a.bias = c->pf("sty.adain_b"); a.bias_mode = 1; a.out = hall; a.ldo = NA;
        c->gemm(a);
    }
    // encode.0: ResBlk1d(H -> 2H, normalize)                       styletts.py:44-69
    sty_norm(s, e, H, H, t0, H, c->pf("sty.enc0.norm1_g"), c->pf("sty.enc0.norm1_b"), 0, 0, ACT_LRELU);
    sty_conv(s, "sty.enc0.c1", t0, H, H, t1, H, dt, H, nullptr, 0, 1.f);
    sty_norm(s, t1, H, H, t0, H, c->pf("sty.enc0.norm2_g"), c->pf("sty.enc0.norm2_b"), 0, 0, ACT_LRELU);
    auto can_fuse_shortcut = [&](const std::string& blk) {
        if (c->dec_sc_fuse >= 2) {                            // development: bits 1.. select the blocks (enc0, dec0, dec1, dec2) that may fuse
            const int bi = blk == "sty.enc0" ? 0 : (blk.size() == 8 ? 1 + (blk[7] - '0') : 9);
            if (!((c->dec_sc_fuse >> (1 + bi)) & 1)) return false;
        }
        return c->dec_sc_fuse && dt != DT_F32 && c->has(blk + ".c2") && c->pair_packed.count(c->t(dt == DT_F16 ? blk + ".c2.h16" : blk + ".c2").dev) != 0;
    };
    if (can_fuse_shortcut("sty.enc0")) {
        sty_conv(s, "sty.enc0.c2", t0, H, H, catA, CW, dt, H2, nullptr, 0, inv_sqrt2, e, H, H);       // (conv2(t) + conv1x1(e)) / sqrt(2) in one launch
    } else {
        sty_conv(s, "sty.enc0.c2", t0, H, H, r, H2, dt, H2, nullptr, 0, 1.f);
        sty_conv(s, "sty.enc0.sc", e, H, H, catA, CW, dt, H2, r, H2, inv_sqrt2);
    }
    // encode.1: ResBlk1d(2H -> 2H), identity shortcut (in place on catA[:, 0:2H])
    sty_norm(s, catA, CW, H2, t0, H2, c->pf("sty.enc1.norm1_g"), c->pf("sty.enc1.norm1_b"), 0, 0, ACT_LRELU);
    sty_conv(s, "sty.enc1.c1", t0, H2, H2, t1, H2, dt, H2, nullptr, 0, 1.f);
    sty_norm(s, t1, H2, H2, t0, H2, c->pf("sty.enc1.norm2_g"), c->pf("sty.enc1.norm2_b"), 0, 0, ACT_LRELU);
    sty_conv(s, "sty.enc1.c2", t0, H2, H2, catA, CW, dt, H2, catA, CW, inv_sqrt2);
    // asr_res = IN_affine(conv1x1(e)) -> columns [2H, 2H+R) of both concat buffers (concat-free torch.cat, styletts.py:195)
    sty_conv(s, "sty.asr", e, H, H, t1, R, dt, R, nullptr, 0, 1.f);
    launch_instnorm_stats(t1, dt, R, B, Lmax, L_d, R, 1e-5f, s.mean, s.rstd, c->stream);
    for (void* cat : {catA, catB})
        launch_norm_affine_act(t1, dt, R, (char*)cat + (size_t)H2 * es, dt, CW, B, Lmax, L_d, R, s.mean, s.rstd, c->pf("sty.asr_g"),
                               c->pf("sty.asr_beta"), 0, 0, ACT_NONE, 0.f, c->stream);
    c->sat_scan((char*)catA + (size_t)H2 * es, dt, (long)Lmax * CW, CW, B, Lmax, L_d, R);
    // decode.0..4: AdainResBlk1d                                    styletts.py:119-139
    struct Blk { int cin, cout; bool cat_out; };
    const Blk blks[5] = {{CW, H2, true}, {CW, H2, true}, {CW, H, false}, {H, H, false}, {H, H, false}};
    void* cur = catA; int cur_ld = CW;
    void* nxtcat = catB;
    void* xa = c->buf("sty.xa", rows * H * es);
    void* xb = c->buf("sty.xb", rows * H * es);
    long hoff = 0;
    for (int i = 0; i < 5; i++) {
        const Blk& bk = blks[i];
        const std::string p = "sty.dec" + std::to_string(i);
        const float* g1 = hall + hoff;            const float* b1 = g1 + bk.cin;  hoff += 2 * bk.cin;
        const float* g2 = hall + hoff;            const float* b2 = g2 + bk.cout; hoff += 2 * bk.cout;
        sty_norm(s, cur, cur_ld, bk.cin, t0, bk.cin, g1, b1, NA, 1, ACT_LRELU);
        sty_conv(s, p + ".c1", t0, bk.cin, bk.cin, t1, bk.cout, dt, bk.cout, nullptr, 0, 1.f);
        sty_norm(s, t1, bk.cout, bk.cout, t0, bk.cout, g2, b2, NA, 1, ACT_LRELU);
        void* out; int out_ld;
        if (bk.cat_out) { out = nxtcat; out_ld = CW; } else { out = (cur == xa) ? xb : xa; out_ld = H; }
        if (c->has(p + ".sc") && can_fuse_shortcut(p)) {
            sty_conv(s, p + ".c2", t0, bk.cout, bk.cout, out, out_ld, dt, bk.cout, nullptr, 0, inv_sqrt2, cur, cur_ld, bk.cin);
        } else if (c->has(p + ".sc")) {
            sty_conv(s, p + ".c2", t0, bk.cout, bk.cout, r, bk.cout, dt, bk.cout, nullptr, 0, 1.f);
            sty_conv(s, p + ".sc", cur, cur_ld, bk.cin, out, out_ld, dt, bk.cout, r, bk.cout, inv_sqrt2);
        } else {
            sty_conv(s, p + ".c2", t0, bk.cout, bk.cout, out, out_ld, dt, bk.cout, cur, cur_ld, inv_sqrt2);
        }
        if (bk.cat_out) { nxtcat = cur; }
        cur = out; cur_ld = out_ld;
    }
    sty_conv(s, "sty.out", cur, cur_ld, H, mel, c->n_mels, DT_F32, c->n_mels, nullptr, 0, 1.f);
}

void run_decode(zvx_ctx* c, const float* feats, const float* spk_d, const int* L_d, int B, int Lmax) {
    if (c->stream != c->front_stream) c->front_dirty_main = true;
    c->stage_begin(ZVX_T_DECODER);
    c->tag = "decoder";
    float* mel = c->fbuf("mel", (size_t)B * std::max(Lmax, 1) * c->n_mels);
    if (c->poison_pads) {
        // every work buffer of the decoders starts as NaN patterns (0xFF bytes: NaN as f32, half and bf16): batch-flattened launches write
        // junk into padding rows and read rows past an utterance's length by design -- every consumer must MASK them (select), never
        // multiply them by zero.  With this switch a violation shows up as NaN in the mel (the test compares with the clean run bit for bit).
        for (auto& kv : c->bufs) {
            const std::string& nm = kv.first;
            const bool mine = nm.rfind("sty.", 0) == 0 || nm.rfind("fft.", 0) == 0 || nm == "dec.x";
            if (mine && kv.second.p && kv.second.base == kv.second.p && nm != "sty.adain_h" && nm != "dec.bg") HIPCHK(hipMemsetAsync(kv.second.p, 0xFF, kv.second.cap, c->stream));
        }
    }
    if (Lmax > 0) {
        if (c->dec_kind == 0) decoder_fs2(c, feats, spk_d, L_d, B, Lmax, mel);
        else decoder_styletts(c, feats, spk_d, L_d, B, Lmax, mel);
        launch_zero_tail_rows(mel, c->n_mels, B, Lmax, L_d, c->n_mels, c->stream);     // rows >= mel_len[b]: zeros, whatever ran before
    }
    c->stage_end(ZVX_T_DECODER);
    c->have_mel = true;
}

// ------------------------------------------------------------------------------------------------
// HiFi-GAN generator (hifigan.py:114-130).  Activations are stored in the *activated* domain:
// every tensor a later conv consumes through leaky_relu(., 0.1) is written as leaky_relu(x); the raw
// residual is recovered exactly-enough by the inverse map in the consumer's epilogue (res_mode 2).
// ------------------------------------------------------------------------------------------------
void run_vocoder(zvx_ctx* c, const float* mel, int ldm, int Lmel_max, const int* mel_len_host, const int* P_host, int B,
                 void* wav_dev, long wav_stride, int pcm16) {
    // 16-bit mode: IEEE half (round 5) -- f16 copies of the weights, f16 activations / running sum, the f16 MFMA at the bf16 rate and 8x
    // less rounding error per tensor; every store saturates (MODE.FP16_OVFL inside the kernels).  zvx_set_int("voc_f16", 0): bf16 (A/B)
    if (c->voc_h16_ok < 0) {
        // decided once: the half copies exist only for 3-D weights with K % 8 == 0 (upload_weights); a generator with a narrower tensor
        // (e.g. a 4-channel last stage) runs entirely on the bf16 kernels, which need no packed copies
        c->voc_h16_ok = 1;
        for (auto& kv : c->tensors)
            if (kv.first.rfind("voc.", 0) == 0 && kv.second.kind == 'w' && kv.second.dims.size() == 3 && !c->has(kv.first + ".h16")) c->voc_h16_ok = 0;
    }
    const bool half_ok = c->dt == DT_BF16 && c->voc_f16 && c->voc_h16_ok == 1 && c->has("voc.pre_w.h16");
    const int nm = c->n_mels;
    // The 16-bit arithmetic is chosen PER STAGE (round 6; zvx_set_int("voc_f16_stages", mask)): domain 0 = the padded mel, conv_pre and its
    // output; domain i = everything of upsampling stage i (the ConvTranspose's OUTPUT, the ResBlocks' tensors, weights and running sum, the
    // stage's result).  A ConvTranspose reads domain i - 1 with that domain's weights and writes domain i (the run-time epilogue converts:
    // one rounding either way).  Bit k of the mask: domain k in IEEE half, else bf16.
    // Default (mask -1): half everywhere except a ResBlock1 stage of 128 channels -- the pair kernel's stage (HiFi-GAN V1's second): 43 % of
    // that generator's matrix work, where the f16 multiplier array's extra power is 0.3 of the 0.8-1.0 ms the all-half generator costs at
    // the board's power limit, for 1/5 of the error budget (tools/ab_voc_stages.py, tools/vocoder_error_budget.py)
    int mask = c->voc_f16_stages;
    if (mask < 0) {
        mask = 0x7fffffff;
        int cw = c->voc_c0;
        for (int i = 0; i < (int)c->voc_rates.size(); i++) { cw /= 2; if (cw == 128 && c->voc_resblock == 1) mask &= ~(1 << (i + 1)); }
    }
    auto dom = [&](int k) -> int { return (half_ok && ((mask >> k) & 1)) ? (int)DT_F16 : c->dt; };
    auto vtd = [&](const std::string& n, int d) -> const Tensor& { return c->t(d == DT_F16 ? n + ".h16" : n); };      // a contraction weight in a domain's dtype
    const size_t es = dtype_size(c->dt);
    int dt = dom(0);                                                  // the domain of the launches being issued
    int Pmax = 0; for (int b = 0; b < B; b++) Pmax = std::max(Pmax, P_host[b]);
    if (Pmax <= 0) return;
    const int ns = (int)c->voc_rates.size(), nk = (int)c->voc_rb_k.size();
    // per-stage valid row counts
    std::vector<int> lens((size_t)(ns + 2) * B);
    for (int b = 0; b < B; b++) { lens[b] = mel_len_host[b]; lens[B + b] = P_host[b]; }
    int mul = 1;
    for (int i = 0; i < ns; i++) { mul *= c->voc_rates[i]; for (int b = 0; b < B; b++) lens[(size_t)(i + 2) * B + b] = P_host[b] * mul; }
    int* lens_d = c->upload_ints("voc.lens", lens.data(), lens.size());
    const int* mel_len_d = lens_d; const int* P_d = lens_d + B;

    size_t maxel = (size_t)Pmax * c->voc_c0; mul = 1;
    for (int i = 0; i < ns; i++) { mul *= c->voc_rates[i]; maxel = std::max(maxel, (size_t)Pmax * mul * (size_t)(c->voc_c0 >> (i + 1))); }
    maxel *= B;
    void* vin = c->buf("voc.in", (size_t)B * Pmax * nm * es);
    void* A = c->buf("voc.A", maxel * es);        // stage input (activated)
    void* X0 = c->buf("voc.X0", maxel * es);      // upsampled stage tensor (activated)
    void* T1 = c->buf("voc.T1", maxel * es);
    void* PP[2] = {c->buf("voc.PP0", maxel * es), c->buf("voc.PP1", maxel * es)};
    void* XS = c->buf("voc.XS", maxel * es);     // running sum over the resblocks of a stage, in the activation dtype
    // temporaries of the ResBlocks that run beside the first one (single requests, below).  Taken HERE: a new buffer is zero-filled
    // on the main stream, which the side streams only follow from the stage-input event on
    // tools/ab_voc_overlap.py: -5 ... -24 % for every batch below the headline's size (most where B leaves a ragged last round of
    // workgroups: B = 6 / 12), +-1 % at 32 x 896 frames, which therefore keeps the serial schedule
    const bool side_ok = nk >= 2 && nk <= 3 && B <= c->voc_overlap_maxb && (long)B * Pmax < c->voc_overlap_frames && !(c->voc_chunk > 0 && c->voc_chunk < B);
    void* sideT1[2] = {nullptr, nullptr}; void* sidePP[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    if (side_ok)
        for (int j = 1; j < nk; j++) {
            sideT1[j - 1] = c->buf("voc.T1." + std::to_string(j), maxel * es);
            sidePP[j - 1][0] = c->buf("voc.PP0." + std::to_string(j), maxel * es); sidePP[j - 1][1] = c->buf("voc.PP1." + std::to_string(j), maxel * es);
        }

    c->tag = "voc.pre";
    launch_mel_pad(mel, DT_F32, ldm, Lmel_max, mel_len_d, vin, dt, nm, Pmax, P_d, B, nm, c->stream);     // model.py:331-335
    c->sat_scan(vin, dt, (long)Pmax * nm, nm, B, Pmax, P_d, nm);
    if (c->front_stream && c->stream != c->front_stream) {   // the mel buffer is free again: a queued call's decoder (front stream) may overwrite it
        HIPCHK(hipEventRecord(c->ev_mel_free, c->stream));
        c->mel_free_pending = true;
    }
    {   // conv_pre, stored as leaky_relu(x, 0.1) (its only consumer, hifigan.py:115-117)
        GemmArgs a = gemm_base(dt);
        a.X = vin; a.x_bs = (long)Pmax * nm; a.ldx = nm; a.W = vtd("voc.pre_w", dt).dev; a.ldw = nm; a.w_ts = (long)c->voc_c0 * nm;
        a.M = Pmax; a.N = c->voc_c0; a.K = nm; a.nbatch = B; a.in_len = P_d; a.out_len = P_d;
        set_taps_1d(a, c->t("voc.pre_w").dim(0), 1);
        a.bias = c->pf("voc.pre_b"); a.bias_mode = 1; a.act = ACT_LRELU; a.slope = 0.1f;
        a.out = A; a.o_bs = (long)Pmax * c->voc_c0; a.ldo = c->voc_c0;
        c->gemm(a);
    }
    int Cin = c->voc_c0; mul = 1;
    for (int i = 0; i < ns; i++) {
        const int u = c->voc_rates[i], ku = c->voc_ksizes[i], Cout = Cin / 2;
        const int rows_in = Pmax * mul, rows = rows_in * u;
        const int* len_in = lens_d + (size_t)(i + 1) * B; const int* len = lens_d + (size_t)(i + 2) * B;
        c->tag = "voc.up" + std::to_string(i + 1);
        const int dt_in = dom(i);
        dt = dom(i + 1);
        const bool h16 = dt == DT_F16;
        auto vt = [&](const std::string& n) -> const Tensor& { return vtd(n, dt); };                // a contraction weight in this stage's dtype
        {   // ConvTranspose1d as a polyphase GEMM: out[t][ph*Cout+co]                  hifigan.py:118
            const std::string up = "voc.up" + std::to_string(i);
            GemmArgs a = gemm_base(dt_in);
            a.out_dtype = dt;
            a.X = A; a.x_bs = (long)rows_in * Cin; a.ldx = Cin; a.ldw = Cin;
            a.M = rows_in; a.K = Cin; a.nbatch = B; a.in_len = len_in; a.out_len = len_in;
            a.bias_mode = 1; a.act = ACT_LRELU; a.slope = 0.1f;
            a.o_bs = (long)rows_in * u * Cout; a.ldo = u * Cout;
            if (dt_in != DT_F32 && c->has(up + "_wlo")) {
                // k = 2u: two 2-tap GEMMs over half the phases each (rows t-1, t / rows t, t+1) instead of one 3-tap GEMM
                const int hc = (u / 2) * Cout;
                a.N = hc; a.ntaps = 2; a.w_ts = (long)hc * Cin; a.flops = 2.0 * B * (double)rows_in * Cin * Cout * ku / 2;
                a.W = vtd(up + "_wlo", dt_in).dev; a.dv[0] = -1; a.dv[1] = 0;
                a.bias = c->pf(up + "_b"); a.out = X0;
                c->gemm(a);
                a.Wp = nullptr;
                a.W = vtd(up + "_whi", dt_in).dev; a.dv[0] = 0; a.dv[1] = 1;
                a.bias = c->pf(up + "_b") + hc; a.out = (char*)X0 + (size_t)hc * es;
                c->gemm(a);
            } else {
                a.W = vtd(up + "_w", dt_in).dev; a.w_ts = (long)u * Cout * Cin; a.N = u * Cout;
                a.ntaps = 3; a.dv[0] = -1; a.dv[1] = 0; a.dv[2] = 1;
                a.bias = c->pf(up + "_b"); a.out = X0;
                a.flops = 2.0 * B * (double)rows_in * Cin * Cout * ku;
                c->gemm(a);
            }
        }
        const float next_slope = (i == ns - 1) ? 0.01f : 0.1f;          // hifigan.py:126 uses the default slope 0.01
        c->tag = "voc.res" + std::to_string(i + 1);
        // The stage's ResBlocks can run on sub-batches of utterances (zvx_set_int("voc_chunk", n)) so that the tensors one
        // sub-batch touches stay in the 256 MB Infinity Cache between launches.  Measured on B = 32 x 896 frames: no gain at
        // any size (stage 2: 7.70 ms whole batch, 7.70 / 7.74 / 8.40 ms at 16 / 8 / 4 utterances) -- these stages are bound by
        // MFMA issue, not by HBM -- so the default is the whole batch.  Results are bit-identical for any sub-batch size.
        const size_t utt_bytes = (size_t)rows * Cout * es;
        const int CH = (c->voc_chunk > 0 && c->voc_chunk < B) ? c->voc_chunk : B;
        // narrow stages (C = 16 / 8: HiFi-GAN V2's last two): all ResBlocks of the stage, their mean and the next stage's activation in ONE
        // launch -- the stage tensor crosses HBM once in, once out (narrowstage.hip)
        if (c->use_stagefuse && !c->sat_check && c->voc_resblock == 1 && dt != DT_F32 && (Cout == 16 || Cout == 8) && nk >= 1 && nk <= 3 && CH == B) {
            bool ok = true;
            for (int j = 0; j < nk; j++) ok = ok && c->voc_rb_d[j].size() == 3;
            StageArgs sa;
            memset(&sa, 0, sizeof sa);
            if (ok) {
                const std::string key = "voc.ns." + std::to_string(i) + (h16 ? ".h16" : "");
                auto it = c->ns_weights.find(key);
                if (it == c->ns_weights.end()) {
                    zvx_ctx::NsWeights nw;
                    int nfrag = 0;
                    for (int j = 0; j < nk; j++) for (int q = 0; q < 6; q++) { nw.woff[6 * j + q] = nfrag; nfrag += narrowstage_steps(Cout, c->voc_rb_k[j]); }
                    nw.W = c->buf(key + ".w", (size_t)nfrag * 1024);
                    nw.bias = c->fbuf(key + ".b", (size_t)6 * nk * Cout);
                    for (int j = 0; j < nk; j++)
                        for (int t = 0; t < 3; t++)
                            for (int q = 0; q < 2; q++) {
                                const std::string nm = "voc.rb" + std::to_string(i * nk + j) + (q ? ".c2_" : ".c1_") + std::to_string(t);
                                const Tensor& w = vt(nm + "_w");
                                if (w.dims.size() != 3 || w.dim(0) != c->voc_rb_k[j] || w.dim(1) != Cout || w.dim(2) != Cout) fail(ZVX_E_MANIFEST, "weights: '%s' has an unexpected shape", nm.c_str());
                                const int ci = 6 * j + 2 * t + q;
                                launch_pack_narrow(w.dev, c->voc_rb_k[j], Cout, (char*)nw.W + (size_t)nw.woff[ci] * 1024, c->stream);
                                HIPCHK(hipMemcpyAsync(nw.bias + (size_t)ci * Cout, c->pf(nm + "_b"), (size_t)Cout * 4, hipMemcpyDeviceToDevice, c->stream));
                            }
                    it = c->ns_weights.emplace(key, nw).first;
                }
                sa.X = X0; sa.x_bs = (long)rows * Cout; sa.ldx = Cout; sa.W = it->second.W; sa.bias = it->second.bias;
                memcpy(sa.woff, it->second.woff, sizeof sa.woff);
                sa.C = Cout; sa.nk = nk;
                for (int j = 0; j < nk; j++) { sa.ks[j] = c->voc_rb_k[j]; for (int t = 0; t < 3; t++) sa.dil[j][t] = c->voc_rb_d[j][t]; }
                sa.out = A; sa.o_bs = (long)rows * Cout; sa.ldo = Cout;
                sa.slope1 = 0.1f; sa.res_inv_slope = 10.0f; sa.slope = next_slope;
                sa.len = len; sa.M = rows; sa.nbatch = B; sa.f16 = h16;
                ok = launch_narrowstage(sa, c->stream, true);
            }
            if (ok) {
                GemmEvent ev{};
                const int vid = Cout == 16 ? 24 : 25;
                const bool prof = c->profile >= 2 && (c->profile_only < 0 || c->profile_only == vid);
                if (prof) { ev.a = c->new_event(); ev.b = c->new_event(); narrowstage_profile_events(ev.a, ev.b); }
                const bool launched = launch_narrowstage(sa, c->stream, false);
                if (prof) narrowstage_profile_events(nullptr, nullptr);
                if (!launched) fail(ZVX_E_INVALID, "launch_narrowstage rejected a stage its dry run accepted (C=%d)", Cout);
                if (prof) {
                    double ksum = 0; for (int j = 0; j < nk; j++) ksum += c->voc_rb_k[j];
                    const double rws = (double)rows * B;
                    ev.variant = vid; ev.flops = 2.0 * rws * Cout * Cout * 6.0 * ksum; ev.rows = (long)rws; ev.N = Cout; ev.K = Cout; ev.taps = (int)ksum; ev.fused = 100 + nk;
                    ev.bytes = rws * Cout * 2.0 * 2.0; ev.tag = c->tag;
                    c->pending.push_back(ev);
                }
                Cin = Cout; mul *= u;
                continue;
            }
        }
        for (int b0 = 0; b0 < B; b0 += CH) {
            const int Bs = std::min(CH, B - b0);
            const size_t boff = (size_t)b0 * utt_bytes;
            const void* X0s = (const char*)X0 + boff;
            void* T1s = (char*)T1 + boff; void* XSs = (char*)XS + boff; void* As = (char*)A + boff;
            void* PPs[2] = {(char*)PP[0] + boff, (char*)PP[1] + boff};
            const int* lens = len + b0;
            // Single requests: the ResBlocks of a stage only meet in the running sum, which their LAST pair updates.  Everything
            // before that (two of three pairs) runs on a stream of its own for the 2nd and 3rd ResBlock, with its own temporaries;
            // the last pairs stay on the main stream in the order 1, 2, 3, so xs accumulates exactly as before (bit-identical).
            // The critical path of a per-pair stage drops from 18 to 10 launches.
            const bool overlap = side_ok;
            hipStream_t const main_stream = c->stream;
            struct Restore { zvx_ctx* c; hipStream_t keep; ~Restore() { c->stream = keep; } } restore{c, main_stream};
            if (overlap) {
                for (int q = 0; q < 2; q++) if (!c->voc_aux[q]) HIPCHK(hipStreamCreateWithFlags(&c->voc_aux[q], hipStreamNonBlocking));
                for (int q = 0; q < 3; q++) if (!c->voc_ev[q]) HIPCHK(hipEventCreateWithFlags(&c->voc_ev[q], hipEventDisableTiming));
                HIPCHK(hipEventRecord(c->voc_ev[0], main_stream));      // the stage input (and everything the previous stage read) is settled
            }
            for (int j = 0; j < nk; j++) {
                hipStream_t const my_aux = (overlap && j >= 1) ? c->voc_aux[j - 1] : nullptr;
                bool on_aux = false;
                if (my_aux) {
                    HIPCHK(hipStreamWaitEvent(my_aux, c->voc_ev[0], 0));
                    T1s = sideT1[j - 1]; PPs[0] = sidePP[j - 1][0]; PPs[1] = sidePP[j - 1][1];
                } else {
                    T1s = (char*)T1 + boff; PPs[0] = (char*)PP[0] + boff; PPs[1] = (char*)PP[1] + boff;
                }
                auto to_aux = [&] { if (my_aux && !on_aux) { c->stream = my_aux; on_aux = true; } };
                auto to_main = [&] {
                    if (on_aux) {
                        HIPCHK(hipEventRecord(c->voc_ev[j], my_aux));
                        c->stream = main_stream; on_aux = false;
                        HIPCHK(hipStreamWaitEvent(main_stream, c->voc_ev[j], 0));
                    }
                };
                const int k = c->voc_rb_k[j];
                const std::vector<int>& dil = c->voc_rb_d[j];
                const int nd = (int)dil.size();
                const std::string rb = "voc.rb" + std::to_string(i * nk + j);
                const void* cur = X0s;
                int pp = 0;
                int t_first = 0;
                if (c->voc_resblock == 1 && dt != DT_F32 && c->use_resstream && !c->sat_check && nd >= 1 && nd <= 3) {
                    // whole ResBlock (or its first two pairs + the last one) as streaming launches: the stage tensor crosses HBM once
                    bool packed_ok = true;
                    for (int t = 0; t < nd; t++)
                        packed_ok = packed_ok && c->packed.count(vt(rb + ".c1_" + std::to_string(t) + "_w").dev) && c->packed.count(vt(rb + ".c2_" + std::to_string(t) + "_w").dev);
                    auto chain = [&](int t0, int np, const void* in, bool closes) {
                        StreamArgs sa;
                        memset(&sa, 0, sizeof sa);
                        sa.X = in; sa.x_bs = (long)rows * Cout; sa.ldx = Cout; sa.C = Cout; sa.ntaps = k; sa.npair = np;
                        for (int q = 0; q < np; q++) {
                            const std::string ts = std::to_string(t0 + q);
                            sa.W1[q] = c->packed[vt(rb + ".c1_" + ts + "_w").dev]; sa.W2[q] = c->packed[vt(rb + ".c2_" + ts + "_w").dev];
                            sa.b1[q] = c->pf(rb + ".c1_" + ts + "_b"); sa.b2[q] = c->pf(rb + ".c2_" + ts + "_b");
                            sa.dil[q] = dil[t0 + q];
                        }
                        sa.opt = c->rs_opt; sa.seg_min = c->rs_seg_min; sa.f16 = h16;
                        if (c->rs_prof) sa.prof = (long long*)c->buf("rs.prof." + rb + "." + std::to_string(t0), 16 * 16 * 8);   // RS_PROFILE builds only
                        sa.slope1 = 0.1f; sa.res_inv_slope = 10.0f; sa.out_scale = 1.f; sa.slope = 0.1f;
                        sa.len = lens; sa.M = rows; sa.nbatch = Bs; sa.o_bs = (long)rows * Cout; sa.ldo = Cout; sa.a_bs = (long)rows * Cout; sa.lda = Cout;
                        if (!closes) { sa.out = PPs[pp]; }
                        else if (nk == 1) { sa.out = As; sa.slope = next_slope; }
                        else {
                            sa.accum = XSs; sa.accum_mode = j == 0 ? 2 : (j < nk - 1 ? 3 : 1);
                            if (j == nk - 1) { sa.out = As; sa.out_scale = 1.0f / nk; sa.slope = next_slope; }
                        }
                        return sa;
                    };
                    if (packed_ok) {
                        StreamArgs whole = chain(0, nd, X0s, true);
                        to_main();
                        if (c->run_stream(whole)) t_first = nd;
                        else if (nd == 3) {
                            StreamArgs head = chain(0, 2, X0s, false);
                            StreamArgs probe = chain(2, 1, PPs[pp], true);
                            if (launch_resstream(head, c->stream, true) >= 0 && launch_resstream(probe, c->stream, true) >= 0) {
                                to_aux();
                                c->run_stream(head);
                                cur = PPs[pp]; pp ^= 1;
                                StreamArgs tail = chain(2, 1, cur, true);
                                to_main();
                                c->run_stream(tail);
                                t_first = nd;
                            }
                        }
                    }
                }
                if (c->voc_resblock == 2 && nd == 2 && dt != DT_F32 && c->use_rb2fuse && !c->sat_check && (Cout == 32 || Cout == 64)) {
                    // a whole ResBlock2 (hifigan.py:77-82: x1 = x + c_0(lrelu(x)); x2 = x1 + c_1(lrelu(x1))) as ONE launch: lrelu(x1) stays in LDS
                    // (rb2fuse_kernel, round 6) -- two trips of the stage tensor per block instead of six
                    const Tensor& w1 = vt(rb + ".c_0_w"); const Tensor& w2 = vt(rb + ".c_1_w");
                    if (c->packed.count(w1.dev) && c->packed.count(w2.dev)) {
                        GemmArgs a = gemm_base(dt);
                        a.M = rows; a.N = Cout; a.K = Cout; a.nbatch = Bs; a.in_len = lens; a.out_len = lens; a.ldw = Cout; a.w_ts = (long)Cout * Cout;
                        a.x_bs = (long)rows * Cout; a.ldx = Cout; a.o_bs = (long)rows * Cout; a.ldo = Cout;
                        a.X = X0s; a.W = w2.dev; a.Wp = c->packed[w2.dev]; a.Wp2 = c->packed[w1.dev];
                        a.bias1 = c->pf(rb + ".c_0_b"); a.slope1 = 0.1f; a.fused = 2;
                        set_taps_1d(a, k, dil[1]);
                        for (int q = 0; q < k; q++) a.dv1[q] = (q - (k - 1) / 2) * dil[0];
                        a.bias = c->pf(rb + ".c_1_b"); a.bias_mode = 1;
                        a.res = X0s; a.r_bs = (long)rows * Cout; a.ldr = Cout; a.res_mode = 2; a.res_inv_slope = 10.0f; a.res_dtype = dt;
                        a.accum = XSs; a.accum_dtype = dt; a.a_bs = (long)rows * Cout; a.lda = Cout;
                        if (nk == 1) { a.accum_mode = 0; a.accum = nullptr; }
                        else if (j == 0) a.accum_mode = 2;
                        else if (j < nk - 1) a.accum_mode = 3;
                        else a.accum_mode = 1;
                        if (j == nk - 1 || nk == 1) { a.out = As; a.out_scale = 1.0f / nk; a.act = ACT_LRELU; a.slope = next_slope; }
                        else a.out = nullptr;
                        a.flops = 2.0 * 2.0 * Bs * (double)rows * Cout * Cout * k;
                        if (gemm_variant_of(a) >= 0) { to_main(); c->gemm(a); t_first = nd; }
                    }
                }
                for (int t = t_first; t < nd; t++) {
                    const bool last = (t == nd - 1);
                    if (last) to_main(); else to_aux();
                    const void* cin_buf = cur;
                    auto rb_base = [&] {
                        GemmArgs a = gemm_base(dt);
                        a.M = rows; a.N = Cout; a.K = Cout; a.nbatch = Bs; a.in_len = lens; a.out_len = lens; a.ldw = Cout; a.w_ts = (long)Cout * Cout;
                        a.x_bs = (long)rows * Cout; a.ldx = Cout; a.o_bs = (long)rows * Cout; a.ldo = Cout;
                        return a;
                    };
                    // what the LAST conv of the iteration does with its result: + bias + x (raw residual recovered from the activated
                    // input), then either the next iteration's input (activated) or the stage's running sum / mean
                    int pp_next = pp;
                    auto rb_tail = [&](GemmArgs& a) {
                        a.bias_mode = 1;
                        a.res = cin_buf; a.r_bs = (long)rows * Cout; a.ldr = Cout; a.res_mode = 2; a.res_inv_slope = 10.0f; a.res_dtype = dt;
                        if (!last) {
                            a.act = ACT_LRELU; a.slope = 0.1f; a.out = PPs[pp];
                            pp_next = pp ^ 1;
                        } else {
                            // xs (+)= resblock output; last kernel size: x = xs / num_kernels, stored activated for the next stage
                            a.accum = XSs; a.accum_dtype = dt; a.a_bs = (long)rows * Cout; a.lda = Cout;
                            if (nk == 1) { a.accum_mode = 0; a.accum = nullptr; }
                            else if (j == 0) a.accum_mode = 2;
                            else if (j < nk - 1) a.accum_mode = 3;
                            else a.accum_mode = 1;
                            if (j == nk - 1) { a.out = As; a.out_scale = 1.0f / nk; a.act = ACT_LRELU; a.slope = next_slope; }
                            else a.out = nullptr;
                        }
                    };
                    GemmArgs a = rb_base();
                    if (c->voc_resblock == 1) {
                        const std::string ts = std::to_string(t);
                        const Tensor& w1 = vt(rb + ".c1_" + ts + "_w");
                        const Tensor& w2 = vt(rb + ".c2_" + ts + "_w");
                        bool fuse = dt != DT_F32 && !c->sat_check && c->packed.count(w1.dev) && c->packed.count(w2.dev);
                        if (fuse) {
                            // one launch: xt = lrelu(c1(x_act)+b1) stays in LDS; x' = c2(xt) + b2 + x      hifigan.py:51-55
                            a.X = cur; a.W = w2.dev; a.Wp = c->packed[w2.dev]; a.Wp2 = c->packed[w1.dev];
                            a.bias1 = c->pf(rb + ".c1_" + ts + "_b"); a.slope1 = 0.1f; a.fused = 1;
                            a.no_pairstream = c->use_pairstream <= 0 ? 1 : (c->use_pairstream == 3 ? 2 : (c->use_pairstream == 4 ? 3 : 0));
                            set_taps_1d(a, k, 1);
                            for (int q = 0; q < k; q++) a.dv1[q] = (q - (k - 1) / 2) * dil[t];
                            a.bias = c->pf(rb + ".c2_" + ts + "_b");
                            a.flops = 2.0 * 2.0 * Bs * (double)rows * Cout * Cout * k;
                            rb_tail(a);
                            // the fused kernels cover a subset of (C, k, dilation, LDS footprint): ask the launcher (dry run) first
                            const int fv = gemm_variant_of(a);
                            // C = 128: the pair kernel, or -- where it declines (small jobs) -- the two conv-slab launches it is bit-identical to:
                            // an utterance must come out the same alone and inside a large batch
                            fuse = fv >= 0 && !(Cout == 128 && fv != 23);
                        }
                        if (!fuse) {
                            // xt = c1(lrelu(x)); stored as lrelu(xt)                       hifigan.py:51-53
                            a = rb_base();
                            a.X = cur; a.W = w1.dev;
                            set_taps_1d(a, k, dil[t]);
                            a.bias = c->pf(rb + ".c1_" + ts + "_b"); a.bias_mode = 1; a.act = ACT_LRELU; a.slope = 0.1f;
                            a.out = T1s;
                            c->gemm(a);
                            // x = c2(.) + x                                                hifigan.py:54-55
                            a = rb_base();
                            a.X = T1s; a.W = w2.dev;
                            set_taps_1d(a, k, 1);
                            a.bias = c->pf(rb + ".c2_" + ts + "_b");
                            rb_tail(a);
                        }
                    } else {
                        // x = c(lrelu(x)) + x                                          hifigan.py:78-81
                        a.X = cur; a.W = vt(rb + ".c_" + std::to_string(t) + "_w").dev;
                        set_taps_1d(a, k, dil[t]);
                        a.bias = c->pf(rb + ".c_" + std::to_string(t) + "_b");
                        rb_tail(a);
                    }
                    c->gemm(a);
                    if (!last) { cur = PPs[pp]; pp = pp_next; }
                }
            }
        }
        Cin = Cout; mul *= u;
    }
    // conv_post + tanh on the first mel_len*hop samples            hifigan.py:127-128, model.py:347
    c->tag = "voc.post";
    dt = dom(ns);
    {
        double nout = 0; for (int b = 0; b < B; b++) nout += (double)mel_len_host[b] * c->hop;
        // rows are zero-filled up to max_b(mel_len) * hop, the extent include/zvx.h promises (and the caller's wav_stride covers):
        // Lmel_max is only the input's row stride and may be larger
        int nmax = 0; for (int b = 0; b < B; b++) nmax = std::max(nmax, mel_len_host[b] * c->hop);
        if ((long)nmax > wav_stride) nmax = (int)wav_stride;
        const int kp = c->t("voc.post_w").dim(0);
        c->timed(2.0 * nout * Cin * kp, nout * (Cin * es + (pcm16 ? 2.0 : 4.0)), [&] {
            launch_conv_post_tanh(A, dt, Cin, (long)Pmax * mul * Cin, c->pf("voc.post_w"), c->t("voc.post_b").host[0], kp,
                                  Cin, wav_dev, wav_stride, pcm16, B, nmax, P_d, c->hop, mel_len_d, c->hop, c->stream);
        });
    }
    c->tag = "other";
}

// ------------------------------------------------------------------------------------------------
// speaker encoder (ResNetSE34V2.py:176-212)
// ------------------------------------------------------------------------------------------------
void run_spkemb(zvx_ctx* c, const float* ref_mels, const int32_t* lens, int B, int Tmax, float* out, int flags) {
    const int dt = c->dt, F0 = c->n_mels, H = c->H;
    const size_t es = c->es();
    for (int b = 0; b < B; b++) if (lens[b] < 2 || lens[b] > Tmax) fail(ZVX_E_INVALID, "ref mel length %d out of range (2..%d)", lens[b], Tmax);
    c->stage_begin(ZVX_T_SPKEMB);
    c->tag = "spkemb";
    float* mels_d = c->fbuf("spk.mels", (size_t)B * Tmax * F0);
    HIPCHK(hipMemcpyAsync(mels_d, ref_mels, (size_t)B * Tmax * F0 * 4, (flags & ZVX_DEVICE_IN) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    // widths per resolution level: w0 = T, w_{l+1} = (w_l - 1)/2 + 1  (3x3 stride-2 pad-1 conv)
    std::vector<int> W(4 * (size_t)B);
    int Wmax[4] = {Tmax, 0, 0, 0};
    for (int b = 0; b < B; b++) { int w = lens[b]; for (int l = 0; l < 4; l++) { W[(size_t)l * B + b] = w; w = (w - 1) / 2 + 1; } }
    for (int l = 1; l < 4; l++) Wmax[l] = (Wmax[l - 1] - 1) / 2 + 1;
    // Map rows are stored one position wider than the widest utterance: that last column is never a valid position, so a
    // stride-1 3 x 3 convolution may run over the FLATTENED map (the neighbour of a row's last position is masked, not the next
    // row's first) -- see launch_gemm.  Every consumer masks columns >= W[b] on the way in.
    int WS[4]; for (int l = 0; l < 4; l++) WS[l] = Wmax[l] + 1;
    int Fh[4]; Fh[0] = F0; for (int l = 1; l < 4; l++) Fh[l] = (Fh[l - 1] - 1) / 2 + 1;
    int* W_d = c->upload_ints("spk.W", W.data(), W.size());

    const int C0 = c->rn_filters[0];
    size_t maxel = 0;
    for (int l = 0; l < 4; l++) maxel = std::max(maxel, (size_t)B * Fh[l] * WS[l] * c->rn_filters[l]);
    void* mA = c->buf("spk.mA", maxel * es); void* mB = c->buf("spk.mB", maxel * es);
    void* mC = c->buf("spk.mC", maxel * es); void* mD = c->buf("spk.mD", maxel * es);
    float* mean = c->fbuf("spk.mean", (size_t)B * 256 * 16); float* rstd = c->fbuf("spk.rstd", (size_t)B * 256 * 16);
    float* sescale = c->fbuf("spk.sescale", (size_t)B * 1024);

    // InstanceNorm1d over time (no affine) folded into the first conv          ResNetSE34V2.py:182-186
    launch_instnorm_stats(mels_d, DT_F32, F0, B, Tmax, W_d, F0, 1e-5f, mean, rstd, c->stream);
    launch_spk_front(mels_d, Tmax, W_d, F0, mean, rstd, c->pf("spk.c1_w"), c->pf("spk.c1_b"), c->pf("spk.bn1_s"), c->pf("spk.bn1_t"), C0,
                     mA, dt, B, WS[0], c->stream);
    void* x = mA; void* o1 = mB; void* o2 = mC; void* rs = mD;
    int lvl = 0, Cin = C0;
    for (size_t li = 0; li < c->rn_layers.size(); li++) {
        const int planes = c->rn_filters[li];
        for (int bi = 0; bi < c->rn_layers[li]; bi++) {
            const std::string p = "spk.l" + std::to_string(li + 1) + "." + std::to_string(bi);
            const int stride = (li > 0 && bi == 0) ? 2 : 1;
            const int lin = lvl, lout = (stride == 2) ? lvl + 1 : lvl;
            const int Hin = Fh[lin], Win = WS[lin], Hout = Fh[lout], Wout = WS[lout];
            const int* win_d = W_d + (size_t)lin * B; const int* wout_d = W_d + (size_t)lout * B;
            int pool_S = 0;                                                   // > 0: conv2's launch wrote the SE pool's partial sums itself
            auto conv3 = [&](const std::string& wn, const void* in, int cin, int hin, int win, const int* inlen, int st, void* out,
                             const float* bias, int act, const float* ps, const float* pt, int ksz, float* se_part = nullptr) {
                GemmArgs a = gemm_base(dt);
                a.X = in; a.x_bs = (long)hin * win * cin; a.ldx = cin; a.W = c->t(wn).dev; a.ldw = cin; a.w_ts = (long)planes * cin;
                a.M = Hout * Wout; a.N = planes; a.K = cin; a.nbatch = B; a.in_len = inlen; a.out_len = wout_d;
                a.stride = st; a.wout = Wout; a.hin = hin; a.win = win;
                a.ntaps = ksz * ksz;
                for (int i = 0; i < ksz; i++) for (int j = 0; j < ksz; j++) { a.du[i * ksz + j] = (int)(i - ksz / 2); a.dv[i * ksz + j] = (int)(j - ksz / 2); }
                if (bias) { a.bias = bias; a.bias_mode = 1; }
                a.act = act; a.post_scale = ps; a.post_shift = pt;
                a.out = out; a.o_bs = (long)Hout * Wout * planes; a.ldo = planes;
                if (se_part) { a.se_part = se_part; a.se_part_S = &pool_S; }
                c->gemm(a);
            };
            // conv1 -> ReLU -> BN1                                          ResNetSE34V2.py:86-88
            // (level transition: conv1 and the shortcut's 1 x 1 convolution, both stride 2 over the same input, in one launch where
            // conv2d_s2_kernel covers the shape)
            bool ds_done = false;
            if (stride == 2 && c->has(p + ".ds") && dt == DT_BF16 && c->spk_s2_fuse) {
                GemmArgs a = gemm_base(dt);
                a.X = x; a.x_bs = (long)Hin * Win * Cin; a.ldx = Cin; a.W = c->t(p + ".c1").dev; a.ldw = Cin; a.w_ts = (long)planes * Cin;
                a.M = Hout * Wout; a.N = planes; a.K = Cin; a.nbatch = B; a.in_len = win_d; a.out_len = wout_d;
                a.stride = 2; a.wout = Wout; a.hin = Hin; a.win = Win; a.ntaps = 9;
                for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { a.du[i * 3 + j] = i - 1; a.dv[i * 3 + j] = j - 1; }
                a.act = ACT_RELU; a.post_scale = c->pf(p + ".bn1_s"); a.post_shift = c->pf(p + ".bn1_t");
                a.out = o1; a.o_bs = (long)Hout * Wout * planes; a.ldo = planes;
                auto itp = c->packed.find(c->t(p + ".ds").dev);
                if (itp != c->packed.end()) {
                    a.ds_out = rs; a.ds_Wp = itp->second; a.ds_bias = c->pf(p + ".ds_b");
                    a.flops = 2.0 * (double)a.M * B * planes * (double)Cin * 10;
                    auto itw = c->packed.find(a.W); if (itw != c->packed.end()) a.Wp = itw->second;
                    a.slab_small = c->slab_small;
                    if (gemm_variant_of(a) >= 0) { c->gemm(a); ds_done = true; }
                }
            }
            if (!ds_done)
            conv3(p + ".c1", x, Cin, Hin, Win, win_d, stride, o1, nullptr, ACT_RELU, c->pf(p + ".bn1_s"), c->pf(p + ".bn1_t"), 3);
            // conv2 (+ folded BN2)                                           :90-91
            // SE: global average pool -> fc -> relu -> fc -> sigmoid        :63-67
            // (the persistent 2-D convolution of the C = 32 / 64 levels leaves the pool's partial sums itself -- one per (row tile, wave):
            // at most ceil(H W / 256) x 4 of them; every other launch is followed by the pool pass)
            const int nsplit = se_pool_splits(Hout, Wout);
            const size_t part_max = std::max((size_t)nsplit, (size_t)((Hout * Wout + 255) / 256) * 4);
            float* separt = c->fbuf("spk.separt", (size_t)B * part_max * planes);
            conv3(p + ".c2", o1, planes, Hout, Wout, wout_d, 1, o2, c->pf(p + ".c2_b"), ACT_NONE, nullptr, nullptr, 3, c->spk_pool_fuse ? separt : nullptr);
            if (pool_S > 0) {
                if ((size_t)pool_S > part_max) fail(ZVX_E_STATE, "fused SE pool wrote %d partials, %zu allocated", pool_S, part_max);
                launch_se_fc(separt, pool_S, Hout, wout_d, c->pf(p + ".se_w1"), c->pf(p + ".se_b1"), c->pf(p + ".se_w2"), c->pf(p + ".se_b2"), planes, planes / 8, sescale, B, c->stream, c->pf(p + ".c2_b"));
            } else {
                launch_se_pool(o2, dt, B, Hout, Wout, wout_d, planes, separt, c->stream);
                launch_se_fc(separt, nsplit, Hout, wout_d, c->pf(p + ".se_w1"), c->pf(p + ".se_b1"), c->pf(p + ".se_w2"), c->pf(p + ".se_b2"), planes, planes / 8, sescale, B, c->stream);
            }
            const void* resid = x;
            if (c->has(p + ".ds")) {                                          // 1x1 stride-s conv + folded BN   :94-95
                if (!ds_done) conv3(p + ".ds", x, Cin, Hin, Win, win_d, stride, rs, c->pf(p + ".ds_b"), ACT_NONE, nullptr, nullptr, 1);
                resid = rs;
            }
            launch_se_apply(o2, resid, o1, dt, sescale, B, Hout, Wout, wout_d, planes, c->stream);     // out*y + residual -> ReLU  :97-98
            std::swap(x, o1);
            lvl = lout; Cin = planes;
        }
    }
    // attention + ASP pooling                                               ResNetSE34V2.py:195-205
    const int Fp = Fh[3], Wp = WS[3], C4 = c->rn_filters[3], D = Fp * C4;
    const int* w3_d = W_d + (size_t)3 * B;
    void* ah = c->buf("spk.ah", (size_t)B * Wp * 128 * es);
    float* logits = c->fbuf("spk.logits", (size_t)B * Wp * D);
    {
        GemmArgs a = gemm_base(dt);       // Conv1d(D -> 128) over features (f, c): one tap per frequency row of the map
        a.X = x; a.x_bs = (long)Fp * Wp * C4; a.ldx = C4; a.W = c->t("spk.att1_w").dev; a.ldw = C4; a.w_ts = (long)128 * C4;
        a.M = Wp; a.N = 128; a.K = C4; a.nbatch = B; a.in_len = w3_d; a.out_len = w3_d;
        a.wout = Wp; a.hin = Fp; a.win = Wp; a.stride = 1;
        a.ntaps = Fp; for (int f = 0; f < Fp; f++) { a.du[f] = (int)f; a.dv[f] = 0; }
        a.bias = c->pf("spk.att1_b"); a.bias_mode = 1; a.act = ACT_RELU; a.post_scale = c->pf("spk.att_bn_s"); a.post_shift = c->pf("spk.att_bn_t");
        a.out = ah; a.o_bs = (long)Wp * 128; a.ldo = 128;
        c->gemm(a);
    }
    {
        GemmArgs a = gemm_base(dt);
        a.X = ah; a.x_bs = (long)Wp * 128; a.ldx = 128; a.W = c->t("spk.att2_w").dev; a.ldw = 128;
        a.M = Wp; a.N = D; a.K = 128; a.nbatch = B; a.in_len = w3_d; a.out_len = w3_d;
        a.bias = c->pf("spk.att2_b"); a.bias_mode = 1;
        a.out = logits; a.out_dtype = DT_F32; a.o_bs = (long)Wp * D; a.ldo = D;
        c->gemm(a);
    }
    const int PD = c->rn_asp ? 2 * D : D;                 // ASP: [mu | sg], SAP: mu     ResNetSE34V2.py:135-143, 199-205
    if (c->t("spk.fc_w").dim(2) != PD) fail(ZVX_E_MANIFEST, "spk.fc_w has %d inputs, the %s pooling produces %d", c->t("spk.fc_w").dim(2), c->rn_asp ? "ASP" : "SAP", PD);
    float* pooled = c->fbuf("spk.pooled", (size_t)B * PD);
    launch_asp_pool(x, dt, logits, B, Fp, Wp, w3_d, C4, pooled, c->rn_asp, c->stream);
    float* emb = c->fbuf("spk.emb", (size_t)B * H);
    // Linear(PD -> hidden): a few rows against a 5120-long K -- one wave per output column      ResNetSE34V2.py:207
    launch_fc_rows(pooled, PD, (const float*)c->t("spk.fc_w").dev, PD, c->pf("spk.fc_b"), emb, H, B, H, PD, c->stream);
    launch_l2norm_rows(emb, B, H, c->stream);                                  // F.normalize   :209-210
    c->stage_end(ZVX_T_SPKEMB);
    HIPCHK(hipMemcpyAsync(out, emb, (size_t)B * H * 4, (flags & ZVX_DEVICE_OUT) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    if (!((flags & ZVX_DEVICE_OUT) && (flags & ZVX_NO_SYNC))) c->sync();
}

// copy device rows [B][rows_max][C] (f32 contiguous) to a caller buffer with row stride
void copy_out_rows(zvx_ctx* c, const float* src, int rows_max, int C, float* dst, long dst_rows_stride, int B, bool device_out) {
    if (!dst) return;
    if (dst_rows_stride < rows_max) fail(ZVX_E_BUFFER, "output row stride %ld < %d rows", dst_rows_stride, rows_max);
    HIPCHK(hipMemcpy2DAsync(dst, (size_t)dst_rows_stride * C * 4, src, (size_t)rows_max * C * 4, (size_t)rows_max * C * 4, B,
                            device_out ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
}

// ------------------------------------------------------------------------------------------------
// log-mel front end of the reference audio (mels.py:357-395 get_mel_from_wav), f32 throughout:
// reflect pad -> |STFT| as a GEMM against the windowed DFT basis (rows = hop-strided frames of the padded signal)
// -> mel basis GEMM -> log(clip(., 1e-5))
// ------------------------------------------------------------------------------------------------
void run_melspec(zvx_ctx* c, const float* wav, const int32_t* nsamples, int B, int Nmax, float* mel_out, int Tmax, int32_t* frames_out) {
    const Tensor& dft = c->t("mel.dft");
    const Tensor& basis = c->t("mel.basis");
    const int n_fft = dft.dim(2), NR = dft.dim(1), KP = basis.dim(2), nf = n_fft / 2 + 1, nm = c->n_mels, hop = c->hop;
    const int pad = (n_fft - hop) / 2;
    std::vector<int> frames(B);
    int Tf = 0;
    for (int b = 0; b < B; b++) {
        const int n = nsamples[b];
        if (n > Nmax || n < pad + 1 || n + 2 * pad < n_fft) fail(ZVX_E_INVALID, "zvx_melspec: utterance %d has %d samples (need %d..%d)", b, n, std::max(pad + 1, n_fft - 2 * pad), Nmax);
        frames[b] = 1 + (n + 2 * pad - n_fft) / hop;
        Tf = std::max(Tf, frames[b]);
    }
    if (Tf > Tmax) fail(ZVX_E_BUFFER, "zvx_melspec: %d frames do not fit Tmax = %d", Tf, Tmax);
    c->stage_begin(ZVX_T_SPKEMB);
    const long Npad = ((long)Nmax + 2 * pad + 3) & ~3L;
    float* wav_d = c->fbuf("mel.wav", (size_t)B * Nmax);
    HIPCHK(hipMemcpyAsync(wav_d, wav, (size_t)B * Nmax * 4, hipMemcpyHostToDevice, c->stream));
    int* n_d = c->upload_ints("mel.n", nsamples, B);
    int* fr_d = c->upload_ints("mel.frames", frames.data(), B);
    float* padded = c->fbuf("mel.pad", (size_t)B * Npad);
    launch_reflect_pad(wav_d, Nmax, n_d, padded, Npad, pad, B, (int)Npad, c->stream);
    float* spec = c->fbuf("mel.spec", (size_t)B * Tf * NR);
    {
        GemmArgs a = gemm_base(DT_F32);
        a.X = padded; a.x_bs = Npad; a.ldx = hop; a.W = dft.dev; a.ldw = n_fft; a.w_ts = (long)NR * n_fft;
        a.M = Tf; a.N = NR; a.K = n_fft; a.nbatch = B; a.in_len = fr_d; a.out_len = fr_d;
        a.out = spec; a.o_bs = (long)Tf * NR; a.ldo = NR;
        c->gemm(a);
    }
    float* mag = c->fbuf("mel.mag", (size_t)B * Tf * KP);
    launch_stft_mag(spec, NR, mag, KP, nf, B, Tf, fr_d, c->stream);
    float* mel_d = c->fbuf("mel.out", (size_t)B * Tf * nm);
    {
        GemmArgs a = gemm_base(DT_F32);
        a.X = mag; a.x_bs = (long)Tf * KP; a.ldx = KP; a.W = basis.dev; a.ldw = KP; a.w_ts = (long)nm * KP;
        a.M = Tf; a.N = nm; a.K = KP; a.nbatch = B; a.in_len = fr_d; a.out_len = fr_d;
        a.out = mel_d; a.o_bs = (long)Tf * nm; a.ldo = nm;
        c->gemm(a);
    }
    launch_log_clip(mel_d, nm, nm, 1e-5f, B, Tf, fr_d, c->stream);
    c->stage_end(ZVX_T_SPKEMB);
    if (mel_out) {
        if (Tmax > Tf) memset(mel_out, 0, (size_t)B * Tmax * nm * 4);
        copy_out_rows(c, mel_d, Tf, nm, mel_out, Tmax, B, false);
    }
    if (frames_out) memcpy(frames_out, frames.data(), B * sizeof(int));
    c->sync();
}

// wav: float rows, or int16 PCM rows with ZVX_PCM16 (stride counted in samples either way).  Row b receives
// mel_len[b]*hop samples followed by zeros up to max_b(mel_len[b])*hop; nothing beyond that is touched.
void do_vocode(zvx_ctx* c, const int32_t* pad_to, void* wav, int64_t wav_stride, int flags) {
    if (!c->have_mel) fail(ZVX_E_STATE, "zvx_vocode: no mel in the context (call zvx_decode first)");
    const int B = c->B;
    std::vector<int> P(B);
    int need = 0;
    for (int b = 0; b < B; b++) { P[b] = std::max(pad_to ? pad_to[b] : 0, c->mel_len_host[b]); need = std::max(need, c->mel_len_host[b] * c->hop); }
    const bool host_async = flags & ZVX_HOST_ASYNC;
    if (!host_async && wav_stride < need) fail(ZVX_E_BUFFER, "wav_stride %lld < %d samples", (long long)wav_stride, need);
    c->stage_begin(ZVX_T_VOCODER);
    void* wdev; long wstride;
    const bool dev_out = (flags & ZVX_DEVICE_OUT) && !host_async;
    const int pcm16 = (flags & ZVX_PCM16) ? 1 : 0;
    const size_t ss = pcm16 ? 2 : 4;
    zvx_ctx::HostSlot* hs = nullptr;
    if (host_async) {
        // slot i & 1 of call i: its previous copy (two calls back) has normally long landed and been read; the device-side staging rows are
        // per slot as well, so this vocoder's conv_post never writes what the previous call's copy is still reading
        const int slot = c->host_next; c->host_next ^= 1; c->host_last = slot;
        hs = &c->host_slot[slot];
        // The copies ride the communication stream where the context has one (world 1: it is idle; a multi-GPU rank gathers on the device
        // instead of delivering to its host): the runtime maps streams onto FOUR hardware queues by default (GPU_MAX_HW_QUEUES), and the
        // context's main / front / predictor / communication streams are four -- a fifth stream shares a queue with one of them, and its
        // wait-for-the-vocoder barrier then holds up that stream's launches too (measured: the whole front-end overlap lost, +1.2 ms)
        if (!c->copy_stream) { if (c->comm_stream) { c->copy_stream = c->comm_stream; c->copy_is_comm = true; } else HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking)); }
        if (!hs->ready) { HIPCHK(hipEventCreateWithFlags(&hs->ready, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&hs->done, hipEventDisableTiming)); }
        if (hs->pending) { HIPCHK(hipEventSynchronize(hs->done)); hs->pending = false; }
        wstride = (std::max(need, 1) + 7) & ~7;
        const size_t bytes = (size_t)B * wstride * ss;
        if (bytes > hs->cap) {
            if (hs->p) { HIPCHK(hipHostFree(hs->p)); hs->p = nullptr; hs->cap = 0; }
            const size_t cap = bytes + bytes / 8 + 256;
            HIPCHK(hipHostMalloc(&hs->p, cap, hipHostMallocDefault));
            hs->cap = cap;
        }
        wdev = c->buf(slot ? "wav.async1" : "wav.async0", bytes);
        hs->B = B; hs->pcm16 = pcm16; hs->stride = wstride; hs->need = need;
    } else if (dev_out) {
        wdev = wav; wstride = wav_stride;
        auto f = c->gather_fence.find(wav);                 // a gather still reading this buffer: the vocoder's writes queue behind it
        if (f != c->gather_fence.end()) HIPCHK(hipStreamWaitEvent(c->stream, f->second, 0));
    }
    else { wstride = (std::max(need, 1) + 7) & ~7; wdev = c->buf("wav", (size_t)B * wstride * ss); }
    run_vocoder(c, c->fbuf("mel", 0), c->n_mels, c->Lmax, c->mel_len_host.data(), P.data(), B, wdev, wstride, pcm16);
    c->stage_end(ZVX_T_VOCODER);
    if (host_async) {
        HIPCHK(hipEventRecord(hs->ready, c->stream));
        HIPCHK(hipStreamWaitEvent(c->copy_stream, hs->ready, 0));
        if (need > 0) HIPCHK(hipMemcpyAsync(hs->p, wdev, (size_t)B * wstride * ss, hipMemcpyDeviceToHost, c->copy_stream));
        HIPCHK(hipEventRecord(hs->done, c->copy_stream));
        hs->pending = true;
        return;                                              // queued: zvx_wait_host(slot) is where the host meets the rows
    }
    if (!dev_out && need > 0)
        HIPCHK(hipMemcpy2DAsync(wav, (size_t)wav_stride * ss, wdev, (size_t)wstride * ss, (size_t)need * ss, B, hipMemcpyDeviceToHost, c->stream));
    if (!(dev_out && (flags & ZVX_NO_SYNC))) c->sync();
}

template <typename F>
zvx_status guarded(zvx_ctx* ctx, F&& f) {
    if (!ctx) return ZVX_E_INVALID;
    try {
        HIPCHK(hipSetDevice(ctx->device));
        const bool outer = ctx->api_depth++ == 0;               // (an entry point that calls another one keeps the outer call's arena)
        struct End { zvx_ctx* c; ~End() { try { if (--c->api_depth == 0) c->arena_end(); } catch (...) {} } } end{ctx};
        if (outer) ctx->arena_begin();
        f();
        return ZVX_OK;
    } catch (const ZvxError& e) {
        ctx->err = e.what();
        ctx->quiesce_side_streams();
        return e.code;
    } catch (const std::exception& e) {
        ctx->err = e.what();
        ctx->quiesce_side_streams();
        return ZVX_E_INVALID;
    }
}

}  // namespace

// ================================================================================================
// C-ABI
// ================================================================================================
extern "C" {

zvx_status zvx_create(const char* manifest, const void* weights, size_t nbytes, int device, zvx_ctx** out) {
    if (!out) return ZVX_E_INVALID;
    *out = nullptr;
    zvx_ctx* c = new zvx_ctx();
    try {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0) fail(ZVX_E_HIP, "no HIP device available (%s)", hipGetErrorString(e));
        if (device < 0 || device >= ndev) fail(ZVX_E_INVALID, "device %d out of range (%d visible)", device, ndev);
        c->device = device;
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreate(&c->stream));
        c->main0 = c->stream;
        for (int s = 0; s < ZVX_T_COUNT; s++) { HIPCHK(hipEventCreate(&c->stage_ev[s][0])); HIPCHK(hipEventCreate(&c->stage_ev[s][1])); c->stage_used[s] = false; c->stage_ms[s] = 0.f; }
        parse_manifest(c, manifest, weights, nbytes);
        read_config(c);
        upload_weights(c);
        *out = c;
        return ZVX_OK;
    } catch (const ZvxError& e) {
        g_create_error = e.what();
        delete c;
        return e.code;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        delete c;
        return ZVX_E_INVALID;
    }
}

void zvx_destroy(zvx_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->front_stream) (void)hipStreamSynchronize(c->front_stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    zvx_comm_destroy(c);
    for (int i = 0; i < 2; i++) { if (c->arena[i].p) (void)hipHostFree(c->arena[i].p); if (c->arena[i].ev) (void)hipEventDestroy(c->arena[i].ev); }
    if (c->copy_stream && !c->copy_is_comm) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (auto& hs : c->host_slot) { if (hs.p) (void)hipHostFree(hs.p); if (hs.ready) (void)hipEventDestroy(hs.ready); if (hs.done) (void)hipEventDestroy(hs.done); }
    for (int i = 0; i < 2; i++) if (c->voc_aux[i]) { (void)hipStreamSynchronize(c->voc_aux[i]); (void)hipStreamDestroy(c->voc_aux[i]); }
    for (int i = 0; i < 3; i++) if (c->voc_ev[i]) (void)hipEventDestroy(c->voc_ev[i]);
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); (void)hipEventDestroy(c->ev_aux[0]); (void)hipEventDestroy(c->ev_aux[1]); }
    if (c->front_stream) {
        (void)hipStreamSynchronize(c->front_stream); (void)hipStreamDestroy(c->front_stream);
        (void)hipEventDestroy(c->ev_front_done); (void)hipEventDestroy(c->ev_mel_free); (void)hipEventDestroy(c->ev_main_join);
    }
    for (auto& kv : c->bufs) if (kv.second.base) (void)hipFree(kv.second.base);
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->stream) {
        for (int s = 0; s < ZVX_T_COUNT; s++) { (void)hipEventDestroy(c->stage_ev[s][0]); (void)hipEventDestroy(c->stage_ev[s][1]); }
        (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

const char* zvx_last_error(const zvx_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int64_t zvx_get_int(const zvx_ctx* c, const char* key) {
    if (!c || !key) return -1;
    const std::string k(key);
    if (k == "precision") return c->dt == DT_F32 ? 1 : 0;
    if (k == "hidden") return c->H;
    if (k == "n_mels") return c->n_mels;
    if (k == "hop") return c->hop;
    if (k == "device") return c->device;
    if (k == "dec_kind") return c->dec_kind;
    if (k == "Lmax") return c->Lmax;
    if (k == "profile") return c->profile;
    if (k == "profile_only") return c->profile_only;
    if (k == "f16_sat_check") return c->sat_check;
    if (k == "f16_sat_events") {                             // clamped 16-bit stores seen since zvx_set_int("f16_sat_check", 1); drains the context's streams
        zvx_ctx* m = const_cast<zvx_ctx*>(c);
        unsigned long long n = 0;
        if (hipSetDevice(m->device) != hipSuccess) return -1;
        try { unsigned long long* d = m->sat_count_dev(); m->sync(); if (hipMemcpy(&n, d, sizeof n, hipMemcpyDeviceToHost) != hipSuccess) return -1; } catch (...) { return -1; }
        return (int64_t)n;
    }
    if (k == "host_slot") return c->host_last;             // the slot the last ZVX_HOST_ASYNC call writes (-1: none yet)
    if (k.rfind("variant_id:", 0) == 0) { for (int i = 0; i < gemm_num_variants(); i++) if (k.substr(11) == gemm_variant_name(i)) return i; return -1; }
    auto it = c->cfg.find(k);
    if (it != c->cfg.end()) return atoll(it->second.c_str());
    return -1;
}

zvx_status zvx_set_int(zvx_ctx* c, const char* key, int64_t value) {
    return guarded(c, [&] {
        if (!key) fail(ZVX_E_INVALID, "key is NULL");
        if (std::string(key) == "profile") { c->sync(); c->profile = (int)value; }
        else if (std::string(key) == "profile_only") { c->sync(); c->profile_only = (int)value; }
        else if (std::string(key) == "shape_log") c->shape_log = (int)value;
        else if (std::string(key) == "resstream") c->use_resstream = (int)value;
        else if (std::string(key) == "pairstream") { if (value == 2) fail(ZVX_E_INVALID, "pairstream 2 (k = 3 on the register-resident pair kernel) was removed in round 6"); c->use_pairstream = (int)value; }
        else if (std::string(key) == "voc_chunk") c->voc_chunk = (int)value;
        else if (std::string(key) == "flash") c->use_flash = (int)value;
        else if (std::string(key) == "attn_f32") c->use_attn_f32 = (int)value;
        else if (std::string(key) == "dec_f16") c->dec_f16 = (int)value;
        else if (std::string(key) == "dec_y16") c->dec_y16 = (int)value;
        else if (std::string(key) == "dec_qkv") c->dec_qkv = (int)value;
        else if (std::string(key) == "voc_f16") c->voc_f16 = (int)value;
        else if (std::string(key) == "voc_f16_stages") c->voc_f16_stages = (int)value;
        else if (std::string(key) == "stagefuse") c->use_stagefuse = (int)value;
        else if (std::string(key) == "rb2fuse") c->use_rb2fuse = (int)value;
        else if (std::string(key) == "f16_sat_check") {       // (re)arms the audit and zeroes its counter
            c->sync(); c->sat_check = value ? 1 : 0;
            HIPCHK(hipMemsetAsync(c->sat_count_dev(), 0, 64, c->stream));
        }
        else if (std::string(key) == "poison_pads") c->poison_pads = (int)value;
        else if (std::string(key) == "dec_flat") c->dec_flat = (int)value;
        else if (std::string(key) == "dec_sc_fuse") c->dec_sc_fuse = (int)value;
        else if (std::string(key) == "norm_fuse_maxb") c->norm_fuse_maxb = (int)value;
        else if (std::string(key) == "va_overlap_maxb") c->va_overlap_maxb = (int)value;
        else if (std::string(key) == "voc_overlap_maxb") c->voc_overlap_maxb = (int)value;
        else if (std::string(key) == "voc_overlap_frames") c->voc_overlap_frames = (long)value;
        else if (std::string(key) == "enc_split") {
            if (value < 0 || value > 2) fail(ZVX_E_INVALID, "enc_split: 0 (exact f32), 1 (bf16 planes) or 2 (half planes)");
            c->enc_split = c->dt == DT_BF16 ? (int)value : 0;
            if (c->enc_split) { c->sync(); build_split_weights(c, c->enc_split == 2); }
        }
        else if (std::string(key) == "front_overlap") { c->sync(); c->front_overlap = (int)value; c->front_dirty_main = true; }
        else if (std::string(key) == "front_prio") {
            c->sync(); c->front_prio = (int)value; c->front_dirty_main = true; c->mel_free_pending = false;
            if (c->front_stream) {                                           // re-created with the new priority on the next zvx_synthesize
                (void)hipStreamDestroy(c->front_stream); c->front_stream = nullptr;
                (void)hipEventDestroy(c->ev_front_done); (void)hipEventDestroy(c->ev_mel_free); (void)hipEventDestroy(c->ev_main_join);
            }
        }
        else if (std::string(key) == "rs_prof") c->rs_prof = (int)value;
        else if (std::string(key) == "rs_opt") c->rs_opt = (int)value;
        else if (std::string(key) == "rs_seg_min") c->rs_seg_min = (int)value;
        else if (std::string(key) == "slab_small") c->slab_small = (int)value;
        else if (std::string(key) == "spk_pool_fuse") c->spk_pool_fuse = (int)value;
        else if (std::string(key) == "spk_s2_fuse") c->spk_s2_fuse = (int)value;
        else if (std::string(key) == "slab_flat") c->slab_flat = (int)value;
        else if (std::string(key) == "max_frames") { if (value < 1 || value > (1 << 24)) fail(ZVX_E_INVALID, "max_frames out of range"); c->max_frames = (int)value; }
        else fail(ZVX_E_INVALID, "unknown option '%s'", key);
    });
}

zvx_status zvx_spkemb(zvx_ctx* c, const float* ref_mels, const int32_t* lens, int B, int Tmax, float* out) {
    return guarded(c, [&] {
        if (!ref_mels || !lens || !out || B <= 0 || Tmax <= 0) fail(ZVX_E_INVALID, "zvx_spkemb: bad arguments");
        run_spkemb(c, ref_mels, lens, B, Tmax, out, 0);
    });
}

zvx_status zvx_spkemb_ex(zvx_ctx* c, const float* ref_mels, const int32_t* lens, int B, int Tmax, float* out, int flags) {
    return guarded(c, [&] {
        if (!ref_mels || !lens || !out || B <= 0 || Tmax <= 0) fail(ZVX_E_INVALID, "zvx_spkemb_ex: bad arguments");
        run_spkemb(c, ref_mels, lens, B, Tmax, out, flags);
    });
}

zvx_status zvx_melspec(zvx_ctx* c, const float* wav, const int32_t* nsamples, int B, int Nmax, float* mel, int Tmax, int32_t* frames) {
    return guarded(c, [&] {
        if (!wav || !nsamples || B <= 0 || Nmax <= 0 || Tmax <= 0) fail(ZVX_E_INVALID, "zvx_melspec: bad arguments");
        run_melspec(c, wav, nsamples, B, Nmax, mel, Tmax, frames);
    });
}

zvx_status zvx_encode(zvx_ctx* c, const int32_t* phoneme, const int32_t* puncts, const int32_t* duration, const int32_t* T,
                      int B, int Tmax, const float* spk, int32_t* mel_len, float* log_duration, float* pitch, float* energy) {
    return guarded(c, [&] {
        if (!phoneme || !puncts || !T || !spk) fail(ZVX_E_INVALID, "zvx_encode: NULL input");
        run_encode(c, phoneme, puncts, duration, T, B, Tmax, spk, mel_len, 0);
        const size_t nid = (size_t)B * Tmax;
        if (log_duration) HIPCHK(hipMemcpyAsync(log_duration, c->fbuf("va.logd", nid), nid * 4, hipMemcpyDeviceToHost, c->stream));
        if (pitch) HIPCHK(hipMemcpyAsync(pitch, c->fbuf("va.pitch", nid), nid * 4, hipMemcpyDeviceToHost, c->stream));
        if (energy) HIPCHK(hipMemcpyAsync(energy, c->fbuf("va.energy", nid), nid * 4, hipMemcpyDeviceToHost, c->stream));
        c->sync();
    });
}

zvx_status zvx_decode(zvx_ctx* c, float* mel_out, int Lstride, int flags) {
    return guarded(c, [&] {
        if (!c->have_features) fail(ZVX_E_STATE, "zvx_decode: no features in the context (call zvx_encode first)");
        const int B = c->B;
        int* L_d = c->upload_ints("dec.L", c->mel_len_host.data(), B);
        run_decode(c, c->fbuf("features", 0), c->fbuf("in.spk", 0), L_d, B, c->Lmax);
        if (mel_out && c->Lmax > 0) copy_out_rows(c, c->fbuf("mel", 0), c->Lmax, c->n_mels, mel_out, Lstride, B, flags & ZVX_DEVICE_OUT);
        c->sync();
    });
}

zvx_status zvx_decode_features(zvx_ctx* c, const float* features, const int32_t* L, int B, int Lmax, const float* spk,
                               float* mel_out, int Lstride) {
    return guarded(c, [&] {
        if (!features || !L || !spk || B <= 0 || Lmax <= 0) fail(ZVX_E_INVALID, "zvx_decode_features: bad arguments");
        for (int b = 0; b < B; b++) if (L[b] < 2 || L[b] > Lmax) fail(ZVX_E_INVALID, "L[%d]=%d out of range (2..%d)", b, L[b], Lmax);
        c->B = B; c->Lmax = Lmax; c->mel_len_host.assign(L, L + B); c->Tmax = 0;
        c->have_features = false; c->have_mel = false;
        float* f = c->fbuf("features", (size_t)B * Lmax * c->H);
        HIPCHK(hipMemcpyAsync(f, features, (size_t)B * Lmax * c->H * 4, hipMemcpyHostToDevice, c->stream));
        float* spk_d = c->fbuf("in.spk", (size_t)B * c->H);
        HIPCHK(hipMemcpyAsync(spk_d, spk, (size_t)B * c->H * 4, hipMemcpyHostToDevice, c->stream));
        int* L_d = c->upload_ints("dec.L", L, B);
        c->have_features = true;
        run_decode(c, f, spk_d, L_d, B, Lmax);
        if (mel_out) copy_out_rows(c, c->fbuf("mel", 0), Lmax, c->n_mels, mel_out, Lstride, B, false);
        c->sync();
    });
}

zvx_status zvx_vocode(zvx_ctx* c, const int32_t* pad_to, void* wav, int64_t wav_stride, int flags) {
    return guarded(c, [&] {
        if (!wav && !(flags & ZVX_HOST_ASYNC)) fail(ZVX_E_INVALID, "zvx_vocode: wav is NULL");
        do_vocode(c, pad_to, wav, wav_stride, flags);
    });
}

zvx_status zvx_vocode_mel(zvx_ctx* c, const float* mel, const int32_t* P, int B, int Pmax, void* wav, int64_t wav_stride, int flags) {
    return guarded(c, [&] {
        if (!mel || !P || (!wav && !(flags & ZVX_HOST_ASYNC)) || B <= 0 || Pmax <= 0) fail(ZVX_E_INVALID, "zvx_vocode_mel: bad arguments");
        for (int b = 0; b < B; b++) if (P[b] < 1 || P[b] > Pmax) fail(ZVX_E_INVALID, "P[%d]=%d out of range (1..%d)", b, P[b], Pmax);
        c->B = B; c->Lmax = Pmax; c->Tmax = 0; c->mel_len_host.assign(P, P + B);
        c->have_features = false; c->have_mel = false;             // the context's batch geometry changes: earlier intermediates are void
        float* m = c->fbuf("mel", (size_t)B * Pmax * c->n_mels);
        HIPCHK(hipMemcpyAsync(m, mel, (size_t)B * Pmax * c->n_mels * 4, (flags & ZVX_DEVICE_IN) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
        c->have_mel = true;
        do_vocode(c, nullptr, wav, wav_stride, flags);
    });
}

zvx_status zvx_synthesize(zvx_ctx* c, const int32_t* phoneme, const int32_t* puncts, const int32_t* duration, const int32_t* T,
                          int B, int Tmax, const float* spk, const int32_t* pad_to, int Lmax_cap, void* wav, int64_t wav_stride,
                          int32_t* mel_len, float* mel_out, int Lstride, float* log_duration, int flags) {
    return guarded(c, [&] {
        const bool host_async = flags & ZVX_HOST_ASYNC;
        if (!phoneme || !puncts || !T || !spk || (!wav && !host_async)) fail(ZVX_E_INVALID, "zvx_synthesize: NULL input");
        if (host_async && ((mel_out && !(flags & ZVX_DEVICE_OUT)) || log_duration)) fail(ZVX_E_INVALID, "zvx_synthesize: ZVX_HOST_ASYNC takes no host mel / log_duration output (they would make the call wait)");
        auto front_end = [&] {
            run_encode(c, phoneme, puncts, duration, T, B, Tmax, spk, mel_len, Lmax_cap);
            if (log_duration) HIPCHK(hipMemcpyAsync(log_duration, c->fbuf("va.logd", 0), (size_t)B * Tmax * 4, hipMemcpyDeviceToHost, c->stream));
            int* L_d = c->upload_ints("dec.L", c->mel_len_host.data(), B);
            run_decode(c, c->fbuf("features", 0), c->fbuf("in.spk", 0), L_d, B, c->Lmax);
            if (mel_out && c->Lmax > 0) copy_out_rows(c, c->fbuf("mel", 0), c->Lmax, c->n_mels, mel_out, Lstride, B, flags & ZVX_DEVICE_OUT);
        };
        // a call that waits for its own result (no ZVX_NO_SYNC) has nothing to overlap with: it stays on the one stream (two streams
        // cost an event round trip, which a single short request would see); front_overlap 2 forces the two-stream schedule (tests)
        if (c->front_overlap == 2 || (c->front_overlap == 1 && (host_async || ((flags & ZVX_NO_SYNC) && (flags & ZVX_DEVICE_OUT))))) {
            // the front end on its own stream (see zvx_ctx::front_stream): ordered behind (1) whatever other entry points did to the
            // front-end buffers on the main stream, (2) the previous vocoder's read of the mel buffer -- and NOT behind that vocoder
            c->front_setup();
            hipStream_t const main_stream = c->stream;
            if (c->front_dirty_main) {
                HIPCHK(hipEventRecord(c->ev_main_join, main_stream));
                HIPCHK(hipStreamWaitEvent(c->front_stream, c->ev_main_join, 0));
                c->front_dirty_main = false; c->mel_free_pending = false;       // (the join covers the mel read too)
            }
            if (c->mel_free_pending) { HIPCHK(hipStreamWaitEvent(c->front_stream, c->ev_mel_free, 0)); c->mel_free_pending = false; }
            {
                struct Swap { zvx_ctx* c; hipStream_t keep; ~Swap() { c->stream = keep; } } sw{c, main_stream};   // restored on every path out
                c->stream = c->front_stream;
                front_end();
                HIPCHK(hipEventRecord(c->ev_front_done, c->front_stream));
            }
            HIPCHK(hipStreamWaitEvent(main_stream, c->ev_front_done, 0));
        } else {
            front_end();
        }
        do_vocode(c, pad_to, wav, wav_stride, flags);
    });
}

zvx_status zvx_fetch(zvx_ctx* c, const char* what, float* out, size_t out_floats) {
    return guarded(c, [&] {
        if (!what || !out) fail(ZVX_E_INVALID, "zvx_fetch: NULL argument");
        const std::string w(what);
        const size_t nid = (size_t)c->B * c->Tmax;
        auto copy_f = [&](const char* name, size_t n) {
            if (out_floats < n) fail(ZVX_E_BUFFER, "zvx_fetch('%s'): need %zu floats", what, n);
            HIPCHK(hipMemcpyAsync(out, c->fbuf(name, n), n * 4, hipMemcpyDeviceToHost, c->stream));
            c->sync();
        };
        auto copy_i = [&](const char* name, size_t n) {
            if (out_floats < n) fail(ZVX_E_BUFFER, "zvx_fetch('%s'): need %zu floats", what, n);
            std::vector<int> tmp(n);
            HIPCHK(hipMemcpyAsync(tmp.data(), c->ibuf(name, n), n * 4, hipMemcpyDeviceToHost, c->stream));
            c->sync();
            for (size_t i = 0; i < n; i++) out[i] = (float)tmp[i];
        };
        if (w == "encoder_out") copy_f("enc.out", nid * c->H);
        else if (w == "features") copy_f("features", (size_t)c->B * c->Lmax * c->H);
        else if (w == "mel") copy_f("mel", (size_t)c->B * c->Lmax * c->n_mels);
        else if (w == "pitch") copy_f("va.pitch", nid);
        else if (w == "energy") copy_f("va.energy", nid);
        else if (w == "log_duration") copy_f("va.logd", nid);
        else if (w == "pitch_idx") copy_i("va.pitch_idx", nid);
        else if (w == "energy_idx") copy_i("va.energy_idx", nid);
        else if (w == "duration") copy_i("va.dur", nid);
        else if (w.rfind("buf:", 0) == 0) {                              // raw bytes of a named context buffer (development aid)
            auto it = c->bufs.find(w.substr(4));
            if (it == c->bufs.end()) fail(ZVX_E_INVALID, "zvx_fetch: no buffer '%s'", what);
            const size_t nb = std::min(out_floats * 4, it->second.cap);
            HIPCHK(hipMemcpyAsync(out, it->second.p, nb, hipMemcpyDeviceToHost, c->stream));
            c->sync();
        }
        else fail(ZVX_E_INVALID, "zvx_fetch: unknown tensor '%s'", what);
    });
}

zvx_status zvx_sync(zvx_ctx* c) { return guarded(c, [&] { c->sync(); }); }

zvx_status zvx_wait_host(zvx_ctx* c, int slot, const void** rows, int64_t* stride, int32_t* nrows, int64_t* valid) {
    return guarded(c, [&] {
        if (slot < 0 || slot > 1) fail(ZVX_E_INVALID, "zvx_wait_host: slot %d (0 or 1)", slot);
        zvx_ctx::HostSlot& hs = c->host_slot[slot];
        if (!hs.done || !hs.p) fail(ZVX_E_STATE, "zvx_wait_host: no ZVX_HOST_ASYNC call has used slot %d", slot);
        if (hs.pending) { HIPCHK(hipEventSynchronize(hs.done)); hs.pending = false; }
        if (rows) *rows = hs.p;
        if (stride) *stride = hs.stride;
        if (nrows) *nrows = hs.B;
        if (valid) *valid = hs.need;
    });
}

// ---- RCCL, resolved at run time ----------------------------------------------------------------------------------
extern "C++" {
namespace {
struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;            // optional (zvx_comm_info)
    decltype(&ncclGetVersion) GetVersion = nullptr;
};
Rccl& rccl() {
    static Rccl r;
    if (r.h) return r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
    if (!r.h) fail(ZVX_E_HIP, "cannot load librccl.so (%s): multi-GPU needs RCCL", dlerror());
#define ZVX_SYM(f) do { r.f = (decltype(r.f))dlsym(r.h, "nccl" #f); if (!r.f) fail(ZVX_E_HIP, "librccl.so lacks nccl" #f); } while (0)
    ZVX_SYM(GetUniqueId); ZVX_SYM(CommInitRank); ZVX_SYM(CommDestroy); ZVX_SYM(Send); ZVX_SYM(Recv); ZVX_SYM(GroupStart); ZVX_SYM(GroupEnd);
    ZVX_SYM(AllReduce); ZVX_SYM(GetErrorString);
#undef ZVX_SYM
    r.CommCount = (decltype(r.CommCount))dlsym(r.h, "ncclCommCount");
    r.GetVersion = (decltype(r.GetVersion))dlsym(r.h, "ncclGetVersion");
    return r;
}
#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) fail(ZVX_E_HIP, "%s failed: %s (%s:%d)", #x, rccl().GetErrorString(r_), __FILE__, __LINE__); } while (0)
}  // namespace
}  // extern "C++"

zvx_status zvx_comm_unique_id(void* id_out) {
    static_assert(sizeof(ncclUniqueId) == ZVX_COMM_ID_BYTES, "ncclUniqueId size");
    if (!id_out) return ZVX_E_INVALID;
    try {
        ncclUniqueId id;
        NCCLCHK(rccl().GetUniqueId(&id));
        memcpy(id_out, &id, sizeof id);
        return ZVX_OK;
    } catch (const ZvxError& e) { g_create_error = e.what(); return e.code; }
}

zvx_status zvx_comm_init(zvx_ctx* c, const void* id, int rank, int world) {
    return guarded(c, [&] {
        if (world < 1 || rank < 0 || rank >= world) fail(ZVX_E_INVALID, "zvx_comm_init: rank %d of %d", rank, world);
        if (c->comm || c->comm_stream) fail(ZVX_E_STATE, "zvx_comm_init: communicator already initialised");
        HIPCHK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->ev_compute, hipEventDisableTiming));
        c->rank = rank; c->world = world;
        if (world > 1 || id) {                              // world == 1 with an id: a real one-rank communicator (self-test of the RCCL path)
            if (!id) fail(ZVX_E_INVALID, "zvx_comm_init: id is NULL");
            ncclUniqueId uid;
            memcpy(&uid, id, sizeof uid);
            NCCLCHK(rccl().CommInitRank(&c->comm, world, uid, rank));
        }
    });
}

zvx_status zvx_comm_gather(zvx_ctx* c, const void* local, size_t bytes, void* recv, int root, int flags) {
    return guarded(c, [&] {
        if (!c->comm_stream) fail(ZVX_E_STATE, "zvx_comm_gather: call zvx_comm_init first");
        if (!local || root < 0 || root >= c->world || (c->rank == root && !recv)) fail(ZVX_E_INVALID, "zvx_comm_gather: bad arguments");
        // the communication stream picks up after everything issued so far on the compute stream
        HIPCHK(hipEventRecord(c->ev_compute, c->stream));
        HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_compute, 0));
        if (c->comm && c->world == 1) {                     // one-rank communicator: the root's row travels through RCCL too
            Rccl& r = rccl();
            NCCLCHK(r.GroupStart());
            NCCLCHK(r.Send(local, bytes, ncclInt8, 0, c->comm, c->comm_stream));
            NCCLCHK(r.Recv(recv, bytes, ncclInt8, 0, c->comm, c->comm_stream));
            NCCLCHK(r.GroupEnd());
        } else if (c->rank == root)
            HIPCHK(hipMemcpyAsync((char*)recv + (size_t)root * bytes, local, bytes, hipMemcpyDeviceToDevice, c->comm_stream));
        if (c->world > 1) {
            Rccl& r = rccl();
            NCCLCHK(r.GroupStart());
            if (c->rank == root) {
                for (int p = 0; p < c->world; p++)
                    if (p != root) NCCLCHK(r.Recv((char*)recv + (size_t)p * bytes, bytes, ncclInt8, p, c->comm, c->comm_stream));
            } else {
                NCCLCHK(r.Send(local, bytes, ncclInt8, root, c->comm, c->comm_stream));
            }
            NCCLCHK(r.GroupEnd());
        }
        // later writers of `local` (the vocoder of a coming call) queue behind this event on the device
        hipEvent_t& ev = c->gather_fence[local];
        if (!ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev, c->comm_stream));
        if (!(flags & ZVX_NO_SYNC)) c->sync();
    });
}

zvx_status zvx_comm_max_f64(zvx_ctx* c, double* value) {
    return guarded(c, [&] {
        if (!c->comm_stream || !value) fail(ZVX_E_STATE, "zvx_comm_max_f64: call zvx_comm_init first");
        double* d = (double*)c->buf("comm.scalar", 64);      // (a first use zero-fills it on the compute stream: taken BEFORE the drain below,
        c->sync();                                           //  the communication stream must not see that fill land on top of its copy)
        if (!c->comm) return;
        HIPCHK(hipMemcpyAsync(d, value, sizeof(double), hipMemcpyHostToDevice, c->comm_stream));
        NCCLCHK(rccl().AllReduce(d, d, 1, ncclFloat64, ncclMax, c->comm, c->comm_stream));
        HIPCHK(hipMemcpyAsync(value, d, sizeof(double), hipMemcpyDeviceToHost, c->comm_stream));
        HIPCHK(hipStreamSynchronize(c->comm_stream));
    });
}

// Who is in the job, as seen by the communicator itself (bench.py puts it into the N > 1 JSON line so that the first multi-GPU
// run is self-diagnosing): out[0] = world of this context, out[1] = ncclCommCount, out[2] = ncclGetVersion code,
// out[3] = number of ranks that contributed to an all-reduce SUM of ones (a live data-path check), out[4 + r] = PCI address of
// rank r's device ((domain << 16) | (bus << 8) | (device << 3) | function), gathered with an all-reduce MAX over per-rank slots.
// Collective: every rank calls it.  Without a communicator (world 1, no id) only this rank's entries are filled.
zvx_status zvx_comm_info(zvx_ctx* c, int64_t* out, int n_out) {
    return guarded(c, [&] {
        if (!c->comm_stream || !out) fail(ZVX_E_STATE, "zvx_comm_info: call zvx_comm_init first");
        const int W = c->world;
        if (n_out < 4 + W) fail(ZVX_E_BUFFER, "zvx_comm_info: need %d entries", 4 + W);
        double* const d = (double*)c->buf("comm.info", (size_t)(W + 1) * 8);      // (allocated -- and zero-filled on the compute stream -- before the drain)
        c->sync();
        for (int i = 0; i < n_out; i++) out[i] = -1;
        out[0] = W;
        char bus[64] = {0};
        long pci = -1;
        if (hipDeviceGetPCIBusId(bus, sizeof bus, c->device) == hipSuccess) {
            unsigned dom = 0, b = 0, d = 0, f = 0;
            if (sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) >= 3) pci = ((long)dom << 16) | ((long)b << 8) | ((long)d << 3) | f;
        }
        std::vector<double> v((size_t)W + 1, 0.0);
        v[0] = 1.0; v[1 + c->rank] = (double)(pci + 1);          // + 1: an address of 0 still reads as "present"
        if (c->comm) {
            Rccl& r = rccl();
            int n = -1, ver = -1;
            if (r.CommCount) NCCLCHK(r.CommCount(c->comm, &n));
            if (r.GetVersion) NCCLCHK(r.GetVersion(&ver));
            out[1] = n; out[2] = ver;
            HIPCHK(hipMemcpyAsync(d, v.data(), (size_t)(W + 1) * 8, hipMemcpyHostToDevice, c->comm_stream));
            NCCLCHK(r.AllReduce(d, d, 1, ncclFloat64, ncclSum, c->comm, c->comm_stream));
            NCCLCHK(r.AllReduce(d + 1, d + 1, W, ncclFloat64, ncclMax, c->comm, c->comm_stream));
            HIPCHK(hipMemcpyAsync(v.data(), d, (size_t)(W + 1) * 8, hipMemcpyDeviceToHost, c->comm_stream));
            HIPCHK(hipStreamSynchronize(c->comm_stream));
        }
        out[3] = (int64_t)(v[0] + 0.5);
        for (int r = 0; r < W; r++) out[4 + r] = (int64_t)v[1 + r] - 1;
    });
}

zvx_status zvx_comm_barrier(zvx_ctx* c) {
    double one = 1.0;
    return zvx_comm_max_f64(c, &one);
}

void zvx_comm_destroy(zvx_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    if (c->comm) { (void)rccl().CommDestroy(c->comm); c->comm = nullptr; }
    for (auto& kv : c->gather_fence) if (kv.second) (void)hipEventDestroy(kv.second);
    c->gather_fence.clear();
    if (c->ev_compute) { (void)hipEventDestroy(c->ev_compute); c->ev_compute = nullptr; }
    if (c->copy_is_comm) { c->copy_stream = nullptr; c->copy_is_comm = false; }      // (host deliveries rode this stream: the next one creates its own)
    if (c->comm_stream) { (void)hipStreamDestroy(c->comm_stream); c->comm_stream = nullptr; }
    c->world = 1; c->rank = 0;
}

zvx_status zvx_dev_alloc(zvx_ctx* c, size_t bytes, void** out) {
    return guarded(c, [&] {
        if (!out || !bytes) fail(ZVX_E_INVALID, "zvx_dev_alloc: bad arguments");
        HIPCHK(hipMalloc(out, bytes));
        HIPCHK(hipMemsetAsync(*out, 0, bytes, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    });
}

zvx_status zvx_dev_free(zvx_ctx* c, void* p) {
    return guarded(c, [&] {
        c->sync();
        auto f = c->gather_fence.find(p);
        if (f != c->gather_fence.end()) { if (f->second) (void)hipEventDestroy(f->second); c->gather_fence.erase(f); }
        if (p) HIPCHK(hipFree(p));
    });
}

zvx_status zvx_dev_from_host(zvx_ctx* c, void* dst, const void* src, size_t bytes) {
    return guarded(c, [&] {
        if (!dst || !src) fail(ZVX_E_INVALID, "zvx_dev_from_host: NULL pointer");
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    });
}

zvx_status zvx_dev_to_host(zvx_ctx* c, void* dst, const void* src, size_t bytes) {
    return guarded(c, [&] {
        if (!dst || !src) fail(ZVX_E_INVALID, "zvx_dev_to_host: NULL pointer");
        c->sync();
        HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    });
}

zvx_status zvx_stage_times(zvx_ctx* c, float ms[ZVX_T_COUNT]) {
    return guarded(c, [&] { c->sync(); for (int s = 0; s < ZVX_T_COUNT; s++) ms[s] = c->stage_ms[s]; });
}

int zvx_kernel_stats(zvx_ctx* c, zvx_kernel_stat* out, int max_out) {
    if (!c) return 0;
    int n = 0;
    guarded(c, [&] {
        c->sync();
        for (auto& s : c->stats) if (s.launches > 0 && n < max_out) out[n++] = s;
    });
    return n;
}

int zvx_tag_stats(zvx_ctx* c, zvx_kernel_stat* out, int max_out) {
    if (!c) return 0;
    int n = 0;
    guarded(c, [&] {
        c->sync();
        for (auto& kv : c->tagstats) if (kv.second.launches > 0 && n < max_out) out[n++] = kv.second;
    });
    return n;
}

zvx_status zvx_reset_stats(zvx_ctx* c) {
    return guarded(c, [&] { c->sync(); for (auto& s : c->stats) { s.launches = 0; s.ms = s.flops = s.bytes = 0; } c->tagstats.clear(); });
}

}  // extern "C"
