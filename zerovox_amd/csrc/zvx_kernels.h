// zvx_kernels.h -- launcher declarations for the gfx950 kernels of libzvx.
// Activations are time-major [row = time][channel] in HBM, fp32 or bf16 (precision mode).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <set>
#include <utility>

namespace zvx {

typedef unsigned short bf16_t;   // raw bf16 bits

enum DType { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };   // DT_F16: IEEE half activations + weights of the StyleTTS decoder (11-bit significand: 8x
                                                        // smaller rounding error than bf16 at the same MFMA rate; every value there is O(1) behind a norm)
static inline size_t dtype_size(int dt) { return dt == DT_F32 ? 4 : 2; }
// compute units of the current device, read once (a function-local static: initialisation is thread-safe -- several contexts may be
// driven from several host threads)
inline int num_cus() {
    static const int n = [] { int dev = 0, v = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256; return v; }();
    return n;
}
// CUs the PERSISTENT kernels size their grids for (round 5's spatial split of the chip between the two streams -- ZVX_CU_SPLIT, a
// process-wide override -- measured as a dead end and is gone: DESIGN.md section 4)
inline int persistent_cus() { return num_cus(); }

// Opt-in to more than 64 KiB of dynamic LDS for a kernel: hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of the
// function, so it is set once per (function, device) -- a second context on another device of the same process opts in again.
// false when the runtime refuses (the launcher then declines the shape instead of launching into an error).
inline bool lds_opt_in(const void* kfn) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({kfn, dev})) return true;
    if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    done.insert({kfn, dev});
    return true;
}

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_TANH = 3 };

#define ZVX_MAX_TAPS 16

// ------------------------------------------------------------------------------------------------
// Gathered-row GEMM ("conv-GEMM"), the workhorse:
//
//   out[z][r][n] = epilogue( alpha * sum_tap sum_k X[z][rowmap(r, tap)][k] * W[tap][n][k] )
//
// X ("row operand", usually activations) and W ("col operand", usually weights, [tap][N][ldw]) are both
// K-contiguous.  rowmap() implements 1-D convolution taps (dilated Conv1d, polyphase ConvTranspose1d,
// plain Linear with one tap) and 2-D taps with stride (ResNet Conv2d on a [H][W][C] map):
//     r -> (u, v) = (r / wout, r % wout);  in_u = u*stride + du[tap];  in_v = v*stride + dv[tap]
//     valid iff 0 <= in_u < hin and 0 <= in_v < in_len[z];  X row = in_u*win + in_v  (else zero)
// Both operands may be batched (z = batch*nheads + head) with batch and head strides, which also covers
// Q.K^T and P.V of the attention (W := K or V^T of that (utterance, head)).
// ------------------------------------------------------------------------------------------------
struct GemmArgs {
    // operands
    const void* X; long x_bs, x_hs; int ldx;
    const void* W; long w_bs, w_hs, w_ts; int ldw;
    const void* Wp;            // optional: W pre-packed in MFMA-fragment order (launch_pack_weights) -> conv-slab kernel
    int halo_l, halo_r;        // filled by the launcher
    // fused ResBlock1 pair (resfuse kernel): out = epilogue(conv2(lrelu(conv1(X) + bias1)) + bias + inverse_lrelu(X))
    const void* Wp2; const float* bias1; int dv1[ZVX_MAX_TAPS]; int fused; float slope1;   // conv1: Wp2/bias1/dv1 (dilated); conv2: Wp/bias/dv
    int no_pairstream;         // fused: 1 = never the streaming pair kernel of pairstream.hip (A/B switch), 2 = use it even for small jobs (tests), 3 = ... with 256-row segments
    int dtype;                 // DType of X and W (same)
    int M, N, K;               // M = max rows per z, N cols, K per tap (multiple of 8 elements bf16 / 4 f32)
    int nbatch, nheads;
    const int* in_len;         // [nbatch] valid input width (1-D: valid rows of X); NULL -> in_len_static
    const int* out_len;        // [nbatch] valid output width (1-D: valid rows of out); NULL -> M
    const int* k_len;          // [nbatch] per-batch K (rounded up to 8 internally); NULL -> K
    int in_len_static;
    // row map
    int ntaps; int du[ZVX_MAX_TAPS], dv[ZVX_MAX_TAPS];   // int (not short): uniform-indexed kernarg reads become s_load
    int stride, wout, hin, win; // wout <= 0 -> 1-D (u = 0, v = r)
    // Filled by launch_gemm when a stride-1 2-D convolution over [hin][win] maps is run on the 1-D conv-slab / conv-reg kernels as a
    // convolution over the FLATTENED map (tap offset du * win + dv): input row g is valid iff 0 <= g < flat_rows and
    // (g % flat_win) < in_len[z] (the columns of an utterance's true width); every output row of the map is written
    int flat_win, flat_rows;
    // Batch-flattened 1-D convolution (conv-slab kernel only; a HINT: every other kernel ignores it and runs the per-utterance
    // launch the other fields describe).  The nbatch utterances lie `bflat` rows apart in X, out and res (x_bs = bflat * ldx,
    // o_bs = bflat * ldo, r_bs = bflat * ldr) with bflat >= M + max |tap offset|, so the batch is ONE row axis of nbatch * bflat
    // rows: row tiles run across utterance boundaries (no half-empty last tile per utterance: 896 frames are 3.5 tiles of 256),
    // rows g with (g % bflat) >= in_len[g / bflat] are staged as zeros.  EVERY row of the flattened axis is written (rows past an
    // utterance's length receive finite-or-not junk): every consumer of such a tensor masks by the utterance lengths on the way in.
    int bflat;
    // Second source (conv-slab kernel only): out += X2 . W2^T, a 1-tap product over K2 more input channels accumulated into the SAME
    // f32 accumulators after the K chunks of X -- a residual block's 1x1 shortcut convolution inside its last k = 3 convolution
    // (styletts.py:60-68: (conv2(.) + conv1x1(x)) / sqrt(2)).  X2 rows map like X rows (same batch stride rule under bflat); Wp
    // is then the COMBINED fragment stream: per 32-channel tile the fragments of W, then those of W2 (pack_weights_pair).
    const void* X2; long x2_bs; int ldx2, K2;
    int xcd_flat;                                                  // conv-slab: remap over the WHOLE grid (batch x tiles), not per utterance (zvx_set_int "slab_flat"; the context's switch)
    int slab_small;                                                // conv-slab tile choice for single requests (zvx_set_int "slab_small"; the context's switch): 0 none, 1 small row tiles, 2 + 32-channel tiles
    int out_split3;            // f32 result written as 16-bit split planes [hi | hi | lo] (row = 3 N, ldo elements apart): the input of the next 3-plane GEMM; 1 = bf16, 2 = IEEE half (lo x 2^11)
    // epilogue: v = alpha*acc + bias; v += res; v += accum; [accum = v]; v *= out_scale; v = act(v);
    //           v = v*post_scale[n] + post_shift[n]; out = (T)v
    float alpha;
    const float* bias; int bias_mode;            // 0 none, 1 per n, 2 per row r
    const void* res; long r_bs, r_hs; int ldr; int res_dtype; int res_mode;  // 0 none, 1 raw, 2 inverse leaky-relu
    float res_inv_slope;                         // res_mode 2: x = y >= 0 ? y : y * res_inv_slope
    void* accum; long a_bs; int lda; int accum_mode;   // bit0: v += accum, bit1: accum = v (after add)
    int accum_dtype;                             // DT_F32 / DT_BF16 storage of the running sum
    float out_scale;
    int act; float slope;
    const float* post_scale; const float* post_shift;
    void* out; long o_bs, o_hs; int ldo; int out_dtype;   // out may be NULL (accum only)
    // squeeze-excite pool fused into the persistent 2-D convolution (conv2d_persist_kernel, MODE 0): partial sums over the valid
    // positions of the f32 results before the bias, [nbatch][S][N]; the launcher writes S to *se_part_S (a HOST int the caller
    // zeroed) when the launch it chose fills them -- otherwise the caller runs the pool pass (launch_se_pool)
    float* se_part; int* se_part_S;
    // level transition of the speaker encoder in one launch (conv2d_s2_kernel): besides the 3 x 3 / stride-2 convolution these
    // arguments describe, the block's 1 x 1 / stride-2 shortcut convolution of the SAME input: packed weights [1][N][K], bias [N],
    // 16-bit output laid out like `out`.  launch_gemm returns -7 when the shape is not covered (probe with gemm_variant_of).
    void* ds_out; const void* ds_Wp; const float* ds_bias;
    // accounting
    double flops;              // algorithmic FLOPs of this launch (filled by the launcher)
};

// Returns the kernel-variant id used (index into gemm_variant_name) or <0 on error.
int launch_gemm(const GemmArgs& a, hipStream_t stream);
// arm (or disarm with nullptrs) a pair of events that the NEXT launch_gemm / launch_resfuse dispatch carries as its own
// start / stop timestamps (no marker packets on the stream)
void gemm_profile_events(hipEvent_t start, hipEvent_t stop);
// the kernel variant launch_gemm / launch_resfuse would pick for these arguments (nothing is dispatched)
int gemm_variant_of(const GemmArgs& a);
// fused HiFi-GAN ResBlock1 pair (conv1 -> lrelu -> conv2 -> + x) for C = 32 / 64 bf16; -1 if the shape is not covered
int launch_resfuse(GemmArgs a, hipStream_t stream);
// fused HiFi-GAN ResBlock2 (two dilated convolutions with their residuals, hifigan.py:77-82) for C = 32 / 64, k = 3 / 5 / 7; -1 if not covered
int launch_rb2fuse(GemmArgs a, hipStream_t stream);
// fragment-order packing of a bf16 weight [ntaps][N][K] for the conv-slab kernel
size_t packed_weight_elems(int ntaps, int N, int K);
// combined stream of two packed weights with the same N (GemmArgs::X2): per 32-channel tile the fragments of A, then those of B
void launch_pack_pair(const void* packedA, int ntapsA, int KA, const void* packedB, int ntapsB, int KB, int N, void* out, hipStream_t s);
void launch_pack_weights(const void* w_bf16, int ntaps, int N, int K, void* out, hipStream_t s);
const char* gemm_variant_name(int id);
int gemm_num_variants();

// ------------------------------------------------------------------------------------------------
// Streaming ResBlock1 chain (resstream.hip): up to three (dilated conv -> lrelu -> conv -> + x) pairs of one HiFi-GAN
// ResBlock (hifigan.py:49-56) in ONE launch for C = 32 / 64.  Every intermediate tensor lives in LDS ring buffers; the
// stage tensor is read once and the result (next input, running sum xs, or the stage mean) written once.
// ------------------------------------------------------------------------------------------------
struct StreamArgs {
    const void* X; long x_bs; int ldx;             // stage input [b][M][ldx] bf16, activated domain (lrelu(x))
    const void* W1[3]; const void* W2[3];          // fragment-packed bf16 weights of pair t: conv1 (dilated), conv2 (dilation 1)
    const float* b1[3]; const float* b2[3];
    int dil[3];
    int C, ntaps, npair;
    void* out; long o_bs; int ldo;                 // bf16 output, activated with `slope` (NULL: running sum only)
    void* accum; long a_bs; int lda; int accum_mode;   // bf16 running sum xs: bit0 v += xs, bit1 xs = v
    float slope1, res_inv_slope, out_scale, slope; // slope1: lrelu between the convs; slope: output activation (1 = none)
    const int* len; int M, nbatch;                 // valid rows per utterance (NULL -> M)
    int S, nseg;                                   // filled by the launcher: rows per segment, segments per utterance
    int dX0, dT, dX[3];                            // filled by the launcher: ring sizes in rows
    double flops;                                  // filled by the launcher
    long long* prof;                               // RS_PROFILE builds: per-wave cycle counters of workgroup 0
    int seg_min;                                   // 0: segments as short as 256 rows when the job cannot fill the chip; < 0: never below 2048 (A/B)
    int f16;                                       // the 16-bit tensors (x, rings, weights, xs, output) are IEEE half instead of bf16
    int opt;                                       // bit 0: staggered wave priorities, bit 1: balanced role -> SIMD table (zvx_set_int("rs_opt", v): A/B switch)
};
// variant id (index into gemm_variant_name) or -1 when the shape is not covered; dry_run: decide only, launch nothing
int launch_resstream(StreamArgs a, hipStream_t stream, bool dry_run);
void resstream_profile_events(hipEvent_t start, hipEvent_t stop);     // like gemm_profile_events, for the next launch_resstream

// ------------------------------------------------------------------------------------------------
// Streaming ResBlock1 PAIR for the wide stages (pairstream.hip): x' = conv2(lrelu(conv1_dilated(x_act) + b1)) + b2 + x in ONE
// launch for C = 128 (hifigan.py:51-55).  conv1 and conv2 are two wave groups of a persistent workgroup that walk a segment
// of one utterance in 128-row steps; T = lrelu(conv1) lives only in an LDS ring, weights stream through per-wave register rings.
// ------------------------------------------------------------------------------------------------
struct PairArgs {
    const void* X; long x_bs; int ldx;             // stage input [b][M][ldx] bf16, activated domain; also the residual
    const void* W1; const void* W2;                // fragment-packed bf16 weights: conv1 (dilated), conv2 (dilation 1)
    const float* b1; const float* b2;
    int C, ntaps, dil;
    void* out; long o_bs; int ldo;                 // bf16 output activated with `slope` (NULL: running sum only)
    void* accum; long a_bs; int lda; int accum_mode;   // bf16 running sum xs: bit0 v += xs, bit1 xs = v
    float slope1, res_inv_slope, out_scale, slope; // slope 1 = no output activation
    const int* len; int M, nbatch;
    int force;                                     // 1: also for jobs below the size where the kernel pays (tests); 2: and with 256-row segments
    int f16;                                       // the 16-bit tensors (x, T, weights, xs, output) are IEEE half instead of bf16
    int S, nseg, DX, DT, G0;                       // filled by the launcher: segment rows, segments per utterance, ring rows
    long long* prof;                               // PS_PROFILE builds: [8 waves][main, epilogue, barrier, -] cycle totals of workgroup 0
};
// true when the shape is covered (and, unless dry_run, launched); ev_start / ev_stop: optional dispatch-carried events
bool launch_pairstream(PairArgs a, hipStream_t stream, bool dry_run, hipEvent_t ev_start, hipEvent_t ev_stop);

// ------------------------------------------------------------------------------------------------
// A whole narrow HiFi-GAN stage (narrowstage.hip): the nk ResBlock1 blocks (three dilated pairs each) of a stage with C = 16 / 8
// channels, their sum / nk and the next stage's input activation in ONE launch; every intermediate lives in LDS, the stage tensor
// crosses HBM once in and once out (hifigan.py:116-125; HiFi-GAN V2's last two stages).
// ------------------------------------------------------------------------------------------------
struct StageArgs {
    const void* X; long x_bs; int ldx;             // stage input [b][M][C] 16-bit, activated domain (lrelu(x, 0.1))
    const void* W;                                 // weight fragments in this kernel's own order (launch_pack_narrow): conv 6 j + 2 t + {0: conv1, 1: conv2}
    int woff[18];                                  // first 1-KiB fragment of each convolution in W
    const float* bias;                             // [6 nk][C], same convolution order
    int C, nk, ks[3], dil[3][3];
    void* out; long o_bs; int ldo;                 // lrelu(mean over the ResBlocks, slope) [b][M][C] 16-bit
    float slope1, res_inv_slope, slope;
    const int* len; int M, nbatch;
    int f16;                                       // the 16-bit tensors are IEEE half instead of bf16
};
bool launch_narrowstage(const StageArgs& a, hipStream_t stream, bool dry_run);
int narrowstage_steps(int C, int k);               // 1-KiB fragments per convolution
void launch_pack_narrow(const void* w16 /*[k][C][C] 16-bit*/, int k, int C, void* out, hipStream_t s);
void narrowstage_profile_events(hipEvent_t start, hipEvent_t stop);

// ------------------------------------------------------------------------------------------------
// Fused attention of the FS2 / SCLN decoder (attention.hip): out = softmax(Q K^T * scale, keys < len) V per (utterance, head),
// bf16 operands, no [L][L] tensor in HBM.  Q and K live in one projection buffer (K at element offset k_off of a row), V
// arrives TRANSPOSED ([head*D + j][key], key-contiguous) from its projection GEMM.
// ------------------------------------------------------------------------------------------------
struct FlashArgs {
    const void* qk; long qk_bs; int ldq; int k_off;     // [b][L][ldq] bf16: Q at column h*D, K at column k_off + h*D
    const void* vt; long vt_bs; int ldv;                 // [b][nheads*D][ldv] bf16 (ldv >= L rounded up to 8)
    void* out; long o_bs; int ldo;                       // [b][L][ldo] bf16, head h at column h*D
    const int* len; int L, D, nheads, nbatch;
    float scale;                                         // 1 / sqrt(D)   (fs2.py:49-50)
    int f16;                                             // the 16-bit tensors are IEEE half instead of bf16
    long long* prof;                                     // FA_PROFILE builds (tools/micro/fa_bench.hip): per-phase cycle totals of workgroup 0, wave 0
};
bool launch_flash_attention(const FlashArgs& a, hipStream_t stream, bool dry_run);
// exact-f32 fused attention of the phoneme encoder: Q | K | V columns of one projection buffer, head h at column off + h*D
struct AttnF32Args {
    const float* qkv; long bs; int ld; int q_off, k_off, v_off;   // [b][L][ld] f32
    float* out; long o_bs; int ldo;                               // [b][L][ldo] f32, head h at column h*D
    unsigned short* planes; int planes_C;                         // optional: the result also as 16-bit split planes [hi | hi | lo], rows of 3 planes_C
    int planes_f16;                                               // planes in IEEE half (lo scaled by 2^11) instead of bf16
    const int* len; int L, D, nheads, nbatch;
    float scale;
};
bool launch_attention_f32(const AttnF32Args& a, hipStream_t stream, bool dry_run);   // timed through flash_profile_events
void flash_profile_events(hipEvent_t start, hipEvent_t stop);

// ------------------------------------------------------------------------------------------------
// Small kernels (ops.hip).  T-typed pointers are void* + dtype.
// ------------------------------------------------------------------------------------------------
// 16-bit [b][rows][ld_in] (first C columns) -> [b][C][ld_out] (rows on the fast axis; ld_out >= rows): V -> V^T for the fused attention
// single requests: InstanceNorm statistics + affine + activation of a 16-bit tensor in ONE launch (bit-identical to the two-kernel path)
void launch_instnorm_fused(const void* x, int x_dt, int ldx, void* y, int y_dt, int ldy, int B, int Lmax, const int* L, int C, float eps,
                           float* mean, float* rstd, const float* gamma, const float* beta, long g_bs, int one_plus, int act, float slope, hipStream_t s);
void launch_transpose16(const void* in, int ld_in, void* out, int ld_out, int B, int rows, int C, hipStream_t s, const int* len = nullptr);   // rows >= len[b] read as zeros
void launch_f32_to_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s);
// x[b][r][c] = 0 for len[b] <= c < cols; es = element size (2 / 4), ld / bs = row / batch strides in elements
void launch_zero_tail_cols(void* x, int es, long ld, long bs, int B, int rows, int cols, const int* len, hipStream_t s);
void launch_cast(const void* in, int in_dt, void* out, int out_dt, size_t n, hipStream_t s);

// f32 -> three 16-bit planes [hi | hi | lo] per row (out [b][rows_max][3C]; rows >= rows[b] -> zeros) and weights
// [nrows][K] -> [nrows][hi | lo | hi]: an f32-class GEMM as one 16-bit GEMM over 3K (see ops.hip).  f16 = 0: bf16 planes;
// f16 = 1: IEEE-half planes, lo scaled by 2^11, weights scaled by `scale` (a power of two; third plane w * scale * 2^-11)
void launch_split3(const float* x, int ldx, void* out, int B, int rows_max, const int* rows, int C, hipStream_t s, int f16 = 0);
void launch_split3_weights(const float* w, void* out, long nrows, int K, hipStream_t s, int f16 = 0, float scale = 1.f);
void launch_absmax(const float* x, size_t n, float* out /* device, 4 bytes */, hipStream_t s);

// encoder front: out[b][t][:] = cat(emb[ph], pemb[pu]) + pe[t]      (fs2.py:372-392)
void launch_embed(const int* phoneme, const int* puncts, const float* emb, int emb_dim, const float* pemb,
                  int pemb_dim, const float* pe, float* out, int B, int Tmax, const int* T, hipStream_t s);

// Row LayerNorm family over C channels, one wave per row.
//   mode 0: torch LayerNorm (biased var, eps in sqrt), gamma/beta [C]
//   mode 1: SCLN (unbiased std, /(sigma+eps)), bg rows of stride bg_bs: beta = bg[b][0:C], gamma = bg[b][C:2C]  (fs2.py:76-90)
// post_add [B][C] (may be NULL) is added after the affine (the style-embedding add, fs2.py:740-741).
void launch_layernorm(const void* x, int x_dt, int ldx, void* y, int y_dt, int ldy, int B, int rows_max,
                      const int* rows, int C, int mode, float eps, const float* gamma, const float* beta,
                      const float* bg, long bg_bs, const float* post_add, hipStream_t s, void* split_planes = nullptr, int planes_f16 = 0);

// scores [z][L][lds] f32 -> P (dtype) [z][L][ldp]: softmax over n < len[b]; zero-fill [len, roundup8(len))
void launch_softmax_rows(const float* scores, int lds, void* P, int p_dt, int ldp, int nbatch, int nheads,
                         int Lmax, const int* len, hipStream_t s);

// out[b][t] = dot(x[b][t][0:C], w) + bias            (VariancePredictor.linear_layer, fs2.py:553-558)
void launch_rowdot(const float* x, int ldx, const float* w, float bias, float* out, int B, int Tmax,
                   const int* T, int C, hipStream_t s);

// idx = clamp(rint(pred*(nb-1)), 0, nb-1); x[b][t][:] += table[idx][:]     (fs2.py:639,649,668,672)
void launch_bucket_embed_add(const float* pred, const float* table, int nbins, float* x, int ldx, int C,
                             int* idx_out, int B, int Tmax, const int* T, hipStream_t s);

// durations: forced (int) or max(rint(exp(logd)-1),0) (fs2.py:678-681) -> dur[b][t], cum[b][t] (inclusive
// prefix sum), mel_len[b]
void launch_durations(const int* forced, const float* logd, int* dur, int* cum, int* mel_len, int B, int Tmax,
                      const int* T, hipStream_t s);

// length regulator (fs2.py:447-455): feats[b][l][:] = x[b][src(l)][:]; optional positional table add
// into a second output of dtype dt (decoder input): dec[b][l][:] = feats + pe[l]
void launch_length_regulate(const float* x, int ldx, const int* cum, const int* T, const int* mel_len,
                            float* feats, int B, int Tmax, int Lmax, int C, hipStream_t s);
// y[b][l][c] = (T)(x[b][l][c] + (pe ? pe[l][c] : 0));  out_rows_max > 0: y's utterances lie that many rows apart (x's: Lmax)
void launch_add_pe_cast(const float* x, const float* pe, void* y, int y_dt, int ldy, int B, int Lmax,
                        const int* L, int C, hipStream_t s, int out_rows_max = 0);

// InstanceNorm statistics over time for x [b][Lmax][ldx] channels [c0, c0+C): mean/rstd [B][C] (biased, eps)
void launch_instnorm_stats(const void* x, int x_dt, int ldx, int B, int Lmax, const int* L, int C, float eps,
                           float* mean, float* rstd, hipStream_t s);
// y = act(((x-mean)*rstd) * g + b) with g = (one_plus ? 1 : 0) + gamma[b*g_bs + c], b = beta[b*g_bs + c]
// (gamma NULL -> plain normalisation); written to y[.. ldy] (may be a column slice of a wider buffer)
void launch_norm_affine_act(const void* x, int x_dt, int ldx, void* y, int y_dt, int ldy, int B, int Lmax,
                            const int* L, int C, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, long g_bs, int one_plus, int act, float slope, hipStream_t s);

// mel [b][Lmax][nm] (dtype) -> vocoder input [b][Pmax][ldv] dtype: rows >= mel_len zero up to P[b]
void launch_mel_pad(const void* mel, int m_dt, int ldm, int Lmax, const int* mel_len, void* v, int v_dt,
                    int ldv, int Pmax, const int* P, int B, int nm, hipStream_t s);
void launch_copy_rows_f32(const void* src, int s_dt, int lds, long s_bs, float* dst, long ldd, long d_bs,
                          int B, int rows_max, const int* rows, int C, hipStream_t s);

// conv_post (C -> 1, k taps) + tanh on the activated final stage [b][Nmax][C]: wav[b][n], n < nlen[b]*hop...
// samples in [out_len*out_mul, Nmax) are written as zeros; pcm16: wav is int16 PCM (x 32760, truncated)
void launch_conv_post_tanh(const void* x, int x_dt, int ldx, long x_bs, const float* w /*[k][C]*/, float bias,
                           int ktaps, int C, void* wav, long wav_bs, int pcm16, int B, int Nmax, const int* in_len,
                           int len_mul, const int* out_len, int out_mul, hipStream_t s);
// half-mode saturation audit: *count += number of elements of x[b][r < rows[b]][0:C] (16-bit, batch stride bs, row stride ld) whose
// magnitude bits are >= 0x7BFF (+-65504 = a clamped store, or Inf / NaN)
void launch_count_sat16(const void* x, long bs, int ld, int B, int rows_max, const int* rows, int C, unsigned long long* count, hipStream_t s);
// x[b][r][0:C] = 0 for rows[b] <= r < rows_max
void launch_zero_tail_rows(float* x, int ldx, int B, int rows_max, const int* rows, int C, hipStream_t s);

// ---- speaker encoder ----
// InstanceNorm1d(80) over time + Conv2d(1->C0, 3x3, pad 1) + ReLU + BN affine -> map [b][F][Wout][C0] (Wout >= Tmax)
void launch_spk_front(const float* mels, int Tmax, const int* lens, int F, const float* mean, const float* rstd,
                      const float* w /*[9][C0]*/, const float* bias, const float* bn_scale, const float* bn_shift,
                      int C0, void* out, int o_dt, int B, int Wout, hipStream_t s);
// SE global average pool, first half: partial[b][s][c] = sum over the s-th of S = se_pool_splits(H, Wmax) row blocks of the
// valid (f, t) positions of map [b][H][Wmax][C]   (C % 8 == 0, C <= 256)
int se_pool_splits(int H, int Wmax);
void launch_se_pool(const void* x, int x_dt, int B, int H, int Wmax, const int* W, int C, float* partial, hipStream_t s);
// second half + MLP: m = sum_s partial / (H * W[b]);  scale = sigmoid(W2 relu(W1 m + b1) + b2) per clip
void launch_se_fc(const float* partial, int S, int H, const int* W, const float* w1, const float* b1, const float* w2, const float* b2, int C,
                  int Cr, float* scale, int B, hipStream_t s, const float* pool_bias = nullptr);
// y = relu(x * scale[b][c] + res)
void launch_se_apply(const void* x, const void* res, void* y, int dt, const float* scale, int B, int H, int Wmax,
                     const int* W, int C, hipStream_t s);
// ASP pooling (ResNetSE34V2.py:197-205): x map [b][F][Wmax][C] viewed as [t][f*C+c]; logits [b][Wmax][F*C] f32
// out [b][2*F*C] = [mu | sg] in (f*C+c) order
// with_std = 0: self-attentive pooling (SAP, ResNetSE34V2.py:199-200): out [b][F*C] = mu only
void launch_asp_pool(const void* x, int x_dt, const float* logits, int B, int F, int Wmax, const int* W, int C,
                     float* out, int with_std, hipStream_t s);
void launch_l2norm_rows(float* x, int B, int C, hipStream_t s);

// ---- log-mel front end (mels.py:357-395) ----
// out[b][i] = wav[b][reflect(i - pad)] for i < n[b] + 2*pad (numpy 'reflect': no edge repeat), 0 beyond
void launch_reflect_pad(const float* wav, long w_bs, const int* n, float* out, long o_bs, int pad, int B, int out_cols, hipStream_t s);
// mag[b][t][f] = sqrt(re^2 + im^2), re = spec[b][t][f], im = spec[b][t][nf + f]; columns [nf, ldm) and rows >= frames[b] -> 0
void launch_stft_mag(const float* spec, int lds_, float* mag, int ldm, int nf, int B, int Tmax, const int* frames, hipStream_t s);
// x[b][t][c] = log(max(x, lo)) for t < frames[b], 0 beyond
void launch_log_clip(float* x, int ldx, int C, float lo, int B, int Tmax, const int* frames, hipStream_t s);

// skinny f32 linear: out[b][n] = bias[n] + dot(x[b], w[n]) for few rows b and long K (K % 4 == 0)
void launch_fc_rows(const float* x, int ldx, const float* w, int ldw, const float* bias, float* out, int ldo, int B, int N, int K, hipStream_t s);

}  // namespace zvx
