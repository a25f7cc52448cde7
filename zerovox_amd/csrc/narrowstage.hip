// narrowstage.hip -- a WHOLE HiFi-GAN stage of narrow width (C = 16 / 8: the last two stages of HiFi-GAN V2, the reference's default
// vocoder, model.py:84) in ONE launch:  hifigan.py:116-125
//
//     out = lrelu( ( rb_k0(x) + rb_k1(x) + rb_k2(x) ) / nk , slope )          rb_k = three times  x += conv_1(lrelu(conv_d(lrelu(x))))
//
// Rounds 1-4 ran these stages as 9 launches each (one per ResBlock pair on resfuse_persist): the 117 MB stage tensor crossed HBM ~20
// times per stage and the 32 x 32 x 16 matrix instruction worked at 1/2 (C = 16) resp. 1/8 (C = 8) of its width -- 0.03-0.09 of the
// matrix roof, 0.18-0.29 of HBM, 2.7 ms of the V2 step.  Here the stage tensor is read once and written once:
//   * a persistent workgroup walks tiles of R rows of one utterance; the tile (+ 64 halo rows either side: 3 h + (d0 + d1 + d2) h = 60
//     for k = 11, dilations 1 / 3 / 5) and every intermediate of the 18 convolutions live in LDS (three streams: x0 / x2 | T | x1; x0 is re-read from L2 per ResBlock), the running sum over
//     the ResBlocks in f32 REGISTERS (rounds 1-4 kept it in a 16-bit tensor in HBM: two roundings fewer per output);
//   * matrix shape 16 x 16 x 32 with the taps STACKED into the contraction: K = 32 = 2 taps x 16 channels (C = 16) resp. 4 taps x 8
//     channels (C = 8), M = output channels, N = 16 time rows.  A lane's B operand is 16 contiguous bytes of one LDS row (row n +
//     tap offset, half-row kb % 2 resp. the whole 8-channel row), its four accumulators are 4 consecutive channels of one row: 8-byte
//     LDS / HBM stores in the time-major layout.  k = 3 / 7 / 11 take 2 / 4 / 6 matrix steps per 16 rows at C = 16 (1 / 2 / 3 at C = 8);
//   * all weight fragments of the stage (72 KiB at C = 16, 36 KiB at C = 8) and the biases sit in LDS for the workgroup's lifetime;
//   * every convolution computes exactly the rows later convolutions need (the halo shrinks along the chain), in 16-row blocks dealt
//     round-robin to the waves; one barrier per convolution.
// Rows outside the utterance are zeros in every stream (each convolution zero-pads ITS input, hifigan.py:39-44), so an utterance's
// result does not depend on the batch or on where tile boundaries fall: streamed and whole-utterance vocoding stay bit-identical.
#include "mfma_util.h"
#include "zvx_kernels.h"

#include <hip/hip_ext.h>

#include <type_traits>

namespace zvx {

static thread_local hipEvent_t g_ns_ev_start = nullptr, g_ns_ev_stop = nullptr;
void narrowstage_profile_events(hipEvent_t start, hipEvent_t stop) { g_ns_ev_start = start; g_ns_ev_stop = stop; }

#define NS_HB 64          // halo rows kept either side of a tile (a multiple of the 16-row block)
#define NS_GUARD 16       // rows in front of / behind the halo that block-rounded convolutions may touch (never consumed)

template <int I, int N, class F>
__device__ __forceinline__ void ns_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); ns_static_for<I + 1, N>(f); }
}
template <bool H>
__device__ __forceinline__ f32x4 ns_mfma(const uint4& a, const uint4& b, const f32x4& c) {
    if constexpr (H) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// fragment order of one convolution's weights for this kernel: step s, lane (m = lane % 16, kb = lane / 16) holds
//   C = 16: W[tap 2 s + kb / 2][out m][in (kb % 2) 8 .. + 8]        C = 8: W[tap 4 s + kb][out m][in 0 .. 8]   (zeros past the last tap / channel)
__global__ void k_pack_narrow(const unsigned short* w /*[k][C][C]*/, int k, int C, uint4* out, int nsteps) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nsteps * 64) return;
    const int lane = idx & 63, s = idx >> 6, m = lane & 15, kb = lane >> 4;
    const int tap = C == 16 ? 2 * s + (kb >> 1) : 4 * s + kb, ch0 = C == 16 ? (kb & 1) * 8 : 0;
    unsigned short v[8];
    for (int e = 0; e < 8; e++) v[e] = (tap < k && m < C && ch0 + e < C) ? w[((long)tap * C + m) * C + ch0 + e] : (unsigned short)0;
    out[idx] = make_uint4(v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16), v[6] | ((unsigned)v[7] << 16));
}
int narrowstage_steps(int C, int k) { const int tpm = 32 / C; return (k + tpm - 1) / tpm; }
void launch_pack_narrow(const void* w16, int k, int C, void* out, hipStream_t s) {
    const int n = narrowstage_steps(C, k) * 64;
    hipLaunchKernelGGL(k_pack_narrow, dim3((n + 255) / 256), dim3(256), 0, s, (const unsigned short*)w16, k, C, (uint4*)out, narrowstage_steps(C, k));
}

template <int C, int R, int NW, bool H16>
struct NsKernel {
    static constexpr int P = C == 16 ? 48 : 16;                        // LDS row pitch: 32 B + 16 B pad (16 rows x 12 dwords cover the 64 banks once) / 16 B
    static constexpr int TPM = 32 / C;                                 // taps per matrix step
    static constexpr int ROWS = R + 2 * NS_HB + 2 * NS_GUARD;          // rows of an LDS stream buffer
    static constexpr int NBT = R / 16;                                 // output blocks of a tile
    static constexpr int NBW = (NBT + NW - 1) / NW;                    // ... per wave
    static constexpr int QN = C == 16 ? 4 : 2;                         // 4-channel groups that exist
    static constexpr bool PAIRS = C == 16;                             // two 16-row blocks per loop iteration (two accumulation chains in flight); C = 8 runs 4 waves per SIMD on 128 registers instead

    const StageArgs& a;
    unsigned char* const A; unsigned char* const T; unsigned char* const B;    // LDS streams: x0 / x2 | T | x1
    const uint4* const wl; const float* const bias_l;
    const int lane, wave, n, grp, kb;
    int m0, len;
    f32x4 sum[NBW];

    __device__ __forceinline__ NsKernel(const StageArgs& a_, unsigned char* lds, int wbytes)
        : a(a_), A(lds), T(lds + ROWS * P), B(lds + 2 * ROWS * P), wl((const uint4*)(lds + 3 * ROWS * P)),
          bias_l((const float*)(lds + 3 * ROWS * P + wbytes)), lane(threadIdx.x & 63), wave(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)),
          n(lane & 15), grp(lane >> 4), kb(lane >> 4), m0(0), len(0) {}

    // KIND 0: conv1 (dilated; T = lrelu(. + b1)), 1: conv2 feeding the next pair (Y = lrelu(. + b2 + x)), 2: a ResBlock's last conv2 (-> sum).
    // DIL is a template constant: tap t of matrix step s sits (TPM s - h) DIL rows from the block row -- an IMMEDIATE of the ds_read -- plus a
    // per-lane part ((kb / 2) resp. kb taps and the half-row) that is folded into the lane's base address once per block: the block loop
    // carries no address arithmetic per matrix step.  The kernel is bound by instruction issue (a SIMD issues ~1 instruction per 4-5
    // cycles, whatever its kind), so every instruction taken out of the 16-row block is time: bias in the accumulator's initial value,
    // no utterance-boundary masks on interior tiles (MASKED = false).
    template <int K, int KIND, int DIL, bool MASKED>
    __device__ __forceinline__ void conv(int ci, const unsigned char* src, unsigned char* dst, const unsigned char* res, int E) {
        constexpr int S = (K + TPM - 1) / TPM, h = (K - 1) / 2;
        uint4 wf[S];
        const uint4* const wq = wl + (long)a.woff[ci] * 64 + lane;
#pragma unroll
        for (int s = 0; s < S; s++) wf[s] = wq[s * 64];
        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
        if (grp < QN) { const float4 t = *(const float4*)(bias_l + ci * C + 4 * grp); bq = (f32x4){t.x, t.y, t.z, t.w}; }
        // lane part of the operand address: its tap within the step's TPM taps, its half-row (C = 16); minus the h DIL rows of the first tap.
        // (The last step's slots past tap K - 1 multiply zeros of A; they read up to TPM - 1 taps further right: inside the guard rows.)
        const int lane_off = ((C == 16 ? (kb >> 1) : kb) * DIL - h * DIL) * P + (C == 16 ? (kb & 1) * 16 : 0);
        const float slope1 = a.slope1, rinv = a.res_inv_slope;
        auto blocks = [&](int blk0, auto nbc, auto ic) __attribute__((always_inline)) {
            constexpr int NB = decltype(nbc)::value, i0 = decltype(ic)::value;
            const int row0 = NS_GUARD + blk0 * 16 + n;                   // this lane's row of the LDS streams (first block)
            const unsigned char* const bp = src + row0 * P + lane_off;
            uint4 xf[NB][S];
            uint2 rq[NB];
#pragma unroll
            for (int u = 0; u < NB; u++) {
#pragma unroll
                for (int s = 0; s < S; s++) xf[u][s] = *(const uint4*)(bp + (u * 16 + TPM * s * DIL) * P);
                rq[u] = make_uint2(0, 0);
                if (KIND >= 1 && grp < QN) rq[u] = *(const uint2*)(res + (row0 + u * 16) * P + grp * 8);
            }
            f32x4 acc[NB];
#pragma unroll
            for (int u = 0; u < NB; u++) acc[u] = bq;
#pragma unroll
            for (int s = 0; s < S; s++)
#pragma unroll
                for (int u = 0; u < NB; u++) acc[u] = ns_mfma<H16>(wf[s], xf[u][s], acc[u]);
#pragma unroll
            for (int u = 0; u < NB; u++) {
                f32x2 v01 = (f32x2){acc[u][0], acc[u][1]}, v23 = (f32x2){acc[u][2], acc[u][3]};
                if (KIND >= 1) { v01 += inv_lrelu2(unpack16<H16>(rq[u].x), rinv); v23 += inv_lrelu2(unpack16<H16>(rq[u].y), rinv); }
                if constexpr (KIND == 2) {
                    sum[i0 + u] += (f32x4){v01.x, v01.y, v23.x, v23.y};
                } else {
                    v01 = lrelu2(v01, slope1); v23 = lrelu2(v23, slope1);
                    uint2 pk = make_uint2(pack16<H16>(v01.x, v01.y), pack16<H16>(v23.x, v23.y));
                    if (MASKED) {                                        // streams are zero outside the utterance
                        const int g = m0 - NS_HB + (blk0 + u) * 16 + n;
                        if (g < 0 || g >= len) pk = make_uint2(0u, 0u);
                    }
                    if (grp < QN) *(uint2*)(dst + (row0 + u * 16) * P + grp * 8) = pk;
                }
            }
        };
        constexpr std::integral_constant<int, 1> one{};
        constexpr std::integral_constant<int, 2> two{};
        if constexpr (KIND == 2) {                                       // the tile's own rows: blocks wave NBW + i <-> sum[i] for every ResBlock
            static_assert(NBT == NBW * NW, "the tile's blocks divide evenly over the waves");
            if constexpr (PAIRS) {
                ns_static_for<0, NBW / 2>([&](auto ic) { constexpr int i = 2 * decltype(ic)::value; blocks(NS_HB / 16 + wave * NBW + i, two, std::integral_constant<int, i>{}); });
                if constexpr (NBW & 1) blocks(NS_HB / 16 + wave * NBW + NBW - 1, one, std::integral_constant<int, NBW - 1>{});
            } else {
                ns_static_for<0, NBW>([&](auto ic) { blocks(NS_HB / 16 + wave * NBW + decltype(ic)::value, one, ic); });
            }
        } else {
            // the rows later convolutions need, rounded to blocks, dealt to the waves in contiguous runs of equal length
            const int lo = (NS_HB - E) / 16, hi = (NS_HB + R + E + 15) / 16, per = (hi - lo + NW - 1) / NW;
            const int b0 = lo + wave * per, b1 = min(b0 + per, hi);
            int blk = b0;
            if constexpr (PAIRS) for (; blk + 1 < b1; blk += 2) blocks(blk, two, std::integral_constant<int, 0>{});
            for (; blk < b1; blk++) blocks(blk, one, std::integral_constant<int, 0>{});
        }
        __syncthreads();
    }

    // stage input rows [m0 - HB, m0 + R + HB) -> A (zeros outside the utterance); 16-byte chunks, coalesced.  Every ResBlock starts from
    // it again (A is the chain's x0 and later its x2): re-read per ResBlock from L2 instead of a fourth LDS stream
    __device__ __forceinline__ void load_x(const unsigned short* Xb) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                                  // the per-thread addresses below are recomputed here, not carried (as spills) through the whole tile loop
        constexpr int CPR = C / 8, TOT = (R + 2 * NS_HB) * CPR;
        for (int i0 = 0; i0 < TOT; i0 += 64 * NW * 4) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + tid + u * 64 * NW, r = i / CPR, q = i - r * CPR, g = m0 - NS_HB + r;
                v[u] = make_uint4(0, 0, 0, 0);
                if (i < TOT && g >= 0 && g < len) v[u] = *(const uint4*)(Xb + (long)g * a.ldx + q * 8);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + tid + u * 64 * NW, r = i / CPR, q = i - r * CPR;
                if (i < TOT) *(uint4*)(A + (NS_GUARD + r) * P + q * 16) = v[u];
            }
        }
        __syncthreads();
    }

    // one ResBlock1 with the dilations (D0, D1, D2) (hifigan.py:49-56); the halo each convolution's output still needs shrinks along the chain
    template <int K, int D0, int D1, int D2, bool MASKED>
    __device__ __forceinline__ void resblock(int j, const unsigned short* Xb) {
        constexpr int h = (K - 1) / 2, ci0 = 0;
        const int ci = 6 * j + ci0;
        constexpr int E4 = h, E3 = h + D2 * h, E2 = 2 * h + D2 * h, E1 = 2 * h + (D1 + D2) * h, E0 = 3 * h + (D1 + D2) * h;
        load_x(Xb);
        conv<K, 0, D0, MASKED>(ci + 0, A, T, nullptr, E0);
        conv<K, 1, 1, MASKED>(ci + 1, T, B, A, E1);
        conv<K, 0, D1, MASKED>(ci + 2, B, T, nullptr, E2);
        conv<K, 1, 1, MASKED>(ci + 3, T, A, B, E3);
        conv<K, 0, D2, MASKED>(ci + 4, A, T, nullptr, E4);
        conv<K, 2, 1, MASKED>(ci + 5, T, nullptr, A, 0);
    }
    template <bool MASKED>
    __device__ __forceinline__ void resblocks(const unsigned short* Xb) {
        for (int j = 0; j < a.nk; j++) {
            const int k = a.ks[j];
            if (k == 3) resblock<3, 1, 3, 5, MASKED>(j, Xb);
            else if (k == 7) resblock<7, 1, 3, 5, MASKED>(j, Xb);
            else if (k == 11) resblock<11, 1, 3, 5, MASKED>(j, Xb);
            else resblock<5, 1, 3, 5, MASKED>(j, Xb);
        }
    }

    __device__ __forceinline__ void tile(int b, int mt) {
        m0 = mt * R;
        const unsigned short* const Xb = (const unsigned short*)a.X + (long)b * a.x_bs;
#pragma unroll
        for (int i = 0; i < NBW; i++) sum[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // interior tiles (every stream row of the tile lies inside the utterance) carry no boundary masks
        if (m0 - NS_HB - NS_GUARD >= 0 && m0 + R + NS_HB + NS_GUARD <= len) resblocks<false>(Xb);
        else resblocks<true>(Xb);
        // out = lrelu(sum / nk, slope): 8 bytes per lane (4 channels of one row); a wave's store covers 16 consecutive rows
        const float inv = 1.0f / (float)a.nk, oslope = a.slope;
        unsigned short* const Ob = (unsigned short*)a.out + (long)b * a.o_bs;
#pragma unroll
        for (int i = 0; i < NBW; i++) {
            const int blk = wave * NBW + i, g = m0 + blk * 16 + n;
            if (blk < NBT && g < len && grp < QN) {
                const f32x2 v01 = lrelu2((f32x2){sum[i][0], sum[i][1]} * inv, oslope), v23 = lrelu2((f32x2){sum[i][2], sum[i][3]} * inv, oslope);
                *(uint2*)(Ob + (long)g * a.ldo + grp * 4) = make_uint2(pack16<H16>(v01.x, v01.y), pack16<H16>(v23.x, v23.y));
            }
        }
    }
};

template <int C, int R, int NW, int WGPC, bool H16>
__global__ __launch_bounds__(64 * NW, NW * WGPC / 4) void narrowstage_kernel(const StageArgs a, int wbytes, int ntm, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    using KN = NsKernel<C, R, NW, H16>;
    if (H16) f16_saturate_mode();
    KN kn(a, lds, wbytes);
    const int tid = threadIdx.x;
    // the stage's weight fragments and biases: once per workgroup
    {
        uint4* const wd = (uint4*)(lds + 3 * KN::ROWS * KN::P);
        const uint4* const ws = (const uint4*)a.W;
        for (int i = tid; i < wbytes / 16; i += 64 * NW) wd[i] = ws[i];
        float* const bd = (float*)(lds + 3 * KN::ROWS * KN::P + wbytes);
        for (int i = tid; i < 6 * a.nk * C; i += 64 * NW) bd[i] = a.bias[i];
    }
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = t / ntm, mt = t - b * ntm;
        int len = a.len ? a.len[b] : a.M;
        len = __builtin_amdgcn_readfirstlane(len);
        if (mt * R >= len) continue;
        // Every tile starts from zeroed streams.  Block-rounded convolutions read a few rows nothing has written in this tile,
        // and the padded tap slots of a convolution's last matrix step read up to (slots - 1 - h) dil rows past what this tile's earlier
        // convolutions wrote -- rows left over from the workgroup's previous tile, which may belong to ANOTHER utterance.  Those products
        // have zero weights, but 0 x NaN is NaN: a non-finite mel of that utterance must not reach this one (include/zvx.h: every other
        // utterance of the batch stays bit for bit).  4-10 LDS stores per thread and tile; unconditional on purpose (no state carried
        // across the tile loop: the C = 8 form sits at its 128-register budget).  The barrier also orders the weight / bias fill above.
        {
            int z = threadIdx.x;
            unsigned z0 = 0, z1 = 0, z2 = 0, z3 = 0;
            asm volatile("" : "+v"(z), "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3));   // materialised HERE: hoisted out of the tile loop these five registers cost the C = 8 form 9 spills
#pragma nounroll
            for (int i = z; i < 3 * KN::ROWS * KN::P / 16; i += 64 * NW) ((uint4*)lds)[i] = make_uint4(z0, z1, z2, z3);
            __syncthreads();
        }
        kn.len = len;
        kn.tile(b, mt);
        __syncthreads();
    }
}

template <int C, int R, int NW, int WGPC>
static bool launch_ns(const StageArgs& a, hipStream_t stream, bool dry_run) {
    using KN = NsKernel<C, R, NW, false>;
    int wfrags = 0;
    for (int j = 0; j < a.nk; j++) wfrags += 6 * narrowstage_steps(C, a.ks[j]);
    const int wbytes = wfrags * 1024;
    const size_t lds = (size_t)3 * KN::ROWS * KN::P + wbytes + (size_t)6 * a.nk * C * 4;
    if (lds > 160 * 1024) return false;
    // every convolution's block-rounded row range, plus the rows its taps reach (incl. the padded tap slots of the last matrix step), must
    // stay inside the stream buffer (halo + guard rows), and what the tile needs inside the halo
    for (int j = 0; j < a.nk; j++) {
        const int k = a.ks[j], h = (k - 1) / 2, slots = narrowstage_steps(C, k) * (32 / C);
        const int d[3] = {a.dil[j][0], a.dil[j][1], a.dil[j][2]};
        const int E1[3] = {3 * h + (d[1] + d[2]) * h, 2 * h + d[2] * h, h};                        // conv1 outputs (dilated taps)
        const int E2[3] = {2 * h + (d[1] + d[2]) * h, h + d[2] * h, 0};                            // conv2 outputs (dilation 1)
        for (int t = 0; t < 3; t++)
            for (int q = 0; q < 2; q++) {
                const int E = q ? E2[t] : E1[t], dl = q ? 1 : d[t];
                if (E + h * dl > NS_HB) return false;
                const int lo = (NS_HB - E) / 16, hi = (NS_HB + R + E + 15) / 16;
                if (NS_GUARD + lo * 16 - h * dl < 0 || NS_GUARD + (hi - 1) * 16 + 15 + (slots - 1 - h) * dl > KN::ROWS - 1) return false;
            }
    }
    if (dry_run) return true;
    const int ntm = (a.M + R - 1) / R, ntiles = ntm * a.nbatch;
    const int ncu = persistent_cus() * WGPC;                                       // WGPC workgroups per CU where the LDS footprint allows (one's loads under the other's matrix steps)
    const dim3 grid(ntiles < ncu ? ntiles : ncu), block(64 * NW);
#define NS_GO(H_) do { auto kfn = narrowstage_kernel<C, R, NW, WGPC, H_>; \
        if (!lds_opt_in((const void*)kfn)) return false; \
        if (g_ns_ev_start) hipExtLaunchKernelGGL(kfn, grid, block, lds, stream, g_ns_ev_start, g_ns_ev_stop, 0, a, wbytes, ntm, ntiles); \
        else hipLaunchKernelGGL(kfn, grid, block, lds, stream, a, wbytes, ntm, ntiles); } while (0)
    if (a.f16) NS_GO(true); else NS_GO(false);
#undef NS_GO
    return true;
}

// true when the stage is covered (and, unless dry_run, launched): C = 16 / 8, ResBlock1 with three pairs per block, kernel sizes in
// {3, 5, 7, 11}, halo 3 h + (d0 + d1 + d2) h <= 64 rows, dense rows
bool launch_narrowstage(const StageArgs& a, hipStream_t stream, bool dry_run) {
    if ((a.C != 16 && a.C != 8) || a.nk < 1 || a.nk > 3 || a.ldx != a.C || a.ldo != a.C || !a.W || !a.bias || !a.X || !a.out) return false;
    for (int j = 0; j < a.nk; j++) {
        const int k = a.ks[j];
        if (!(k == 3 || k == 5 || k == 7 || k == 11) || a.dil[j][0] != 1 || a.dil[j][1] != 3 || a.dil[j][2] != 5) return false;   // the dilations are template constants (HiFi-GAN's ResBlock1 set)
    }
    if (a.C == 16) return launch_ns<16, 384, 8, 1>(a, stream, dry_run);            // 3 x 544 rows x 48 B + 72 KiB of weights = 150 KiB: one workgroup of 8 waves per CU
    return launch_ns<8, 512, 8, 2>(a, stream, dry_run);                               // 3 x 672 x 16 B + 36 KiB = 68 KiB: two workgroups of 8 waves per CU (4 waves per SIMD)
}

}  // namespace zvx
