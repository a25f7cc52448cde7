// gemm.hip -- gathered-row GEMM ("conv-GEMM") on gfx950 MFMA.  See GemmArgs in zvx_kernels.h.
//
// One kernel covers every contraction of the synthesis path: dilated Conv1d (HiFi-GAN ResBlocks,
// hifigan.py:25-86; FFN conv k=9, fs2.py:175-187; StyleTTS k=3 convs, styletts.py:28-29), polyphase
// ConvTranspose1d (hifigan.py:100-103), Linear (fs2.py:118-128), the attention products (fs2.py:49-56)
// and the ResNet Conv2d (ResNetSE34V2.py:74-76) via 2-D taps.
//
// Mapping to CDNA4: a 256-thread workgroup (4 waves) owns a BM(time) x BN(channel) output tile.  Both
// operands are K-contiguous in HBM and are staged as 64-byte K-slices (32 bf16 / 16 f32) through LDS with
// an 80-byte row pitch (conflict-free ds_read_b128 for 16 distinct rows).  Weights feed MFMA srcA and
// activations srcB, so every lane ends up holding 4 consecutive CHANNELS of one time row per accumulator
// quad: the epilogue reads/writes 8-byte (bf16) / 16-byte (f32) vectors of the time-major tensors.
//   bf16: v_mfma_f32_32x32x16_bf16, fp32 accumulate.    f32: v_mfma_f32_32x32x2_f32 (exact f32 fma chain).
// Global loads of K-step s+1 are issued before the MFMAs of step s and written to the other LDS buffer
// afterwards (one barrier per step).  blockIdx is remapped so each XCD walks a contiguous tile range
// (all channel tiles of a time tile share that XCD's L2).
#include "zvx_kernels.h"

namespace zvx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // native vector: usable as a tied ("+v") inline-asm operand

#define PITCH 80      // bytes per LDS row: 64 B of K + 16 B pad
#define KBYTES 64

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);   // round-to-nearest-even (finite inputs)
    return (unsigned short)(u >> 16);
}

template <int DT>
__device__ __forceinline__ void load4(const void* base, long idx, float v[4]) {
    if (DT == DT_F32) {
        float4 t = *(const float4*)((const float*)base + idx);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        uint2 t = *(const uint2*)((const unsigned short*)base + idx);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    }
}
__device__ __forceinline__ void load4_dyn(const void* base, int dt, long idx, float v[4]) {
    if (dt == DT_F32) load4<DT_F32>(base, idx, v); else load4<DT_BF16>(base, idx, v);
}
__device__ __forceinline__ void store4_dyn(void* base, int dt, long idx, const float v[4]) {
    if (dt == DT_F32) {
        *(float4*)((float*)base + idx) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        uint2 t;
        t.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        t.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        *(uint2*)((unsigned short*)base + idx) = t;
    }
}

// ---- shared epilogue: lane holds, per 32x32 tile and accumulator quad g, 4 consecutive channels of one time row ----
template <int TM, int TN>
__device__ __forceinline__ void epilogue(const GemmArgs& a, f32x16 (&acc)[TN][TM], int b, int h, int row_base, int col_base,
                                         int out_len, bool two_d, int lane) {
    const long ooff = (long)b * a.o_bs + (long)h * a.o_hs;
    const long roff = (long)b * a.r_bs + (long)h * a.r_hs;
    const long aoff = (long)b * a.a_bs;
#pragma unroll
    for (int j = 0; j < TM; j++) {
        const int r = row_base + j * 32 + (lane & 31);
        bool rok = r < a.M;
        if (two_d) rok = rok && (r % a.wout) < out_len; else rok = rok && r < out_len;
        if (!rok) continue;
        const float brow = (a.bias_mode == 2) ? a.bias[r] : 0.f;
#pragma unroll
        for (int i = 0; i < TN; i++) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int n = col_base + i * 32 + 8 * g + 4 * (lane >> 5);
                if (n >= a.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = acc[i][j][4 * g + e] * a.alpha + brow;
                if (a.bias_mode == 1) {
                    const float4 bb = *(const float4*)(a.bias + n);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if (a.res_mode) {
                    float rr[4];
                    load4_dyn(a.res, a.res_dtype, roff + (long)r * a.ldr + n, rr);
                    if (a.res_mode == 2) {
#pragma unroll
                        for (int e = 0; e < 4; e++) rr[e] = rr[e] >= 0.f ? rr[e] : rr[e] * a.res_inv_slope;
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] += rr[e];
                }
                if (a.accum_mode) {
                    float* ap = a.accum + aoff + (long)r * a.lda + n;
                    if (a.accum_mode & 1) {
                        const float4 t = *(const float4*)ap;
                        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
                    }
                    if (a.accum_mode & 2) *(float4*)ap = make_float4(v[0], v[1], v[2], v[3]);
                }
                if (a.out) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float t = v[e] * a.out_scale;
                        if (a.act == ACT_RELU) t = fmaxf(t, 0.f);
                        else if (a.act == ACT_LRELU) t = t >= 0.f ? t : t * a.slope;
                        v[e] = t;
                    }
                    if (a.post_scale) {
                        const float4 ps = *(const float4*)(a.post_scale + n), pt = *(const float4*)(a.post_shift + n);
                        v[0] = v[0] * ps.x + pt.x; v[1] = v[1] * ps.y + pt.y;
                        v[2] = v[2] * ps.z + pt.z; v[3] = v[3] * ps.w + pt.w;
                    }
                    store4_dyn(a.out, a.out_dtype, ooff + (long)r * a.ldo + n, v);
                }
            }
        }
    }
}

template <int DT, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs a) {
    constexpr int ES = (DT == DT_F32) ? 4 : 2;       // element size
    constexpr int EPC = 16 / ES;                      // elements per 16-byte chunk
    constexpr int BKE = KBYTES / ES;                  // elements per K-step
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int XCH = BM * 4 / 256;                 // X chunks per thread per step
    constexpr int WCH = (BN * 4 + 255) / 256;         // W chunks per thread per step
    static_assert(WM * WN == 4, "4 waves");

    __shared__ __attribute__((aligned(16))) unsigned char lds[2][(BM + BN) * PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave % WM, wc = wave / WM;

    // ---- tile coordinates (XCD-aware, bijective remap) ----
    const int ntn = (a.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int wg;
    {
        const int id = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int nt = wg % ntn, mt = wg / ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    const int z = blockIdx.y, b = z / a.nheads, h = z - b * a.nheads;

    const int out_len = a.out_len ? a.out_len[b] : a.M;
    const int in_len = a.in_len ? a.in_len[b] : a.in_len_static;
    const bool two_d = a.wout > 0;
    if (!two_d && m0 >= out_len) return;
    if (m0 >= a.M) return;
    int Kb = a.K;
    if (a.k_len) { Kb = (a.k_len[b] + 7) & ~7; if (Kb > a.K) Kb = a.K; }

    const unsigned char* Xp = (const unsigned char*)a.X + ((long)b * a.x_bs + (long)h * a.x_hs) * ES;
    const unsigned char* Wp = (const unsigned char*)a.W + ((long)b * a.w_bs + (long)h * a.w_hs) * ES;

    // ---- per-thread staging roles (fixed across K-steps) ----
    int xu[XCH], xv[XCH];
    const int kc = tid & 3;                            // 16-byte chunk within the 64-byte K-slice
#pragma unroll
    for (int i = 0; i < XCH; i++) {
        const int r = m0 + (tid >> 2) + i * 64;
        if (two_d) { xu[i] = (r / a.wout) * a.stride; xv[i] = (r % a.wout) * a.stride; }
        else { xu[i] = 0; xv[i] = r; }
    }

    const int kchunks = (Kb + BKE - 1) / BKE;
    const int S = a.ntaps * kchunks;

    uint4 xreg[XCH], wreg[WCH];

    auto load_regs = [&](int s) {
        const int tap = s / kchunks;
        const int k = (s - tap * kchunks) * BKE + kc * EPC;
        const bool kok = k < Kb;
        const int du = a.du[tap], dv = a.dv[tap];
#pragma unroll
        for (int i = 0; i < XCH; i++) {
            const int iu = xu[i] + du, iv = xv[i] + dv;
            const bool ok = kok && iu >= 0 && iu < a.hin && iv >= 0 && iv < in_len;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) v = *(const uint4*)(Xp + ((long)(iu * a.win + iv) * a.ldx + k) * ES);
            xreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WCH; i++) {
            const int c = tid + i * 256;
            const int n = n0 + (c >> 2);
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((BN * 4 >= 256 * (i + 1) || c < BN * 4) && kok && n < a.N)
                v = *(const uint4*)(Wp + ((long)tap * a.w_ts + (long)n * a.ldw + k) * ES);
            wreg[i] = v;
        }
    };
    auto store_lds = [&](int buf) {
        unsigned char* base = lds[buf];
#pragma unroll
        for (int i = 0; i < XCH; i++)
            *(uint4*)(base + ((tid >> 2) + i * 64) * PITCH + kc * 16) = xreg[i];
#pragma unroll
        for (int i = 0; i < WCH; i++) {
            const int c = tid + i * 256;
            if (BN * 4 >= 256 * (i + 1) || c < BN * 4)
                *(uint4*)(base + (BM + (c >> 2)) * PITCH + kc * 16) = wreg[i];
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int j = 0; j < TM; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    const int xrow = wr * (BM / WM) + (lane & 31);
    const int wrow = BM + wc * (BN / WN) + (lane & 31);
    const int koff = (lane >> 5) * 16;

    if (S > 0) {
        load_regs(0);
        store_lds(0);
    }
    __syncthreads();
    for (int s = 0; s < S; s++) {
        if (s + 1 < S) load_regs(s + 1);
        const unsigned char* base = lds[s & 1];
#pragma unroll
        for (int sub = 0; sub < 2; sub++) {
            uint4 xf[TM], wf[TN];
#pragma unroll
            for (int j = 0; j < TM; j++) xf[j] = *(const uint4*)(base + (xrow + j * 32) * PITCH + sub * 32 + koff);
#pragma unroll
            for (int i = 0; i < TN; i++) wf[i] = *(const uint4*)(base + (wrow + i * 32) * PITCH + sub * 32 + koff);
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int j = 0; j < TM; j++) {
                    if (DT == DT_BF16) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, wf[i]), __builtin_bit_cast(bf16x8, xf[j]), acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wf[i].x), __uint_as_float(xf[j].x), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wf[i].y), __uint_as_float(xf[j].y), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wf[i].z), __uint_as_float(xf[j].z), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wf[i].w), __uint_as_float(xf[j].w), acc[i][j], 0, 0, 0);
                    }
                }
        }
        if (s + 1 < S) store_lds((s + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue ----
    epilogue<TM, TN>(a, acc, b, h, m0 + wr * (BM / WM), n0 + wc * (BN / WN), out_len, two_d, lane);
}

// ---- row-major epilogue: accumulators -> per-wave LDS transpose -> every lane owns 8 consecutive channels of a row ----
// Global traffic of the epilogue (output, residual, f32 accumulator) becomes full-line: a wave instruction touches
// 8 (NW=64) or 16 (NW=32) rows x 128/64 contiguous bytes instead of 32 rows x 16 bytes.  The per-element arithmetic is kept
// to a handful of VALU ops (all mode switches are wave-uniform branches OUTSIDE the element loops; leaky-relu = max(x, s*x),
// its inverse = min(y, y/s); bf16 packing via the hardware RNE convert): with two waves per SIMD the epilogue is otherwise
// VALU-bound (measured 27k cycles per tile for the branchy per-element version vs 7k for the LDS transposes).
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};                 // v_cvt_pk_bf16_f32 on gfx950 (round-to-nearest-even)
    return __builtin_bit_cast(unsigned, v);
}

template <int TM, int TN, bool RES_LDS = false>
__device__ __forceinline__ void epilogue_rows(const GemmArgs& a, f32x16 (&acc)[TN][TM], int b, int row_base, int col_base,
                                              int out_len, int lane, unsigned char* stage /* >= 32*(TN*32*4+16) bytes, this wave's */,
                                              const unsigned char* res_lds = nullptr /* RES_LDS: LDS row of output row `row_base` */,
                                              int res_pitch = 0) {
    constexpr int NW = TN * 32;                 // channels handled by this wave
    constexpr int EP = NW * 4 + 16;             // LDS row pitch in bytes
    constexpr int LPR = NW / 8;                 // lanes per row
    constexpr int RPP = 64 / LPR;               // rows per pass
    constexpr int NP = 32 / RPP;
    const long ooff = (long)b * a.o_bs, roff = (long)b * a.r_bs, aoff = (long)b * a.a_bs;
    const int c8 = lane % LPR;
    const int n = col_base + c8 * 8;
    const bool nok = n < a.N;
    // wave-uniform mode words, read once
    const float alpha = a.alpha, oscale = a.out_scale, slope = a.slope, rinv = a.res_inv_slope;
    const int act = a.act, res_mode = RES_LDS ? 2 : a.res_mode, accum_mode = a.accum_mode, bias_mode = a.bias_mode;
    const bool has_out = a.out != nullptr, out_bf16 = a.out_dtype == DT_BF16, has_post = a.post_scale != nullptr;
    float bcol[8];
#pragma unroll
    for (int e = 0; e < 8; e++) bcol[e] = 0.f;
    if (bias_mode == 1 && nok) {
        const float4 b0 = *(const float4*)(a.bias + n), b1 = *(const float4*)(a.bias + n + 4);
        bcol[0] = b0.x; bcol[1] = b0.y; bcol[2] = b0.z; bcol[3] = b0.w; bcol[4] = b1.x; bcol[5] = b1.y; bcol[6] = b1.z; bcol[7] = b1.w;
    }
    u32x4 pk[TM][NP];                            // bf16 results, stored in one burst at the end
    // bf16 residual rows from global memory are requested one 32-row block ahead (their latency hides behind the
    // previous block's LDS transpose + arithmetic)
    const bool res_glb = !RES_LDS && res_mode && a.res_dtype == DT_BF16;
    uint4 rnext[NP];
    auto res_prefetch = [&](int j) {
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const int r = row_base + j * 32 + p * RPP + lane / LPR;
            const bool okp = r < a.M && r < out_len && nok;
            rnext[p] = *(const uint4*)((const unsigned short*)a.res + (okp ? roff + (long)r * a.ldr + n : 0));
        }
    };
    if (res_glb) res_prefetch(0);
#pragma unroll
    for (int j = 0; j < TM; j++) {
        // write this wave's 32 x NW block (MFMA layout: lane = time row, 4 consecutive channels per quad)
#pragma unroll
        for (int i = 0; i < TN; i++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                *(float4*)(stage + (lane & 31) * EP + (i * 32 + 8 * g + 4 * (lane >> 5)) * 4) =
                    make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the wave's own LDS writes have landed
        // phase A: every load of the 32-row block in flight at once (branch-free: invalid lanes read element 0)
        float v[NP][8];
        uint4 rraw[NP];
        float4 aa0[NP], aa1[NP];
        bool ok[NP];
        if (res_glb) {
#pragma unroll
            for (int p = 0; p < NP; p++) rraw[p] = rnext[p];
            if (j + 1 < TM) res_prefetch(j + 1);
        }
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const int rl = p * RPP + lane / LPR;
            const int r = row_base + j * 32 + rl;
            ok[p] = r < a.M && r < out_len && nok;
            const float4 v0 = *(const float4*)(stage + rl * EP + c8 * 32), v1 = *(const float4*)(stage + rl * EP + c8 * 32 + 16);
            v[p][0] = v0.x; v[p][1] = v0.y; v[p][2] = v0.z; v[p][3] = v0.w; v[p][4] = v1.x; v[p][5] = v1.y; v[p][6] = v1.z; v[p][7] = v1.w;
            if (RES_LDS) rraw[p] = *(const uint4*)(res_lds + (j * 32 + rl) * res_pitch + (col_base + c8 * 8) * 2);
            if (accum_mode & 1) {
                const float* ap = a.accum + (ok[p] ? aoff + (long)r * a.lda + n : 0);
                aa0[p] = *(const float4*)ap; aa1[p] = *(const float4*)(ap + 4);
            }
        }
        // phase B: arithmetic (uniform branches outside the element loops)
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const int r = row_base + j * 32 + p * RPP + lane / LPR;
            float* t = v[p];
            if (alpha != 1.f) {
#pragma unroll
                for (int e = 0; e < 8; e++) t[e] *= alpha;
            }
            if (bias_mode == 1) {
#pragma unroll
                for (int e = 0; e < 8; e++) t[e] += bcol[e];
            } else if (bias_mode == 2) {
                const float brow = ok[p] ? a.bias[r] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) t[e] += brow;
            }
            if (res_mode) {
                float q[8];
                if (RES_LDS || a.res_dtype == DT_BF16) {
                    q[0] = __uint_as_float(rraw[p].x << 16); q[1] = __uint_as_float(rraw[p].x & 0xffff0000u);
                    q[2] = __uint_as_float(rraw[p].y << 16); q[3] = __uint_as_float(rraw[p].y & 0xffff0000u);
                    q[4] = __uint_as_float(rraw[p].z << 16); q[5] = __uint_as_float(rraw[p].z & 0xffff0000u);
                    q[6] = __uint_as_float(rraw[p].w << 16); q[7] = __uint_as_float(rraw[p].w & 0xffff0000u);
                } else {
                    const float* rp = (const float*)a.res + (ok[p] ? roff + (long)r * a.ldr + n : 0);
                    const float4 t0 = *(const float4*)rp, t1 = *(const float4*)(rp + 4);
                    q[0] = t0.x; q[1] = t0.y; q[2] = t0.z; q[3] = t0.w; q[4] = t1.x; q[5] = t1.y; q[6] = t1.z; q[7] = t1.w;
                }
                if (res_mode == 2) {            // inverse leaky-relu (1/slope > 1): x = min(y, y/slope)
#pragma unroll
                    for (int e = 0; e < 8; e++) t[e] += fminf(q[e], q[e] * rinv);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e++) t[e] += q[e];
                }
            }
            if (accum_mode & 1) {
                t[0] += aa0[p].x; t[1] += aa0[p].y; t[2] += aa0[p].z; t[3] += aa0[p].w;
                t[4] += aa1[p].x; t[5] += aa1[p].y; t[6] += aa1[p].z; t[7] += aa1[p].w;
            }
            if ((accum_mode & 2) && ok[p]) {
                float* ap = a.accum + aoff + (long)r * a.lda + n;
                *(float4*)ap = make_float4(t[0], t[1], t[2], t[3]);
                *(float4*)(ap + 4) = make_float4(t[4], t[5], t[6], t[7]);
            }
            if (has_out) {
                if (oscale != 1.f) {
#pragma unroll
                    for (int e = 0; e < 8; e++) t[e] *= oscale;
                }
                if (act == ACT_LRELU) {          // 0 <= slope <= 1: leaky-relu = max(x, slope*x)
#pragma unroll
                    for (int e = 0; e < 8; e++) t[e] = fmaxf(t[e], t[e] * slope);
                } else if (act == ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; e++) t[e] = fmaxf(t[e], 0.f);
                }
                if (has_post && nok) {
                    const float4 s0 = *(const float4*)(a.post_scale + n), s1 = *(const float4*)(a.post_scale + n + 4);
                    const float4 h0 = *(const float4*)(a.post_shift + n), h1 = *(const float4*)(a.post_shift + n + 4);
                    t[0] = t[0] * s0.x + h0.x; t[1] = t[1] * s0.y + h0.y; t[2] = t[2] * s0.z + h0.z; t[3] = t[3] * s0.w + h0.w;
                    t[4] = t[4] * s1.x + h1.x; t[5] = t[5] * s1.y + h1.y; t[6] = t[6] * s1.z + h1.z; t[7] = t[7] * s1.w + h1.w;
                }
                if (out_bf16) {
                    pk[j][p] = (u32x4){pack_bf16x2(t[0], t[1]), pack_bf16x2(t[2], t[3]), pack_bf16x2(t[4], t[5]), pack_bf16x2(t[6], t[7])};
                } else if (ok[p]) {
                    float* op = (float*)a.out + ooff + (long)r * a.ldo + n;
                    *(float4*)op = make_float4(t[0], t[1], t[2], t[3]);
                    *(float4*)(op + 4) = make_float4(t[4], t[5], t[6], t[7]);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);      // reads done before the next j overwrites the stage
    }
    if (has_out && out_bf16) {
        unsigned short* obase = (unsigned short*)a.out + ooff + n;
        const int ldo = a.ldo;
#pragma unroll
        for (int j = 0; j < TM; j++)
#pragma unroll
            for (int p = 0; p < NP; p++) {
                const int r = row_base + j * 32 + p * RPP + lane / LPR;
                if (r < a.M && r < out_len && nok) *(u32x4*)(obase + (long)r * ldo) = pk[j][p];
            }
    }
}

// ================================================================================================
// conv-slab kernel: bf16 1-D convolutions / linears against STATIC weights.
//
// Each workgroup owns BM time rows x BN channels.  Per 64-channel K-chunk the input slab
// (BM + halo rows) x 64 ch is staged ONCE into LDS and re-used by every tap (a tap is just a row offset
// into the slab), instead of being re-staged per tap.  Weights never touch LDS: they are pre-packed at
// load time into MFMA-fragment order ([channel tile][K-chunk][tap][k16][lane][8 bf16], 1 KiB per
// fragment) so that a wave streams its own channel tiles with perfectly coalesced 16-byte loads straight
// into the srcA registers, one tap ahead of the MFMAs.  The main loop has no barrier except the two
// around each slab refill; several workgroups per CU overlap one block's refill with another's MFMAs.
// ================================================================================================
#define SLAB_KC 64
#define SLAB_PITCH 144      // 128 B of channels + 16 B pad: 16 consecutive rows hit 16 distinct 16-B slots

__global__ void k_pack_w(const unsigned short* w, int ntaps, int N, int K, unsigned short* out, int nkc, long total_frag_lanes) {
    // out index: ((((nt * nkc + kc) * ntaps + tap) * 4 + k16) * 64 + lane) * 8 + e
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total_frag_lanes) return;
    const int lane = idx & 63; long t = idx >> 6;
    const int k16 = t & 3; t >>= 2;
    const int tap = t % ntaps; t /= ntaps;
    const int kc = t % nkc; const int nt = t / nkc;
    const int n = nt * 32 + (lane & 31);
    const int k0 = kc * SLAB_KC + k16 * 16 + 8 * (lane >> 5);
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int k = k0 + e;
        v[e] = (n < N && k < K) ? w[((long)tap * N + n) * K + k] : (unsigned short)0;
    }
    uint4 o;
    o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16);
    o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
    *(uint4*)(out + idx * 8) = o;
}

size_t packed_weight_elems(int ntaps, int N, int K) {
    const int nkc = (K + SLAB_KC - 1) / SLAB_KC, nt = (N + 31) / 32;
    return (size_t)nt * nkc * ntaps * 4 * 64 * 8;
}
void launch_pack_weights(const void* w_bf16, int ntaps, int N, int K, void* out, hipStream_t s) {
    const int nkc = (K + SLAB_KC - 1) / SLAB_KC, nt = (N + 31) / 32;
    const long total = (long)nt * nkc * ntaps * 4 * 64;
    hipLaunchKernelGGL(k_pack_w, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const unsigned short*)w_bf16, ntaps, N, K,
                       (unsigned short*)out, nkc, total);
}

template <int BM, int BN, int WM, int WN, bool FULLK, int MINW>
__global__ __launch_bounds__(256, MINW) void convslab_kernel(const GemmArgs a) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int NIT = ((BM + 64) * 8 + 255) / 256;      // staging iterations (halo <= 64 rows)
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char slab[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave % WM, wc = wave / WM;
    const int ntn = (a.N + BN - 1) / BN;
    int wg;
    {
        const int nwg = gridDim.x, id = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int nt = wg % ntn, mt = wg / ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    const int b = blockIdx.y;
    const int out_len = a.out_len ? a.out_len[b] : a.M;
    const int in_len = a.in_len ? a.in_len[b] : a.in_len_static;
    if (m0 >= out_len || m0 >= a.M) return;

    const int HL = a.halo_l, SR = BM + a.halo_l + a.halo_r;
    const int nkc = (a.K + SLAB_KC - 1) / SLAB_KC, n16 = a.K >> 4, ntaps = a.ntaps;
    const unsigned short* Xp = (const unsigned short*)a.X + (long)b * a.x_bs;

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int j = 0; j < TM; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    // this wave's first 32-channel tile and its packed-weight stream
    const int nt32 = (n0 + wc * (BN / WN)) >> 5;
    const int nt32_total = (a.N + 31) >> 5;
    const uint4* Wq = (const uint4*)a.Wp;
    const int xrow0 = HL + wr * (BM / WM) + (lane & 31);
    const int koff = (lane >> 5) * 16;
    unsigned char* ring = slab + ((SR * SLAB_PITCH + 1023) & ~1023) + wave * (4 * TN * 1024);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)slab;   // LDS byte address of the window
    const int dvreg = a.dv[lane & (ZVX_MAX_TAPS - 1)];              // lane t holds tap t's row offset (read back with v_readlane)

    for (int kc = 0; kc < nkc; kc++) {
        if (kc) __syncthreads();
        // ---- weights: per-wave LDS ring filled by LDS-DMA (global_load_lds, 1 KiB fragment per instruction) ----
        // The packed weight stream of this wave's channel tiles is contiguous over (tap, k16): step st = tap*4+kk
        // lives at wbase + st KiB and lands in ring slot kk.  DMAs run 3 steps ahead of the MFMAs; the only
        // synchronisation is this wave's own counted vmcnt (no barrier, no VGPR staging).
        const uint4* wbase[TN];
#pragma unroll
        for (int i = 0; i < TN; i++) {
            const int t32 = nt32 + i;
            wbase[i] = Wq + (((long)(t32 < nt32_total ? t32 : 0) * nkc + kc) * ntaps) * 4 * 64 + lane;
        }
        const int nsteps = ntaps * 4;
        const long next_chunk = (kc + 1 < nkc) ? (long)ntaps * 4 * 64 : -1;   // packed stream: chunk kc+1 follows chunk kc
        auto dma = [&](int st, int slot) {
            // tail (st >= nsteps): these three requests ARE the next K-chunk's first fragments (slots 0..2) -- its weight
            // latency hides behind this chunk's last MFMAs and the slab refill; on the last chunk: harmless re-loads
            long off = (long)st * 64;
            if (st >= nsteps) off = next_chunk >= 0 ? next_chunk + (long)(st - nsteps) * 64 : (long)(nsteps - 1) * 64;
#pragma unroll
            for (int i = 0; i < TN; i++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase[i] + off),
                                                 (__attribute__((address_space(3))) void*)(ring + (slot * TN + i) * 1024), 16, 0, 0);
        };
        // the first three weight fragments are requested BEFORE the slab fill so that their L2 latency overlaps the fill's
        if (!(a.dbg & 2) && kc == 0) { dma(0, 0); dma(1, 1); dma(2, 2); }    // later chunks: requested by the previous chunk's tail
        // ---- stage the slab: rows [m0-HL, m0+BM+HR) x 64 channels of chunk kc (all loads in flight, then the stores) ----
        {
            uint4 sv[NIT];
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int c = tid + it * 256;
                const int row = c >> 3, q = c & 7;
                const int g = m0 - HL + row, k = kc * SLAB_KC + q * 8;
                sv[it] = make_uint4(0, 0, 0, 0);
                if (!(a.dbg & 4) && c < SR * 8 && g >= 0 && g < in_len && k < a.K) sv[it] = *(const uint4*)(Xp + (long)g * a.ldx + k);
            }
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int c = tid + it * 256;
                if (c < SR * 8) *(uint4*)(slab + (c >> 3) * SLAB_PITCH + (c & 7) * 16) = sv[it];
            }
        }
        __syncthreads();
        int nk16 = 4;
        if (!FULLK) { nk16 = n16 - kc * 4; if (nk16 > 4) nk16 = 4; }

        // LDS reads of the main loop are inline asm: hipcc would otherwise drain every pending LDS-DMA (vmcnt(0))
        // in front of each ds_read.  Fragments of step s+1 are read while the MFMAs of step s run (sets A/B).
        uint4 xA[TM], wA[TN], xB[TM], wB[TN];
        const unsigned ring_rd = (unsigned)(size_t)(ring - slab) + lane * 16;     // byte offset inside the dynamic LDS window
        auto rd = [&](uint4 (&xf)[TM], uint4 (&wf)[TN], int slot, unsigned rowoff, int kk) {
#pragma unroll
            for (int i = 0; i < TN; i++)
                asm volatile("ds_read_b128 %0, %1" : "=v"(wf[i]) : "v"(lds_base + ring_rd + (slot * TN + i) * 1024));
#pragma unroll
            for (int j = 0; j < TM; j++)
                asm volatile("ds_read_b128 %0, %1" : "=v"(xf[j]) : "v"(lds_base + rowoff + j * 32 * SLAB_PITCH + kk * 32));
        };
        auto mma = [&](uint4 (&xf)[TM], uint4 (&wf)[TN]) {
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int j = 0; j < TM; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[i]),
                                                                       __builtin_bit_cast(bf16x8, xf[j]), acc[i][j], 0, 0, 0);
        };
        auto rowoff_of = [&](int tap) {
            const int d = __builtin_amdgcn_readlane(dvreg, tap < ntaps ? tap : ntaps - 1);
            return (unsigned)((xrow0 + d) * SLAB_PITCH + koff);
        };
        if (a.dbg & 2) continue;
        if (TN == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        unsigned rowoff = rowoff_of(0);
        rd(xA, wA, 0, rowoff, 0);
        for (int tap = 0; tap < ntaps; tap++) {
            const unsigned rowoff_next = rowoff_of(tap + 1);
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                dma(tap * 4 + kk + 3, (kk + 3) & 3);
                // fragment of step s+1 must have landed: only the DMAs of steps s+2, s+3 may stay in flight
                if (TN == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                if (kk & 1) rd(xA, wA, (kk + 1) & 3, kk == 3 ? rowoff_next : rowoff, (kk + 1) & 3);
                else        rd(xB, wB, (kk + 1) & 3, rowoff, kk + 1);
                // wait for THIS step's set only: the TM+TN reads just issued may remain outstanding
                if (TM + TN == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                else if (TM + TN == 5) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
                else if (TM + TN == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (FULLK || kk < nk16) { if (kk & 1) mma(xB, wB); else mma(xA, wA); }
                __builtin_amdgcn_sched_barrier(0);
            }
            rowoff = rowoff_next;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // drain tail DMAs / reads before ring and slab are reused
    }
    if (a.dbg & 1) { if (acc[0][0][0] == 123.456f) ((float*)a.out)[0] = 1.f; return; }
    __syncthreads();                             // every wave is done with the slab: its LDS becomes the transpose stage
    epilogue_rows<TM, TN>(a, acc, b, m0 + wr * (BM / WM), n0 + wc * (BN / WN), out_len, lane, slab + wave * (32 * (TN * 32 * 4 + 16)));
}

// ================================================================================================
// conv-reg kernel: square bf16 convolutions with few channels (C = 32 / 64: HiFi-GAN stages 3-4).
// These layers are HBM-bound and their whole weight set is small, so each wave keeps the weight fragments
// of its 32 output channels for ALL taps in registers (NT*C/16 fragments, loaded once, coalesced, from the
// packed stream) and the main loop is nothing but ds_read_b128 of the LDS slab + MFMA: no weight traffic,
// no waits on global memory, no barrier between the slab fill and the epilogue.
// ================================================================================================
template <int C, int NT, int BM, int WM, int WN, int MINW>
__global__ __launch_bounds__(256, MINW) void convreg_kernel(const GemmArgs a) {
    constexpr int TM = BM / WM / 32;
    constexpr int KS = C / 16;                           // k16 steps per tap
    constexpr int PITCH_ = C * 2 + 16;                   // LDS row pitch (80 B / 144 B: conflict-free b128 reads)
    constexpr int CPR = C / 8;                           // 16-byte chunks per row
    constexpr int NIT = ((BM + 64) * CPR + 255) / 256;
    static_assert(WM * WN == 4 && WN * 32 == C, "wave layout");
    extern __shared__ __attribute__((aligned(16))) unsigned char slab[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave % WM, wc = wave / WM;
    int wg;
    {
        const int nwg = gridDim.x, id = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int m0 = wg * BM, b = blockIdx.y;
    const int out_len = a.out_len ? a.out_len[b] : a.M;
    const int in_len = a.in_len ? a.in_len[b] : a.in_len_static;
    if (m0 >= out_len || m0 >= a.M) return;
    const int HL = a.halo_l, SR = BM + a.halo_l + a.halo_r;
    const unsigned short* Xp = (const unsigned short*)a.X + (long)b * a.x_bs;

    // ---- all weight fragments of this wave's 32 output channels -> registers (packed stream: [nt32][tap][4 k16 slots]) ----
    uint4 w[NT][KS];
    {
        const uint4* Wq = (const uint4*)a.Wp + ((long)wc * NT * 4) * 64 + lane;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int kk = 0; kk < KS; kk++) w[t][kk] = Wq[(t * 4 + kk) * 64];
    }
    // ---- slab: rows [m0-HL, m0+BM+HR) x C channels, all loads in flight, then the LDS stores ----
    {
        uint4 sv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int c = tid + it * 256;
            const int row = c / CPR, q = c % CPR;
            const int g = m0 - HL + row;
            sv[it] = make_uint4(0, 0, 0, 0);
            if (c < SR * CPR && g >= 0 && g < in_len) sv[it] = *(const uint4*)(Xp + (long)g * a.ldx + q * 8);
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int c = tid + it * 256;
            if (c < SR * CPR) *(uint4*)(slab + (c / CPR) * PITCH_ + (c % CPR) * 16) = sv[it];
        }
    }
    __syncthreads();

    f32x16 acc[1][TM];
#pragma unroll
    for (int j = 0; j < TM; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[0][j][e] = 0.f;
    const int xrow0 = HL + wr * (BM / WM) + (lane & 31);
    const int koff = (lane >> 5) * 16;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const unsigned char* rowp = slab + (xrow0 + a.dv[t]) * PITCH_ + koff;
#pragma unroll
        for (int kk = 0; kk < KS; kk++)
#pragma unroll
            for (int j = 0; j < TM; j++) {
                const uint4 xf = *(const uint4*)(rowp + j * 32 * PITCH_ + kk * 32);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[t][kk]), __builtin_bit_cast(bf16x8, xf),
                                                                   acc[0][j], 0, 0, 0);
            }
    }
    __syncthreads();
    epilogue_rows<TM, 1>(a, acc, b, m0 + wr * (BM / WM), wc * 32, out_len, lane, slab + wave * (32 * (32 * 4 + 16)));
}

template <int C, int BM, int WM, int WN, int MINW>
static bool launch_convreg_c(const GemmArgs& a, hipStream_t stream) {
    dim3 grid((a.M + BM - 1) / BM, a.nbatch);
    size_t lds = (size_t)(BM + a.halo_l + a.halo_r) * (C * 2 + 16);
    const size_t stage = (size_t)4 * 32 * (32 * 4 + 16);
    if (lds < stage) lds = stage;
    switch (a.ntaps) {
        case 3: hipLaunchKernelGGL((convreg_kernel<C, 3, BM, WM, WN, MINW>), grid, dim3(256), lds, stream, a); return true;
        case 7: hipLaunchKernelGGL((convreg_kernel<C, 7, BM, WM, WN, MINW>), grid, dim3(256), lds, stream, a); return true;
        case 11: hipLaunchKernelGGL((convreg_kernel<C, 11, BM, WM, WN, MINW>), grid, dim3(256), lds, stream, a); return true;
    }
    return false;
}

// ================================================================================================
// resfuse kernel: one HiFi-GAN ResBlock1 iteration (hifigan.py:51-55) in a single launch for C = 32 / 64:
//     xt = lrelu(conv1_dilated(lrelu(x)) + b1)  -> stays in LDS (bf16)  ->  x' = conv2(xt) + b2 + x
// x arrives in the activated domain (lrelu(x)); the tile's conv1 output (BM rows incl. conv2's halo) never
// leaves the CU, which removes 3 of the 5 HBM passes of the unfused pair.  Weights of both convs live in
// registers (conv2's are fetched while conv1's results are written to LDS).
// ================================================================================================
template <int C, int NT, int BM, int WM, int WN, int MINW>
__global__ __launch_bounds__(256, MINW) void resfuse_kernel(const GemmArgs a) {
    constexpr int TM = BM / WM / 32;
    constexpr int KS = C / 16;
    constexpr int PITCH_ = C * 2 + 16;
    constexpr int CPR = C / 8;
    constexpr int H2 = (NT - 1) / 2;                      // conv2 halo (dilation 1)
    constexpr int BMO = BM - 2 * H2;                      // output rows per tile
    constexpr int NIT = ((BM + 64) * CPR + 255) / 256;
    static_assert(WM * WN == 4 && WN * 32 == C, "wave layout");
    extern __shared__ __attribute__((aligned(16))) unsigned char slab[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave % WM, wc = wave / WM;
    int wg;
    {
        const int nwg = gridDim.x, id = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int m0 = wg * BMO, b = blockIdx.y;
    const int out_len = a.out_len ? a.out_len[b] : a.M;
    const int in_len = a.in_len ? a.in_len[b] : a.in_len_static;
    if (m0 >= out_len || m0 >= a.M) return;
    const int H1 = a.halo_l;                              // conv1 halo = dilation*(NT-1)/2 (symmetric)
    const int SR = BM + 2 * H1;
    unsigned char* t1 = slab + ((SR * PITCH_ + 15) & ~15);
    const unsigned short* Xp = (const unsigned short*)a.X + (long)b * a.x_bs;

    // conv1's weight fragments -> registers; conv2's go to a second register set when both fit (NT*KS <= 24),
    // otherwise they replace conv1's tap by tap inside the conv1 loop (each right after that tap's last MFMA).
    constexpr bool TWO_SETS = (NT * KS <= 24);
    uint4 w[NT][KS];
    uint4 w2[TWO_SETS ? NT : 1][TWO_SETS ? KS : 1];
    const uint4* W1q = (const uint4*)a.Wp2 + ((long)wc * NT * 4) * 64 + lane;
    const uint4* W2q = (const uint4*)a.Wp + ((long)wc * NT * 4) * 64 + lane;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int kk = 0; kk < KS; kk++) {
            w[t][kk] = W1q[(t * 4 + kk) * 64];
            if (TWO_SETS) w2[t][kk] = W2q[(t * 4 + kk) * 64];
        }
    {   // slab rows s <-> global row m0 - H2 - H1 + s
        uint4 sv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int c = tid + it * 256;
            const int row = c / CPR, q = c % CPR;
            const int g = m0 - H2 - H1 + row;
            sv[it] = make_uint4(0, 0, 0, 0);
            if (c < SR * CPR && g >= 0 && g < in_len) sv[it] = *(const uint4*)(Xp + (long)g * a.ldx + q * 8);
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int c = tid + it * 256;
            if (c < SR * CPR) *(uint4*)(slab + (c / CPR) * PITCH_ + (c % CPR) * 16) = sv[it];
        }
    }
    __syncthreads();

    f32x16 acc[1][TM];
    const int koff = (lane >> 5) * 16;
    const int wrow = wr * (BM / WM);
    // ---- conv1 (dilated): T1 row i <-> global row m0 - H2 + i, reads slab rows i + H1 + dv1[t] ----
#pragma unroll
    for (int j = 0; j < TM; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[0][j][e] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const unsigned char* rowp = slab + (wrow + (lane & 31) + H1 + a.dv1[t]) * PITCH_ + koff;
#pragma unroll
        for (int kk = 0; kk < KS; kk++) {
#pragma unroll
            for (int j = 0; j < TM; j++) {
                const uint4 xf = *(const uint4*)(rowp + j * 32 * PITCH_ + kk * 32);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[t][kk]), __builtin_bit_cast(bf16x8, xf),
                                                                   acc[0][j], 0, 0, 0);
            }
            if (!TWO_SETS) w[t][kk] = W2q[(t * 4 + kk) * 64];        // conv2's fragment takes the freed register
        }
    }
    // ---- T1 = lrelu(acc + b1) as bf16, zero outside the sequence (conv2 zero-pads ITS input, hifigan.py:39-44) ----
#pragma unroll
    for (int j = 0; j < TM; j++) {
        const int i = wrow + j * 32 + (lane & 31);
        const int g = m0 - H2 + i;
        const bool inside = g >= 0 && g < in_len;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int co = wc * 32 + 8 * q + 4 * (lane >> 5);
            const float4 bb = *(const float4*)(a.bias1 + co);
            float v[4] = {acc[0][j][4 * q] + bb.x, acc[0][j][4 * q + 1] + bb.y, acc[0][j][4 * q + 2] + bb.z, acc[0][j][4 * q + 3] + bb.w};
#pragma unroll
            for (int e = 0; e < 4; e++) { v[e] = v[e] >= 0.f ? v[e] : v[e] * a.slope1; if (!inside) v[e] = 0.f; }
            uint2 pk;
            pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            *(uint2*)(t1 + i * PITCH_ + co * 2) = pk;
        }
    }
    __syncthreads();
    // ---- conv2 (dilation 1): output row j <-> global m0 + j, reads T1 rows j + H2 + dv[t] ----
#pragma unroll
    for (int j = 0; j < TM; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[0][j][e] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const unsigned char* rowp = t1 + (wrow + (lane & 31) + H2 + a.dv[t]) * PITCH_ + koff;
#pragma unroll
        for (int kk = 0; kk < KS; kk++)
#pragma unroll
            for (int j = 0; j < TM; j++) {
                const uint4 xf = *(const uint4*)(rowp + j * 32 * PITCH_ + kk * 32);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, TWO_SETS ? w2[TWO_SETS ? t : 0][TWO_SETS ? kk : 0] : w[t][kk]),
                                                                   __builtin_bit_cast(bf16x8, xf), acc[0][j], 0, 0, 0);
            }
    }
    __syncthreads();                                       // T1 is dead: its area becomes the transpose stage
    const int lim = (m0 + BMO < out_len) ? m0 + BMO : out_len;
    // residual x = inverse-lrelu of the slab rows of this tile (output row j <-> slab row j + H2 + H1): no global re-read
    epilogue_rows<TM, 1, true>(a, acc, b, m0 + wrow, wc * 32, lim, lane, t1 + wave * (32 * (32 * 4 + 16)),
                               slab + (wrow + H2 + H1) * PITCH_, PITCH_);
}

template <int C, int BM, int WM, int WN, int MINW>
static bool launch_resfuse_c(const GemmArgs& a, hipStream_t stream) {
    const int h2 = (a.ntaps - 1) / 2, bmo = BM - 2 * h2;
    dim3 grid((a.M + bmo - 1) / bmo, a.nbatch);
    const size_t pitch = C * 2 + 16;
    size_t lds = (((size_t)(BM + 2 * a.halo_l) * pitch + 15) & ~(size_t)15) + (size_t)(BM + 2 * h2 + 32) * pitch;
    switch (a.ntaps) {
        case 3: hipLaunchKernelGGL((resfuse_kernel<C, 3, BM, WM, WN, MINW>), grid, dim3(256), lds, stream, a); return true;
        case 7: hipLaunchKernelGGL((resfuse_kernel<C, 7, BM, WM, WN, MINW>), grid, dim3(256), lds, stream, a); return true;
        case 11: hipLaunchKernelGGL((resfuse_kernel<C, 11, BM, WM, WN, MINW>), grid, dim3(256), lds, stream, a); return true;
    }
    return false;
}

// Fused ResBlock1 pair; returns the variant id or -1 when the shape is not covered (caller then issues the two convs).
int launch_resfuse(GemmArgs a, hipStream_t stream) {
    if (a.dtype != DT_BF16 || !a.Wp || !a.Wp2 || a.N != a.K || a.nheads != 1 || a.wout > 0) return -1;
    if (!(a.ntaps == 3 || a.ntaps == 7 || a.ntaps == 11)) return -1;
    int h1 = 0;
    for (int i = 0; i < a.ntaps; i++) { const int d = a.dv1[i] < 0 ? -a.dv1[i] : a.dv1[i]; if (d > h1) h1 = d; }
    if (h1 > 32) return -1;
    a.halo_l = a.halo_r = h1;
    a.fused = 1;
    if (a.N == 32 && launch_resfuse_c<32, 256, 4, 1, 2>(a, stream)) return 16;
    if (a.N == 64 && launch_resfuse_c<64, 128, 2, 2, 2>(a, stream)) return 17;
    return -1;
}

struct Variant { const char* name; int dt, bm, bn; };
static const Variant kVariants[] = {
    {"gemm_bf16_128x128", DT_BF16, 128, 128}, {"gemm_bf16_256x64", DT_BF16, 256, 64},
    {"gemm_bf16_256x32", DT_BF16, 256, 32},   {"gemm_f32_128x128", DT_F32, 128, 128},
    {"gemm_f32_256x64", DT_F32, 256, 64},     {"gemm_f32_256x32", DT_F32, 256, 32},
    {"convslab_bf16_128x256", DT_BF16, 128, 256}, {"convslab_bf16_256x128", DT_BF16, 256, 128},
    {"convslab_bf16_256x64", DT_BF16, 256, 64},   {"convslab_bf16_256x32", DT_BF16, 256, 32},
    {"(unused)", DT_BF16, 0, 0}, {"(unused)", DT_BF16, 0, 0}, {"(unused)", DT_BF16, 0, 0}, {"(unused)", DT_BF16, 0, 0},
    {"convreg_bf16_c32", DT_BF16, 512, 32},       {"convreg_bf16_c64", DT_BF16, 256, 64},
    {"resfuse_bf16_c32", DT_BF16, 256, 32},       {"resfuse_bf16_c64", DT_BF16, 128, 64},
    {"gemm_bf16_64x64", DT_BF16, 64, 64},         {"gemm_f32_64x64", DT_F32, 64, 64},
};
const char* gemm_variant_name(int id) { return kVariants[id].name; }
int gemm_num_variants() { return (int)(sizeof(kVariants) / sizeof(kVariants[0])); }

template <int BM, int BN, int WM, int WN, int MINW>
static void launch_slab_variant(const GemmArgs& a, dim3 grid, size_t lds, hipStream_t stream) {
    if (a.K % SLAB_KC == 0) hipLaunchKernelGGL((convslab_kernel<BM, BN, WM, WN, true, MINW>), grid, dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((convslab_kernel<BM, BN, WM, WN, false, MINW>), grid, dim3(256), lds, stream, a);
}

static int launch_convslab(GemmArgs a, hipStream_t stream) {
    int hl = 0, hr = 0;
    for (int i = 0; i < a.ntaps; i++) { hl = a.dv[i] < -hl ? -a.dv[i] : hl; hr = a.dv[i] > hr ? a.dv[i] : hr; }
    a.halo_l = hl; a.halo_r = hr;
    static const char* noreg = getenv("ZVX_NO_CONVREG");
    if (!noreg && a.N == a.K && (a.ntaps == 3 || a.ntaps == 7 || a.ntaps == 11)) {
        if (a.N == 32 && launch_convreg_c<32, 512, 4, 1, 2>(a, stream)) return 14;
        if (a.N == 64 && launch_convreg_c<64, 256, 2, 2, 2>(a, stream)) return 15;
    }
    // tile choice: padded N weighted by the tile's MFMA efficiency
    static const int bns[4] = {256, 128, 64, 32};
    static const double eff[4] = {0.95, 1.0, 0.7, 0.4};
    int best = 0; double best_cost = -1;
    for (int i = 0; i < 4; i++) {
        const double cost = (double)((a.N + bns[i] - 1) / bns[i]) * bns[i] / eff[i];
        if (best_cost < 0 || cost < best_cost - 1e-9) { best_cost = cost; best = i; }
    }
    static const int bms[4] = {128, 256, 256, 256};
    const int bn = bns[best], bm = bms[best];
    const int ntn = (a.N + bn - 1) / bn;
    const int ntm = (a.M + bm - 1) / bm;
    dim3 grid(ntn * ntm, a.nbatch);
    const int tn = (bn >= 128) ? 2 : 1;
    size_t lds = (((size_t)(bm + hl + hr) * SLAB_PITCH + 1023) & ~(size_t)1023) + (size_t)4 * 4 * tn * 1024;
    const size_t stage = (size_t)4 * 32 * (tn * 128 + 16);
    if (lds < stage) lds = stage;
    switch (best) {
        case 0: launch_slab_variant<128, 256, 1, 4, 2>(a, grid, lds, stream); break;
        case 1: launch_slab_variant<256, 128, 2, 2, 2>(a, grid, lds, stream); break;
        case 2: launch_slab_variant<256, 64, 2, 2, 2>(a, grid, lds, stream); break;
        case 3: launch_slab_variant<256, 32, 4, 1, 2>(a, grid, lds, stream); break;
    }
    return 6 + best;
}

int launch_gemm(const GemmArgs& a, hipStream_t stream) {
    if (a.N <= 0 || a.M <= 0 || a.nbatch <= 0) return -1;
    if (a.ntaps < 1 || a.ntaps > ZVX_MAX_TAPS) return -2;
    // N not a multiple of 4: the last 4-wide store spills into [N, roundup4(N)) of the row (ldo must cover it)
    if (a.N % 4 && (a.bias_mode == 1 || a.res_mode || a.accum_mode || a.post_scale || a.ldo < ((a.N + 3) & ~3))) return -2;
    if (a.Wp && a.dtype == DT_BF16 && a.wout <= 0 && a.nheads == 1 && a.w_bs == 0 && !a.k_len && a.K % 16 == 0 && a.N % 8 == 0 && a.ldo % 8 == 0) {
        int lo = 0, hi = 0;
        for (int i = 0; i < a.ntaps; i++) { if (a.dv[i] < lo) lo = a.dv[i]; if (a.dv[i] > hi) hi = a.dv[i]; }
        if (hi - lo <= 64 && hi >= 0 && lo <= 0) return launch_convslab(a, stream);
    }
    // tile choice: padded N weighted by the tile's MFMA efficiency, ties -> wider BN
    static const int bns[3] = {128, 64, 32};
    static const double eff[3] = {1.0, 0.75, 0.45};
    int best = 0; double best_cost = -1;
    for (int i = 0; i < 3; i++) {
        const double cost = (double)((a.N + bns[i] - 1) / bns[i]) * bns[i] / eff[i];
        if (best_cost < 0 || cost < best_cost - 1e-9) { best_cost = cost; best = i; }
    }
    int bn = bns[best], bm = (bn == 128) ? 128 : 256;
    int ntn = (a.N + bn - 1) / bn, ntm = (a.M + bm - 1) / bm;
    // small problems (e.g. the f32 phoneme encoder: 128 rows per utterance): a 64x64 tile fills the 256 CUs
    if ((long)ntn * ntm * a.nbatch * a.nheads < 512 && a.N >= 64) {
        bm = bn = 64; ntn = (a.N + 63) / 64; ntm = (a.M + 63) / 64;
        dim3 grid(ntn * ntm, a.nbatch * a.nheads), block(256);
        if (a.dtype == DT_BF16) { hipLaunchKernelGGL((gemm_kernel<DT_BF16, 64, 64, 2, 2>), grid, block, 0, stream, a); return 18; }
        hipLaunchKernelGGL((gemm_kernel<DT_F32, 64, 64, 2, 2>), grid, block, 0, stream, a);
        return 19;
    }
    dim3 grid(ntn * ntm, a.nbatch * a.nheads), block(256);
    const int base = (a.dtype == DT_BF16) ? 0 : 3;
    const int id = base + best;
    switch (id) {
        case 0: hipLaunchKernelGGL((gemm_kernel<DT_BF16, 128, 128, 2, 2>), grid, block, 0, stream, a); break;
        case 1: hipLaunchKernelGGL((gemm_kernel<DT_BF16, 256, 64, 4, 1>), grid, block, 0, stream, a); break;
        case 2: hipLaunchKernelGGL((gemm_kernel<DT_BF16, 256, 32, 4, 1>), grid, block, 0, stream, a); break;
        case 3: hipLaunchKernelGGL((gemm_kernel<DT_F32, 128, 128, 2, 2>), grid, block, 0, stream, a); break;
        case 4: hipLaunchKernelGGL((gemm_kernel<DT_F32, 256, 64, 4, 1>), grid, block, 0, stream, a); break;
        case 5: hipLaunchKernelGGL((gemm_kernel<DT_F32, 256, 32, 4, 1>), grid, block, 0, stream, a); break;
    }
    return id;
}

}  // namespace zvx
