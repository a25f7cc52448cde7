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
    const long ooff = (long)b * a.o_bs + (long)h * a.o_hs;
    const long roff = (long)b * a.r_bs + (long)h * a.r_hs;
    const long aoff = (long)b * a.a_bs;
#pragma unroll
    for (int j = 0; j < TM; j++) {
        const int r = m0 + wr * (BM / WM) + j * 32 + (lane & 31);
        bool rok = r < a.M;
        if (two_d) rok = rok && (r % a.wout) < out_len; else rok = rok && r < out_len;
        if (!rok) continue;
        const float brow = (a.bias_mode == 2) ? a.bias[r] : 0.f;
#pragma unroll
        for (int i = 0; i < TN; i++) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int n = n0 + wc * (BN / WN) + i * 32 + 8 * g + 4 * (lane >> 5);
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

struct Variant { const char* name; int dt, bm, bn; };
static const Variant kVariants[] = {
    {"gemm_bf16_128x128", DT_BF16, 128, 128}, {"gemm_bf16_256x64", DT_BF16, 256, 64},
    {"gemm_bf16_256x32", DT_BF16, 256, 32},   {"gemm_f32_128x128", DT_F32, 128, 128},
    {"gemm_f32_256x64", DT_F32, 256, 64},     {"gemm_f32_256x32", DT_F32, 256, 32},
};
const char* gemm_variant_name(int id) { return kVariants[id].name; }
int gemm_num_variants() { return (int)(sizeof(kVariants) / sizeof(kVariants[0])); }

int launch_gemm(const GemmArgs& a, hipStream_t stream) {
    if (a.N <= 0 || a.M <= 0 || a.nbatch <= 0) return -1;
    if (a.ntaps < 1 || a.ntaps > ZVX_MAX_TAPS) return -2;
    // N not a multiple of 4: the last 4-wide store spills into [N, roundup4(N)) of the row (ldo must cover it)
    if (a.N % 4 && (a.bias_mode == 1 || a.res_mode || a.accum_mode || a.post_scale || a.ldo < ((a.N + 3) & ~3))) return -2;
    // tile choice: least padded N, ties -> wider BN
    static const int bns[3] = {128, 64, 32};
    int best = 0; long best_pad = -1;
    for (int i = 0; i < 3; i++) {
        long pad = (long)((a.N + bns[i] - 1) / bns[i]) * bns[i];
        if (best_pad < 0 || pad < best_pad) { best_pad = pad; best = i; }
    }
    const int bn = bns[best], bm = (bn == 128) ? 128 : 256;
    const int ntn = (a.N + bn - 1) / bn, ntm = (a.M + bm - 1) / bm;
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
