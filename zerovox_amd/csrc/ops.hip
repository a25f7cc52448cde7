// ops.hip -- the HBM-bound kernels of the synthesis path (norms, softmax, gathers, pooling, conv_post).
// All tensors are time-major [row][channel]; 64-lane waves; one wave per row for row reductions,
// channel-contiguous vector loads elsewhere.
#include "zvx_kernels.h"

namespace zvx {

__device__ __forceinline__ float ld(const void* p, int dt, long i) {
    if (dt == DT_F16) return (float)((const _Float16*)p)[i];
    return dt == DT_F32 ? ((const float*)p)[i] : __uint_as_float(((unsigned)((const unsigned short*)p)[i]) << 16);
}
__device__ __forceinline__ unsigned short tobf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
// 8 consecutive 16-bit values (one 16-byte load) -> f32, and back; dt = DT_BF16 or DT_F16
__device__ __forceinline__ void unpack8(const uint4 t, int dt, float v[8]) {
    if (dt == DT_F16) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 a = __builtin_bit_cast(h2, t.x), b = __builtin_bit_cast(h2, t.y), c = __builtin_bit_cast(h2, t.z), d = __builtin_bit_cast(h2, t.w);
        v[0] = (float)a.x; v[1] = (float)a.y; v[2] = (float)b.x; v[3] = (float)b.y; v[4] = (float)c.x; v[5] = (float)c.y; v[6] = (float)d.x; v[7] = (float)d.y;
    } else {
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
        v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
        v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
    }
}
__device__ __forceinline__ float clamp_h(float v) { return __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }   // f16 stores saturate instead of producing Inf
__device__ __forceinline__ uint4 pack8(const float v[8], int dt) {
    uint4 o;
    if (dt == DT_F16) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        o.x = __builtin_bit_cast(unsigned, (h2){(_Float16)clamp_h(v[0]), (_Float16)clamp_h(v[1])}); o.y = __builtin_bit_cast(unsigned, (h2){(_Float16)clamp_h(v[2]), (_Float16)clamp_h(v[3])});
        o.z = __builtin_bit_cast(unsigned, (h2){(_Float16)clamp_h(v[4]), (_Float16)clamp_h(v[5])}); o.w = __builtin_bit_cast(unsigned, (h2){(_Float16)clamp_h(v[6]), (_Float16)clamp_h(v[7])});
    } else {
        o.x = (unsigned)tobf(v[0]) | ((unsigned)tobf(v[1]) << 16); o.y = (unsigned)tobf(v[2]) | ((unsigned)tobf(v[3]) << 16);
        o.z = (unsigned)tobf(v[4]) | ((unsigned)tobf(v[5]) << 16); o.w = (unsigned)tobf(v[6]) | ((unsigned)tobf(v[7]) << 16);
    }
    return o;
}
__device__ __forceinline__ void st(void* p, int dt, long i, float v) {
    if (dt == DT_F16) { ((_Float16*)p)[i] = (_Float16)clamp_h(v); return; }
    if (dt == DT_F32) ((float*)p)[i] = v; else ((unsigned short*)p)[i] = tobf(v);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_LRELU) return v >= 0.f ? v : v * slope;
    if (act == ACT_TANH) return tanhf(v);
    return v;
}

// ---------------------------------------------------------------- casts
__global__ void k_cast(const void* in, int idt, void* out, int odt, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) st(out, odt, i, ld(in, idt, i));
}
void launch_cast(const void* in, int in_dt, void* out, int out_dt, size_t n, hipStream_t s) {
    if (!n) return;
    int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_cast, dim3(blocks), dim3(256), 0, s, in, in_dt, out, out_dt, n);
}
// ---------------------------------------------------------------- 16-bit transpose  [b][rows][ld_in] -> [b][C][ld_out]
__global__ __launch_bounds__(256) void k_transpose16(const unsigned short* in, int ld_in, unsigned short* out, int ld_out, int rows_max, int C, const int* len) {
    __shared__ unsigned short tile[64][66];
    const int b = blockIdx.z, r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    // rows past the utterance's own length are never written by the producing GEMM (it masks by out_len): they read as zeros here,
    // whatever an earlier call (or another precision's use of the shared buffer) left there -- 0 * stale NaN = NaN in P.V otherwise
    const int rows = len ? min(len[b], rows_max) : rows_max;
    const unsigned short* ip = in + (long)b * rows_max * ld_in;
    unsigned short* op = out + (long)b * C * ld_out;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < rows && c0 + c < C) ? ip[(long)(r0 + r) * ld_in + c0 + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (c0 + c < C && r0 + r < ld_out) op[(long)(c0 + c) * ld_out + r0 + r] = (r0 + r < rows) ? tile[r][c] : (unsigned short)0;
    }
}
void launch_transpose16(const void* in, int ld_in, void* out, int ld_out, int B, int rows, int C, hipStream_t s, const int* len) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(k_transpose16, dim3((C + 63) / 64, (ld_out + 63) / 64, B), dim3(256), 0, s, (const unsigned short*)in, ld_in, (unsigned short*)out, ld_out, rows, C, len);
}

void launch_f32_to_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s) { launch_cast(in, DT_F32, out, DT_BF16, n, s); }

// ---------------------------------------------------------------- 16-bit split planes of an f32 tensor
// f32-class GEMMs on the 16-bit MFMA: x = hi + lo, w = wh + wl, and
//     x.w ~= hi.wh + hi.wl + lo.wh
// is ONE 16-bit GEMM over a K axis of three planes: activations [hi | hi | lo] against weights [wh | wl | wh], accumulated in
// f32 by the MFMA.  k_split3 writes the activation planes (rows past an utterance's length as zeros), k_split3_w the weights.
//   bf16 planes (f16 = 0, the round-2/3 path, kept as an A/B): 8-bit significands, hi + lo carries 16 bits, the dropped lo.wl
//     term is 2^-18 relative -- 5e-5 on the encoder output.
//   IEEE-half planes (f16 = 1, default): 11-bit significands, hi + lo carries 22 bits + sign (2^-24 relative: an f32 half-ulp),
//     the dropped term is 2^-24.  Half has 5 exponent bits, so the planes are SCALED by powers of two to stay normal:
//       activations [xh | xh | xl * 2^11]           (xl = x - xh is 2^-12 |x| at most; x itself sits behind a LayerNorm)
//       weights     [wh | wl | w * 2^-11] * 2^s     (s per tensor: max |w| 2^s in [2^14, 2^15); the epilogue multiplies by 2^-s)
//     every product pair shares the scale 2^s: xh.wh, xh.wl, (xl 2^11).(w 2^(s-11)).
__device__ __forceinline__ void split2(float v, unsigned short& hi, unsigned short& lo) {
    hi = tobf(v);
    lo = tobf(v - __uint_as_float(((unsigned)hi) << 16));
}
__device__ __forceinline__ void split2h(float v, unsigned short& hi, unsigned short& lo) {
    const _Float16 h = (_Float16)clamp_h(v);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, (_Float16)clamp_h((v - (float)h) * 2048.f));
}
__device__ __forceinline__ void split2x(float v, int f16, unsigned short& hi, unsigned short& lo) { if (f16) split2h(v, hi, lo); else split2(v, hi, lo); }
__global__ __launch_bounds__(256) void k_split3(const float* x, int ldx, unsigned short* out, int rows_max, const int* rows, int C, int f16) {
    const int b = blockIdx.y, r = blockIdx.x;
    const bool live = r < (rows ? rows[b] : rows_max);
    const float* xr = x + ((long)b * rows_max + r) * ldx;
    unsigned short* o = out + ((long)b * rows_max + r) * 3 * C;
    for (int c = threadIdx.x * 4; c < C; c += blockDim.x * 4) {                 // C % 4 == 0
        unsigned short h[4], l[4];
        if (live) { const float4 v = *(const float4*)(xr + c); split2x(v.x, f16, h[0], l[0]); split2x(v.y, f16, h[1], l[1]); split2x(v.z, f16, h[2], l[2]); split2x(v.w, f16, h[3], l[3]); }
        else { for (int e = 0; e < 4; e++) h[e] = l[e] = 0; }
        const uint2 hv = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
        const uint2 lv = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
        *(uint2*)(o + c) = hv; *(uint2*)(o + C + c) = hv; *(uint2*)(o + 2 * C + c) = lv;
    }
}
void launch_split3(const float* x, int ldx, void* out, int B, int rows_max, const int* rows, int C, hipStream_t s, int f16) {
    if (rows_max <= 0) return;
    hipLaunchKernelGGL(k_split3, dim3(rows_max, B), dim3(C >= 1024 ? 256 : 128), 0, s, x, ldx, (unsigned short*)out, rows_max, rows, C, f16);
}
__global__ void k_split3_w(const float* w, unsigned short* out, long nrows, int K, int f16, float scale) {
    const long r = blockIdx.x;
    if (r >= nrows) return;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        unsigned short h, l, h3;
        if (f16) {
            const float v = w[r * K + k] * scale;                                // scale = 2^s: exact
            const _Float16 wh = (_Float16)clamp_h(v);
            h = __builtin_bit_cast(unsigned short, wh);
            l = __builtin_bit_cast(unsigned short, (_Float16)(v - (float)wh));
            h3 = __builtin_bit_cast(unsigned short, (_Float16)clamp_h(v * (1.0f / 2048.f)));
        } else { split2(w[r * K + k], h, l); h3 = h; }
        out[r * 3 * K + k] = h; out[r * 3 * K + K + k] = l; out[r * 3 * K + 2 * K + k] = h3;
    }
}
void launch_split3_weights(const float* w, void* out, long nrows, int K, hipStream_t s, int f16, float scale) {
    hipLaunchKernelGGL(k_split3_w, dim3((unsigned)nrows), dim3(256), 0, s, w, (unsigned short*)out, nrows, K, f16, scale);
}
// max |x| over n floats (load-time: the per-tensor scale of the half-precision weight planes); out must be zeroed
__global__ void k_absmax(const float* x, size_t n, unsigned* out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));           // non-negative floats order like their bit patterns
}
void launch_absmax(const float* x, size_t n, float* out, hipStream_t s) {
    (void)hipMemsetAsync(out, 0, 4, s);
    if (!n) return;
    int blocks = (int)((n + 255) / 256); if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_absmax, dim3(blocks), dim3(256), 0, s, x, n, (unsigned*)out);
}

// ---------------------------------------------------------------- embedding + positional encoding
__global__ void k_embed(const int* ph, const int* pu, const float* emb, int ed, const float* pemb, int pd,
                        const float* pe, float* out, int Tmax, const int* T) {
    const int b = blockIdx.y, t = blockIdx.x;
    if (t >= T[b]) return;
    const int H = ed + pd;
    const int p = ph[b * Tmax + t], q = pu[b * Tmax + t];
    float* o = out + ((long)b * Tmax + t) * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        float v = c < ed ? emb[(long)p * ed + c] : pemb[(long)q * pd + (c - ed)];
        o[c] = v + pe[(long)t * H + c];
    }
}
void launch_embed(const int* phoneme, const int* puncts, const float* emb, int emb_dim, const float* pemb,
                  int pemb_dim, const float* pe, float* out, int B, int Tmax, const int* T, hipStream_t s) {
    hipLaunchKernelGGL(k_embed, dim3(Tmax, B), dim3(128), 0, s, phoneme, puncts, emb, emb_dim, pemb, pemb_dim, pe, out, Tmax, T);
}

// ---------------------------------------------------------------- LayerNorm / SCLN, one wave per row
__global__ void k_layernorm(const void* x, int xdt, int ldx, void* y, int ydt, int ldy, int rows_max, const int* rows,
                            int C, int mode, float eps, const float* gamma, const float* beta, const float* bg,
                            long bg_bs, const float* post_add, unsigned short* planes, int planes_f16) {
    const int b = blockIdx.y;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int nr = rows ? rows[b] : rows_max;
    if (r >= nr) {
        // planes: the [hi | hi | lo] 16-bit split of the result for the next split-product GEMM (what k_split3 would write), zeros past the utterance
        if (planes && r < rows_max) { unsigned short* o = planes + ((long)b * rows_max + r) * 3 * C; for (int c = lane; c < 3 * C; c += 64) o[c] = 0; }
        return;
    }
    const long xo = ((long)b * rows_max + r) * ldx, yo = ((long)b * rows_max + r) * ldy;
    if ((C & 7) == 0 && C <= 1024 && (ldx & 7) == 0 && (ldy & 7) == 0) {
        // (round 6: every output form takes this path -- the phoneme encoder's f32 rows + 16-bit split planes and the variance predictors' f32 rows too)
        // 16-bit output rows (the FFT-block decoder's 12 LayerNorm / SCLN passes per call, f32 or 16-bit in): the row is read ONCE, as
        // 16-byte vectors (8 elements per lane and round, one or two rounds), and stays in registers for the mean, the centred second
        // moment and the affine -- the element-wise form below reads it three times, element by element: 52 us for 28672 x 528 (1.7 TB/s)
        float v[2][8];
        bool has[2];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int c = (lane + 64 * i) * 8;
            has[i] = c < C;
            if (has[i] && xdt == DT_F32) {
                const float4 t0 = *(const float4*)((const float*)x + xo + c), t1 = *(const float4*)((const float*)x + xo + c + 4);
                v[i][0] = t0.x; v[i][1] = t0.y; v[i][2] = t0.z; v[i][3] = t0.w; v[i][4] = t1.x; v[i][5] = t1.y; v[i][6] = t1.z; v[i][7] = t1.w;
            } else if (has[i]) unpack8(*(const uint4*)((const unsigned short*)x + xo + c), xdt, v[i]);
            else { for (int e = 0; e < 8; e++) v[i][e] = 0.f; }
#pragma unroll
            for (int e = 0; e < 8; e++) s += v[i][e];
        }
        const float mu = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int e = 0; e < 8; e++) { const float d = has[i] ? v[i][e] - mu : 0.f; q += d * d; }
        q = wave_sum(q);
        const float inv = mode == 0 ? 1.0f / sqrtf(q / C + eps) : 1.0f / (sqrtf(q / (C - 1)) + eps);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            if (!has[i]) continue;
            const int c = (lane + 64 * i) * 8;
            const float* gp = mode == 0 ? gamma + c : bg + (long)b * bg_bs + C + c;
            const float* bp = mode == 0 ? beta + c : bg + (long)b * bg_bs + c;
            const float4 g0 = *(const float4*)gp, g1 = *(const float4*)(gp + 4), b0 = *(const float4*)bp, b1 = *(const float4*)(bp + 4);
            const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = (v[i][e] - mu) * inv * g[e] + be[e];
            if (post_add) {
                const float4 p0 = *(const float4*)(post_add + (long)b * C + c), p1 = *(const float4*)(post_add + (long)b * C + c + 4);
                o[0] += p0.x; o[1] += p0.y; o[2] += p0.z; o[3] += p0.w; o[4] += p1.x; o[5] += p1.y; o[6] += p1.z; o[7] += p1.w;
            }
            if (ydt != DT_F32) *(uint4*)((unsigned short*)y + yo + c) = pack8(o, ydt);
            else {
                *(float4*)((float*)y + yo + c) = make_float4(o[0], o[1], o[2], o[3]);
                *(float4*)((float*)y + yo + c + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
            if (planes) {                                    // [hi | hi | lo] of the result: the next split-product GEMM's operand (what k_split3 would write)
                unsigned short hh[8], ll[8];
#pragma unroll
                for (int e = 0; e < 8; e++) split2x(o[e], planes_f16, hh[e], ll[e]);
                const uint4 hv = make_uint4(hh[0] | ((unsigned)hh[1] << 16), hh[2] | ((unsigned)hh[3] << 16), hh[4] | ((unsigned)hh[5] << 16), hh[6] | ((unsigned)hh[7] << 16));
                const uint4 lv = make_uint4(ll[0] | ((unsigned)ll[1] << 16), ll[2] | ((unsigned)ll[3] << 16), ll[4] | ((unsigned)ll[5] << 16), ll[6] | ((unsigned)ll[7] << 16));
                unsigned short* po = planes + ((long)b * rows_max + r) * 3 * C + c;
                *(uint4*)po = hv; *(uint4*)(po + C) = hv; *(uint4*)(po + 2 * C) = lv;
            }
        }
        return;
    }
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += ld(x, xdt, xo + c);
    const float mu = wave_sum(s) / C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { float d = ld(x, xdt, xo + c) - mu; q += d * d; }
    q = wave_sum(q);
    float inv;
    if (mode == 0) inv = 1.0f / sqrtf(q / C + eps);            // torch LayerNorm: biased var, eps inside sqrt
    else inv = 1.0f / (sqrtf(q / (C - 1)) + eps);              // SCLN: unbiased std, (sigma + eps)  fs2.py:79-81
    for (int c = lane; c < C; c += 64) {
        float g, be;
        if (mode == 0) { g = gamma[c]; be = beta[c]; }
        else { be = bg[(long)b * bg_bs + c]; g = bg[(long)b * bg_bs + C + c]; }   // b = first half (fs2.py:85)
        float v = (ld(x, xdt, xo + c) - mu) * inv * g + be;
        if (post_add) v += post_add[(long)b * C + c];
        st(y, ydt, yo + c, v);
        if (planes) {
            unsigned short hi, lo;
            split2x(v, planes_f16, hi, lo);
            unsigned short* o = planes + ((long)b * rows_max + r) * 3 * C;
            o[c] = hi; o[C + c] = hi; o[2 * C + c] = lo;
        }
    }
}
void launch_layernorm(const void* x, int x_dt, int ldx, void* y, int y_dt, int ldy, int B, int rows_max,
                      const int* rows, int C, int mode, float eps, const float* gamma, const float* beta,
                      const float* bg, long bg_bs, const float* post_add, hipStream_t s, void* split_planes, int planes_f16) {
    hipLaunchKernelGGL(k_layernorm, dim3((rows_max + 3) / 4, B), dim3(256), 0, s, x, x_dt, ldx, y, y_dt, ldy, rows_max,
                       rows, C, mode, eps, gamma, beta, bg, bg_bs, post_add, (unsigned short*)split_planes, planes_f16);
}

// ---------------------------------------------------------------- row softmax with key-length mask
__global__ void k_softmax_rows(const float* sc, int lds, void* P, int pdt, int ldp, int nheads, int Lmax, const int* len) {
    const int z = blockIdx.y, b = z / nheads;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int L = len[b];
    if (r >= L) return;
    const float* row = sc + ((long)z * Lmax + r) * lds;
    const long po = ((long)z * Lmax + r) * ldp;
    float m = -INFINITY;
    for (int c = lane; c < L; c += 64) m = fmaxf(m, row[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < L; c += 64) s += expf(row[c] - m);
    s = wave_sum(s);
    const float inv = 1.0f / s;
    const int Lp = (L + 7) & ~7;
    for (int c = lane; c < Lp; c += 64) st(P, pdt, po + c, c < L ? expf(row[c] - m) * inv : 0.f);
}
void launch_softmax_rows(const float* scores, int lds, void* P, int p_dt, int ldp, int nbatch, int nheads,
                         int Lmax, const int* len, hipStream_t s) {
    hipLaunchKernelGGL(k_softmax_rows, dim3((Lmax + 3) / 4, nbatch * nheads), dim3(256), 0, s, scores, lds, P, p_dt, ldp, nheads, Lmax, len);
}

// ---------------------------------------------------------------- row dot (variance predictor head)
__global__ void k_rowdot(const float* x, int ldx, const float* w, float bias, float* out, int Tmax, const int* T, int C) {
    const int b = blockIdx.y, t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T[b]) return;
    const float* row = x + ((long)b * Tmax + t) * ldx;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += row[c] * w[c];
    s = wave_sum(s);
    if (lane == 0) out[b * Tmax + t] = s + bias;
}
void launch_rowdot(const float* x, int ldx, const float* w, float bias, float* out, int B, int Tmax,
                   const int* T, int C, hipStream_t s) {
    hipLaunchKernelGGL(k_rowdot, dim3((Tmax + 3) / 4, B), dim3(256), 0, s, x, ldx, w, bias, out, Tmax, T, C);
}

// ---------------------------------------------------------------- bucketise + embedding add
__global__ void k_bucket_embed_add(const float* pred, const float* table, int nb, float* x, int ldx, int C, int* idx_out,
                                   int Tmax, const int* T) {
    const int b = blockIdx.y, t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T[b]) return;
    float p = rintf(pred[b * Tmax + t] * (float)(nb - 1));        // torch.round: half-to-even
    p = fminf(fmaxf(p, 0.f), (float)(nb - 1));
    const int idx = (int)p;
    if (lane == 0 && idx_out) idx_out[b * Tmax + t] = idx;
    float* row = x + ((long)b * Tmax + t) * ldx;
    for (int c = lane; c < C; c += 64) row[c] += table[(long)idx * C + c];
}
void launch_bucket_embed_add(const float* pred, const float* table, int nbins, float* x, int ldx, int C,
                             int* idx_out, int B, int Tmax, const int* T, hipStream_t s) {
    hipLaunchKernelGGL(k_bucket_embed_add, dim3((Tmax + 3) / 4, B), dim3(256), 0, s, pred, table, nbins, x, ldx, C, idx_out, Tmax, T);
}

// ---------------------------------------------------------------- durations + inclusive scan (one wave / utterance)
__global__ void k_durations(const int* forced, const float* logd, int* dur, int* cum, int* mel_len, int Tmax, const int* T) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = T[b];
    long long carry = 0;                                        // 64-bit: the sum saturates instead of wrapping
    for (int t0 = 0; t0 < n; t0 += 64) {
        const int t = t0 + lane;
        int d = 0;
        if (t < n) {
            if (forced) d = min(max(forced[b * Tmax + t], 0), 65536);                     // fs2.py:452 max(int(d),0)
            else d = (int)fminf(fmaxf(rintf(expf(logd[b * Tmax + t]) - 1.0f), 0.f), 65536.f);   // fs2.py:678-681; NaN -> 0; the upper
                                                                     // clamp only keeps the int conversion and the prefix sum defined
            dur[b * Tmax + t] = d;
        }
        int v = d;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int u = __shfl_up(v, o, 64); if (lane >= o) v += u; }
        if (t < n) cum[b * Tmax + t] = (int)min(carry + v, 0x7fffffffLL);
        carry += __shfl(v, 63, 64);
    }
    if (lane == 0) mel_len[b] = (int)min(carry, 0x7fffffffLL);
}
void launch_durations(const int* forced, const float* logd, int* dur, int* cum, int* mel_len, int B, int Tmax,
                      const int* T, hipStream_t s) {
    hipLaunchKernelGGL(k_durations, dim3(B), dim3(64), 0, s, forced, logd, dur, cum, mel_len, Tmax, T);
}

// ---------------------------------------------------------------- length regulator: scan-indexed coalesced row gather
__global__ void k_length_regulate(const float* x, int ldx, const int* cum, const int* T, const int* mel_len, float* feats,
                                  int Tmax, int Lmax, int C) {
    const int b = blockIdx.y, l = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (l >= mel_len[b]) return;
    const int* cb = cum + b * Tmax;
    int lo = 0, hi = T[b] - 1;                 // first t with cum[t] > l
    while (lo < hi) { int mid = (lo + hi) >> 1; if (cb[mid] > l) hi = mid; else lo = mid + 1; }
    const float4* src = (const float4*)(x + ((long)b * Tmax + lo) * ldx);
    float4* dst = (float4*)(feats + ((long)b * Lmax + l) * C);
    for (int c = lane; c < C / 4; c += 64) dst[c] = src[c];
}
void launch_length_regulate(const float* x, int ldx, const int* cum, const int* T, const int* mel_len,
                            float* feats, int B, int Tmax, int Lmax, int C, hipStream_t s) {
    if (Lmax <= 0) return;
    hipLaunchKernelGGL(k_length_regulate, dim3((Lmax + 3) / 4, B), dim3(256), 0, s, x, ldx, cum, T, mel_len, feats, Tmax, Lmax, C);
}

__global__ void k_add_pe_cast(const float* x, const float* pe, void* y, int ydt, int ldy, int Lmax, const int* L, int C, int out_rows) {
    const int b = blockIdx.y, l = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (l >= L[b]) return;
    const float* row = x + ((long)b * Lmax + l) * C;
    const long yo = ((long)b * out_rows + l) * ldy;
    for (int c = lane; c < C; c += 64) st(y, ydt, yo + c, row[c] + (pe ? pe[(long)l * C + c] : 0.f));
}
void launch_add_pe_cast(const float* x, const float* pe, void* y, int y_dt, int ldy, int B, int Lmax,
                        const int* L, int C, hipStream_t s, int out_rows_max) {
    if (Lmax <= 0) return;
    hipLaunchKernelGGL(k_add_pe_cast, dim3((Lmax + 3) / 4, B), dim3(256), 0, s, x, pe, y, y_dt, ldy, Lmax, L, C, out_rows_max > 0 ? out_rows_max : Lmax);
}

// ---------------------------------------------------------------- per-(utterance, channel) statistics over valid rows
// x [b][H*Wmax][ldx]; row r valid iff (r % Wmax) < W[b].  Block = 32 row-groups x 8 lanes, each lane owns 8
// consecutive channels (one 16-byte load per row for bf16); single pass with a per-channel shift (the first
// valid row) so that sum / sum-of-squares do not cancel; LDS tree over the row-groups.
__global__ __launch_bounds__(256) void k_colstats(const void* x, int xdt, int ldx, int H, int Wmax, const int* W, int C, float eps,
                                                  float* mean, float* rstd) {
    __shared__ float red[2][32][65];
    const int b = blockIdx.y, cl = (threadIdx.x & 7) * 8, c0 = blockIdx.x * 64 + cl, g = threadIdx.x >> 3;
    const int Wb = W[b];
    const long base = (long)b * H * Wmax * ldx;
    float s1[8], s2[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { s1[e] = 0.f; s2[e] = 0.f; sh[e] = 0.f; }
    const bool cok = c0 < C;                                  // C % 8 == 0 for every caller
    auto load8 = [&](long off, float v[8]) {
        if (xdt != DT_F32) {
            unpack8(*(const uint4*)((const unsigned short*)x + off), xdt, v);
        } else {
            const float4 t0 = *(const float4*)((const float*)x + off), t1 = *(const float4*)((const float*)x + off + 4);
            v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
        }
    };
    if (cok && Wb > 0) load8(base + c0, sh);                  // shift = row 0 of this utterance (valid: Wb > 0)
    if (cok && xdt != DT_F32) {
        // four rows of the thread in flight per iteration (same rows, same order of accumulation as the plain loop below); a single
        // request would otherwise pay one memory round trip per row of its chain (14 for a 448-frame utterance)
        const unsigned short* xp = (const unsigned short*)x + base + c0;
        for (int hh = 0; hh < H; hh++)
            for (int w = g; w < Wb; w += 128) {
                uint4 t[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int wu = w + 32 * u < Wb ? w + 32 * u : w; t[u] = *(const uint4*)(xp + ((long)hh * Wmax + wu) * ldx); }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (w + 32 * u >= Wb) break;
                    float v[8];
                    unpack8(t[u], xdt, v);
#pragma unroll
                    for (int e = 0; e < 8; e++) { const float d = v[e] - sh[e]; s1[e] += d; s2[e] += d * d; }
                }
            }
    } else if (cok)
        for (int hh = 0; hh < H; hh++)
            for (int w = g; w < Wb; w += 32) {
                float v[8];
                load8(base + ((long)hh * Wmax + w) * ldx + c0, v);
#pragma unroll
                for (int e = 0; e < 8; e++) { const float d = v[e] - sh[e]; s1[e] += d; s2[e] += d * d; }
            }
#pragma unroll
    for (int e = 0; e < 8; e++) { red[0][g][cl + e] = s1[e]; red[1][g][cl + e] = s2[e]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = blockIdx.x * 64 + threadIdx.x;
        float a1 = 0.f, a2 = 0.f;
        for (int i = 0; i < 32; i++) { a1 += red[0][i][threadIdx.x]; a2 += red[1][i][threadIdx.x]; }
        if (c < C) {
            const float cnt = (float)H * (float)Wb;
            const float shift = ld(x, xdt, base + c);
            const float m1 = a1 / cnt;
            mean[(long)b * C + c] = shift + m1;
            if (rstd) rstd[(long)b * C + c] = 1.0f / sqrtf(fmaxf(a2 / cnt - m1 * m1, 0.f) + eps);   // biased var (InstanceNorm1d)
        }
    }
}
// Single requests: statistics AND normalise-affine-activation of a 16-bit [L][C] tensor in one launch (one workgroup per
// (utterance, 64 channels): the same 32 row groups x 8 lanes as k_colstats, then the same threads walk the rows again).  Saves the
// second launch (~12 us of a ~25 us pair at batch 1); bit-identical to k_colstats + k_norm_affine_act (same accumulation order, same
// scale / shift expressions).  Larger batches keep the two kernels: the apply pass wants more workgroups than 17 per utterance.
__global__ __launch_bounds__(256) void k_instnorm_fused(const void* x, int xdt, int ldx, void* y, int ydt, int ldy, int Lmax, const int* L, int C, float eps,
                                                        float* mean, float* rstd, const float* gamma, const float* beta, long g_bs, int one_plus,
                                                        int act, float slope) {
    __shared__ float red[2][32][65];
    __shared__ float stat[2][64];
    const int b = blockIdx.y, cl = (threadIdx.x & 7) * 8, c0 = blockIdx.x * 64 + cl, g = threadIdx.x >> 3;
    const int Wb = L[b];
    const long base = (long)b * Lmax * ldx;
    float s1[8], s2[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { s1[e] = 0.f; s2[e] = 0.f; sh[e] = 0.f; }
    const bool cok = c0 < C;
    const unsigned short* xp = (const unsigned short*)x + base + c0;
    if (cok && Wb > 0) unpack8(*(const uint4*)xp, xdt, sh);
    if (cok)
        for (int w = g; w < Wb; w += 128) {
            uint4 t[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int wu = w + 32 * u < Wb ? w + 32 * u : w; t[u] = *(const uint4*)(xp + (long)wu * ldx); }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (w + 32 * u >= Wb) break;
                float v[8];
                unpack8(t[u], xdt, v);
#pragma unroll
                for (int e = 0; e < 8; e++) { const float d = v[e] - sh[e]; s1[e] += d; s2[e] += d * d; }
            }
        }
#pragma unroll
    for (int e = 0; e < 8; e++) { red[0][g][cl + e] = s1[e]; red[1][g][cl + e] = s2[e]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = blockIdx.x * 64 + threadIdx.x;
        float a1 = 0.f, a2 = 0.f;
        for (int i = 0; i < 32; i++) { a1 += red[0][i][threadIdx.x]; a2 += red[1][i][threadIdx.x]; }
        float m = 0.f, r = 0.f;
        if (c < C) {
            const float cnt = (float)Wb;
            const float shift = ld(x, xdt, base + c);
            const float m1 = a1 / cnt;
            m = shift + m1;
            r = 1.0f / sqrtf(fmaxf(a2 / cnt - m1 * m1, 0.f) + eps);
            mean[(long)b * C + c] = m; rstd[(long)b * C + c] = r;
        }
        stat[0][threadIdx.x] = m; stat[1][threadIdx.x] = r;
    }
    __syncthreads();
    if (!cok) return;
    float sc[8], sf[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float m = stat[0][cl + e], r = stat[1][cl + e];
        float gg = 1.f, be = 0.f;
        if (gamma) { gg = (one_plus ? 1.f : 0.f) + gamma[b * g_bs + c0 + e]; be = beta[b * g_bs + c0 + e]; }
        sc[e] = r * gg; sf[e] = be - m * r * gg;                // (x - m) * r * g + be
    }
    unsigned short* yp = (unsigned short*)y + (long)b * Lmax * ldy + c0;
    for (int w = g; w < Wb; w += 128) {
        uint4 t[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int wu = w + 32 * u < Wb ? w + 32 * u : w; t[u] = *(const uint4*)(xp + (long)wu * ldx); }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (w + 32 * u >= Wb) break;
            float v[8];
            unpack8(t[u], xdt, v);
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = act_apply(v[e] * sc[e] + sf[e], act, slope);
            *(uint4*)(yp + (long)(w + 32 * u) * ldy) = pack8(v, ydt);
        }
    }
}
void launch_instnorm_fused(const void* x, int x_dt, int ldx, void* y, int y_dt, int ldy, int B, int Lmax, const int* L, int C, float eps,
                           float* mean, float* rstd, const float* gamma, const float* beta, long g_bs, int one_plus, int act, float slope, hipStream_t s) {
    if (Lmax <= 0) return;
    hipLaunchKernelGGL(k_instnorm_fused, dim3((C + 63) / 64, B), dim3(256), 0, s, x, x_dt, ldx, y, y_dt, ldy, Lmax, L, C, eps, mean, rstd, gamma, beta,
                       g_bs, one_plus, act, slope);
}
void launch_instnorm_stats(const void* x, int x_dt, int ldx, int B, int Lmax, const int* L, int C, float eps,
                           float* mean, float* rstd, hipStream_t s) {
    hipLaunchKernelGGL(k_colstats, dim3((C + 63) / 64, B), dim3(256), 0, s, x, x_dt, ldx, 1, Lmax, L, C, eps, mean, rstd);
}
// SE global average pool over an [H][Wmax][C] map (valid columns w < W[b]): the map can be 20 000+ positions of only 32
// channels, so the rows are split over S blocks per utterance; partial[b][s][c] holds each block's plain sum and
// k_se_fc folds the S partial sums in a fixed order (deterministic) before the squeeze-excite MLP.
__global__ __launch_bounds__(256) void k_se_pool_partial(const void* x, int xdt, int H, int Wmax, const int* W, int C, int rows_per_blk, float* partial) {
    __shared__ float red[256][9];
    const int b = blockIdx.y, sblk = blockIdx.x, S = gridDim.x;
    const int lpr = C >> 3, rpp = 256 / lpr;                  // lanes per row (8 channels each), rows per pass
    const int c8 = threadIdx.x % lpr, rg = threadIdx.x / lpr;
    const int Wb = W[b], total = H * Wmax;
    const long base = (long)b * total * C;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = 0.f;
    const int r0 = sblk * rows_per_blk, r1 = min(total, r0 + rows_per_blk);
    if (rg < rpp && xdt == DT_BF16) {
        // four rows of the thread in flight per iteration (the pass is HBM-bound: one dependent 16-byte load per iteration left
        // most of the bandwidth idle), the column of a row tracked incrementally instead of one integer division per row
        const unsigned short* xp = (const unsigned short*)x + base + c8 * 8;
        int r = r0 + rg, w = r % Wmax;
        const int step_w = rpp % Wmax;
        for (; r < r1; r += 4 * rpp) {
            uint4 t[4]; bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int ru = r + u * rpp;
                ok[u] = ru < r1 && w < Wb;
                t[u] = ok[u] ? *(const uint4*)(xp + (long)ru * C) : make_uint4(0, 0, 0, 0);
                w += step_w; if (w >= Wmax) w -= Wmax;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {                                           // (same order of additions as the one-row loop: rows ascending)
                if (!ok[u]) continue;
                acc[0] += __uint_as_float(t[u].x << 16); acc[1] += __uint_as_float(t[u].x & 0xffff0000u);
                acc[2] += __uint_as_float(t[u].y << 16); acc[3] += __uint_as_float(t[u].y & 0xffff0000u);
                acc[4] += __uint_as_float(t[u].z << 16); acc[5] += __uint_as_float(t[u].z & 0xffff0000u);
                acc[6] += __uint_as_float(t[u].w << 16); acc[7] += __uint_as_float(t[u].w & 0xffff0000u);
            }
        }
    } else if (rg < rpp)
        for (int r = r0 + rg; r < r1; r += rpp) {
            if (r % Wmax >= Wb) continue;
            const long off = base + (long)r * C + c8 * 8;
            {
                const float4 t0 = *(const float4*)((const float*)x + off), t1 = *(const float4*)((const float*)x + off + 4);
                acc[0] += t0.x; acc[1] += t0.y; acc[2] += t0.z; acc[3] += t0.w; acc[4] += t1.x; acc[5] += t1.y; acc[6] += t1.z; acc[7] += t1.w;
            }
        }
#pragma unroll
    for (int e = 0; e < 8; e++) red[threadIdx.x][e] = acc[e];
    __syncthreads();
    if (threadIdx.x < C) {                                    // C <= 256
        const int cc = threadIdx.x >> 3, e = threadIdx.x & 7;
        float a = 0.f;
        for (int g = 0; g < rpp; g++) a += red[g * lpr + cc][e];
        partial[((long)b * S + sblk) * C + threadIdx.x] = a;
    }
}
int se_pool_splits(int H, int Wmax) { const int total = H * Wmax; int S = (total + 511) / 512; return S < 1 ? 1 : (S > 64 ? 64 : S); }
void launch_se_pool(const void* x, int x_dt, int B, int H, int Wmax, const int* W, int C, float* partial, hipStream_t s) {
    const int S = se_pool_splits(H, Wmax), total = H * Wmax;
    hipLaunchKernelGGL(k_se_pool_partial, dim3(S, B), dim3(256), 0, s, x, x_dt, H, Wmax, W, C, (total + S - 1) / S, partial);
}

// One wave = 64 x 8-channel vectors of a row (1 KiB of bf16); a block covers 64 rows x 512 channels, each lane keeps the
// per-(utterance, channel) scale/shift of its 8 channels in registers for all of its rows.
__global__ __launch_bounds__(256) void k_norm_affine_act(const void* x, int xdt, int ldx, void* y, int ydt, int ldy, int Lmax, const int* L, int C,
                                                         const float* mean, const float* rstd, const float* gamma, const float* beta, long g_bs,
                                                         int one_plus, int act, float slope) {
    const int b = blockIdx.z;
    const int Lb = L[b];
    const int l0 = blockIdx.y * 64;
    if (l0 >= Lb) return;
    const int c = (blockIdx.x * 64 + (threadIdx.x & 63)) * 8;
    if (c >= C) return;                                       // C % 8 == 0 for every caller
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float m = mean[(long)b * C + c + e], r = rstd[(long)b * C + c + e];
        float g = 1.f, be = 0.f;
        if (gamma) { g = (one_plus ? 1.f : 0.f) + gamma[b * g_bs + c + e]; be = beta[b * g_bs + c + e]; }
        sc[e] = r * g; sh[e] = be - m * r * g;                // (x - m) * r * g + be
    }
    const int lend = (l0 + 64 < Lb) ? l0 + 64 : Lb;
    if (xdt != DT_F32 && ydt != DT_F32) {
        // 16-bit -> 16-bit (the decoder's InstanceNorm / AdaIN passes): four rows of the wave in flight per iteration.  Neutral at batch 32
        // (the pass is bandwidth-bound there); a single request runs 16 dependent row round trips per wave otherwise.
        const unsigned short* xp = (const unsigned short*)x + (long)b * Lmax * ldx + c;
        unsigned short* yp = (unsigned short*)y + (long)b * Lmax * ldy + c;
        for (int l = l0 + (threadIdx.x >> 6); l < lend; l += 16) {
            uint4 t[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int lu = l + 4 * u < lend ? l + 4 * u : l; t[u] = *(const uint4*)(xp + (long)lu * ldx); }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (l + 4 * u >= lend) break;
                float v[8];
                unpack8(t[u], xdt, v);
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = act_apply(v[e] * sc[e] + sh[e], act, slope);
                *(uint4*)(yp + (long)(l + 4 * u) * ldy) = pack8(v, ydt);
            }
        }
        return;
    }
    for (int l = l0 + (threadIdx.x >> 6); l < lend; l += 4) {
        const long xo = ((long)b * Lmax + l) * ldx + c, yo = ((long)b * Lmax + l) * ldy + c;
        float v[8];
        if (xdt != DT_F32) {
            unpack8(*(const uint4*)((const unsigned short*)x + xo), xdt, v);
        } else {
            const float4 t0 = *(const float4*)((const float*)x + xo), t1 = *(const float4*)((const float*)x + xo + 4);
            v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
        }
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = act_apply(v[e] * sc[e] + sh[e], act, slope);
        if (ydt != DT_F32) {
            *(uint4*)((unsigned short*)y + yo) = pack8(v, ydt);
        } else {
            *(float4*)((float*)y + yo) = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)((float*)y + yo + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}
void launch_norm_affine_act(const void* x, int x_dt, int ldx, void* y, int y_dt, int ldy, int B, int Lmax,
                            const int* L, int C, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, long g_bs, int one_plus, int act, float slope, hipStream_t s) {
    if (Lmax <= 0) return;
    hipLaunchKernelGGL(k_norm_affine_act, dim3((C / 8 + 63) / 64, (Lmax + 63) / 64, B), dim3(256), 0, s, x, x_dt, ldx, y, y_dt, ldy, Lmax,
                       L, C, mean, rstd, gamma, beta, g_bs, one_plus, act, slope);
}

// ---------------------------------------------------------------- mel -> zero-padded vocoder input
__global__ void k_mel_pad(const void* mel, int mdt, int ldm, int Lmax, const int* mel_len, void* v, int vdt, int ldv, int Pmax,
                          const int* P, int nm) {
    const int b = blockIdx.y, p = blockIdx.x;
    if (p >= P[b]) return;
    const bool real = p < mel_len[b];
    for (int c = threadIdx.x; c < nm; c += blockDim.x)
        st(v, vdt, ((long)b * Pmax + p) * ldv + c, real ? ld(mel, mdt, ((long)b * Lmax + p) * ldm + c) : 0.f);
}
void launch_mel_pad(const void* mel, int m_dt, int ldm, int Lmax, const int* mel_len, void* v, int v_dt,
                    int ldv, int Pmax, const int* P, int B, int nm, hipStream_t s) {
    hipLaunchKernelGGL(k_mel_pad, dim3(Pmax, B), dim3(128), 0, s, mel, m_dt, ldm, Lmax, mel_len, v, v_dt, ldv, Pmax, P, nm);
}

__global__ void k_copy_rows_f32(const void* src, int sdt, int lds, long s_bs, float* dst, long ldd, long d_bs, int rows_max,
                                const int* rows, int C) {
    const int b = blockIdx.y, r = blockIdx.x;
    if (r >= (rows ? rows[b] : rows_max)) return;
    for (int c = threadIdx.x; c < C; c += blockDim.x) dst[b * d_bs + r * ldd + c] = ld(src, sdt, b * s_bs + (long)r * lds + c);
}
void launch_copy_rows_f32(const void* src, int s_dt, int lds, long s_bs, float* dst, long ldd, long d_bs,
                          int B, int rows_max, const int* rows, int C, hipStream_t s) {
    if (rows_max <= 0) return;
    hipLaunchKernelGGL(k_copy_rows_f32, dim3(rows_max, B), dim3(128), 0, s, src, s_dt, lds, s_bs, dst, ldd, d_bs, rows_max, rows, C);
}

// rows [rows[b], rows_max) of x [b][rows_max][ldx] (first C columns) := 0 -- outputs handed to the caller never carry
// what an earlier, longer call left in the reused buffer
__global__ void k_zero_tail_rows(float* x, int ldx, int rows_max, const int* rows, int C) {
    const int b = blockIdx.y, r = blockIdx.x;
    if (r < rows[b]) return;
    for (int c = threadIdx.x; c < C; c += blockDim.x) x[((long)b * rows_max + r) * ldx + c] = 0.f;
}
void launch_zero_tail_rows(float* x, int ldx, int B, int rows_max, const int* rows, int C, hipStream_t s) {
    if (rows_max <= 0) return;
    hipLaunchKernelGGL(k_zero_tail_rows, dim3(rows_max, B), dim3(128), 0, s, x, ldx, rows_max, rows, C);
}

// Saturation audit of the IEEE-half mode (zvx_set_int "f16_sat_check"): every 16-bit store of that mode clamps at +-65504
// (MODE.FP16_OVFL), so a clamped result IS the bit pattern 0x7BFF / 0xFBFF.  Counts those patterns (and Inf / NaN patterns, which the
// clamp should have made impossible) in the valid rows of a stored tensor x[b][r][0:C]; one 64-bit atomic per workgroup that found any.
__global__ void k_count_sat16(const unsigned short* x, long bs, int ld, int rows_max, const int* rows, int C, unsigned long long* count) {
    const int b = blockIdx.y;
    const int nr = rows ? min(rows[b], rows_max) : rows_max;
    unsigned n = 0;
    for (int r = blockIdx.x; r < nr; r += gridDim.x) {
        const unsigned short* row = x + (long)b * bs + (long)r * ld;
        for (int c = threadIdx.x; c < C; c += blockDim.x) n += (row[c] & 0x7FFF) >= 0x7BFF;
    }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
    __shared__ unsigned part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) { const unsigned t = part[0] + part[1] + part[2] + part[3]; if (t) atomicAdd(count, (unsigned long long)t); }
}
void launch_count_sat16(const void* x, long bs, int ld, int B, int rows_max, const int* rows, int C, unsigned long long* count, hipStream_t s) {
    if (B <= 0 || rows_max <= 0 || C <= 0) return;
    hipLaunchKernelGGL(k_count_sat16, dim3(rows_max < 1024 ? rows_max : 1024, B), dim3(256), 0, s, (const unsigned short*)x, bs, ld, rows_max, rows, C, count);
}

// x[b][r][c] = 0 for len[b] <= c < cols (element size es = 2 / 4 bytes): the key-contiguous V^T of the unfused attention path, whose
// columns past an utterance's length come from rows nothing has defined (0 x NaN would not be 0 in the P.V product)
__global__ void k_zero_tail_cols(unsigned char* x, int es, long ld, long bs, int rows, int cols, const int* len) {
    const int b = blockIdx.z, r = blockIdx.y;
    const int l0 = len[b];
    for (int c = l0 + blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x) {
        unsigned char* p = x + ((long)b * bs + (long)r * ld + c) * es;
        if (es == 4) *(unsigned*)p = 0u; else *(unsigned short*)p = (unsigned short)0;
    }
}
void launch_zero_tail_cols(void* x, int es, long ld, long bs, int B, int rows, int cols, const int* len, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return;
    hipLaunchKernelGGL(k_zero_tail_cols, dim3(1, rows, B), dim3(256), 0, s, (unsigned char*)x, es, ld, bs, rows, cols, len);
}

// ---------------------------------------------------------------- conv_post (C -> 1) + tanh      hifigan.py:127-128
// x is the activated last stage [b][Nmax][ldx]; one thread per output sample, weights broadcast from LDS.
// conv_post (C -> 1, k taps) + tanh: a block of 256 samples stages its (256 + k - 1) input rows in LDS once (coalesced
// 16-byte loads, padded pitch) instead of every thread pulling its k rows through L1; the accumulation order per sample
// is unchanged (tap-major, then channel groups), so results are bit-identical to the direct form.
// Samples in [out_len*out_mul, Nmax) of every row are written as zeros, so the caller's row never depends on what an
// earlier call left in a reused buffer.  pcm16: the row is int16 PCM, (short)trunc(tanh(.) * 32760) (demo.py:29-35).
template <int KT, int CC>
__global__ __launch_bounds__(256) void k_conv_post_tanh(const void* x, int xdt, int ldx, long x_bs, const float* __restrict__ w, float bias, int kt, int C,
                                 void* wav, long wav_bs, int Nmax, int pcm16, const int* in_len, int len_mul, const int* out_len, int out_mul) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int es = xdt != DT_F32 ? 2 : 4, rowb = C * es, pitch = rowb + 16, half = (kt - 1) / 2, nrows = 256 + kt - 1;
    // KT, CC > 0 (HiFi-GAN: 7 taps x 32 channels, bf16): the tap / channel loops unroll and the weights are read with uniform
    // addresses straight from `w` (scalar loads, SGPR operands) instead of 56 LDS reads per sample; otherwise staged in LDS
    float* wl = (float*)(sm + (size_t)nrows * pitch);
    if (KT == 0) for (int i = threadIdx.x; i < kt * C; i += blockDim.x) wl[i] = w[i];
    const int b = blockIdx.y;
    const long n0 = (long)blockIdx.x * 256;
    const long nin = (long)in_len[b] * len_mul, nout = (long)out_len[b] * out_mul;
    auto put = [&](long n, float v) {
        if (pcm16) ((short*)wav)[b * wav_bs + n] = (short)(v * 32760.0f);      // float -> int conversion truncates (numpy astype)
        else ((float*)wav)[b * wav_bs + n] = v;
    };
    if (n0 >= nout) {                                        // whole block past the utterance's end: zero tail
        if (n0 + threadIdx.x < Nmax) put(n0 + threadIdx.x, 0.f);
        return;
    }
    const int cpr = rowb >> 4;                               // 16-byte chunks per row
    for (int i = threadIdx.x; i < nrows * cpr; i += 256) {
        const int r = i / cpr, q = i % cpr;
        const long m = n0 + r - half;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (m >= 0 && m < nin) v = *(const uint4*)((const unsigned char*)x + (b * x_bs + m * ldx) * es + q * 16);
        *(uint4*)(sm + r * pitch + q * 16) = v;
    }
    __syncthreads();
    const long n = n0 + threadIdx.x;
    if (n >= nout) { if (n < Nmax) put(n, 0.f); return; }
    float acc = bias;
    if (KT > 0) {
#pragma unroll
        for (int k = 0; k < KT; k++) {
            const long m = n + k - half;
            if (m < 0 || m >= nin) continue;
            const unsigned char* row = sm + (threadIdx.x + k) * pitch;
#pragma unroll
            for (int c = 0; c < CC; c += 8) {
                const uint4 t = *(const uint4*)(row + c * 2);
                const float* ww = w + k * CC + c;
                float v[8];
                unpack8(t, xdt, v);                                            // bf16 or IEEE half (the vocoder's 16-bit dtype)
                acc += v[0] * ww[0] + v[1] * ww[1] + v[2] * ww[2] + v[3] * ww[3] + v[4] * ww[4] + v[5] * ww[5] + v[6] * ww[6] + v[7] * ww[7];
            }
        }
        put(n, tanhf(acc));
        return;
    }
    for (int k = 0; k < kt; k++) {
        const long m = n + k - half;
        if (m < 0 || m >= nin) continue;
        const unsigned char* row = sm + (threadIdx.x + k) * pitch;
        if (xdt != DT_F32) {
            for (int c = 0; c < C; c += 8) {
                const uint4 t = *(const uint4*)(row + c * 2);
                const float* ww = wl + k * C + c;
                float v[8];
                unpack8(t, xdt, v);
                acc += v[0] * ww[0] + v[1] * ww[1] + v[2] * ww[2] + v[3] * ww[3] + v[4] * ww[4] + v[5] * ww[5] + v[6] * ww[6] + v[7] * ww[7];
            }
        } else {
            for (int c = 0; c < C; c += 4) {
                const float4 t = *(const float4*)(row + c * 4);
                const float* ww = wl + k * C + c;
                acc += t.x * ww[0] + t.y * ww[1] + t.z * ww[2] + t.w * ww[3];
            }
        }
    }
    put(n, tanhf(acc));
}
void launch_conv_post_tanh(const void* x, int x_dt, int ldx, long x_bs, const float* w, float bias,
                           int ktaps, int C, void* wav, long wav_bs, int pcm16, int B, int Nmax, const int* in_len,
                           int len_mul, const int* out_len, int out_mul, hipStream_t s) {
    if (Nmax <= 0) return;
    const size_t es = x_dt != DT_F32 ? 2 : 4;
    const size_t lds = (size_t)(256 + ktaps - 1) * (C * es + 16) + (size_t)ktaps * C * sizeof(float);
    if (x_dt != DT_F32 && ktaps == 7 && C == 32)
        hipLaunchKernelGGL((k_conv_post_tanh<7, 32>), dim3((Nmax + 255) / 256, B), dim3(256), lds, s, x, x_dt, ldx, x_bs, w,
                           bias, ktaps, C, wav, wav_bs, Nmax, pcm16, in_len, len_mul, out_len, out_mul);
    else
        hipLaunchKernelGGL((k_conv_post_tanh<0, 0>), dim3((Nmax + 255) / 256, B), dim3(256), lds, s, x, x_dt, ldx, x_bs, w,
                           bias, ktaps, C, wav, wav_bs, Nmax, pcm16, in_len, len_mul, out_len, out_mul);
}

// ---------------------------------------------------------------- speaker encoder pieces
// first layer: InstanceNorm1d(F) over time folded in (mean/rstd given) + Conv2d(1->C0,3x3,p1) + ReLU + BN affine
// One workgroup = one clip x SF_TB consecutive time positions x ALL frequency rows.  The normalised input patch [F + 2][SF_TB + 2] is
// staged in LDS with COALESCED reads of the [time][F] log-mel (a row of F floats per time step); round 4's kernel gathered it with one
// thread per (f, t): consecutive lanes read 320 bytes apart, 4 useful bytes per 128-byte line, every frequency row's workgroups pulling
// the whole mel through L2 again -- ~3 GB of L2 traffic for a 21 MB input, and the launch wrote its 333 MB of output at 0.85 TB/s.
// A thread owns 8 output channels (72 weights + 24 bias / BN values in registers) of one time position and walks the frequency rows;
// a wave's store covers 16 consecutive positions x 64 bytes.  Same arithmetic order per output as before (bit-identical).
#define SF_TB 64
__global__ __launch_bounds__(256) void k_spk_front(const float* mels, int Tmax, const int* lens, int F, const float* mean, const float* rstd,
                            const float* w, const float* bias, const float* bs, const float* bt, int C0, void* out, int odt, int Wout) {
    extern __shared__ float patch[];                             // [F + 2][SF_TB + 2]: frequency rows -1 .. F, times t0 - 1 .. t0 + SF_TB
    constexpr int PW = SF_TB + 2;
    const int b = blockIdx.y, t0 = blockIdx.x * SF_TB, cpp = C0 >> 3;
    const int Tb = lens[b];
    if (t0 >= Tb) return;
    for (int i = threadIdx.x; i < (F + 2) * PW; i += blockDim.x) patch[i] = 0.f;
    __syncthreads();
    // coalesced: consecutive threads read consecutive frequencies of one time step
    for (int i = threadIdx.x; i < PW * F; i += blockDim.x) {
        const int j = i / F, ff = i - j * F, tt = t0 + j - 1;
        if (tt >= 0 && tt < Tb) patch[(ff + 1) * PW + j] = (mels[((long)b * Tmax + tt) * F + ff] - mean[b * F + ff]) * rstd[b * F + ff];
    }
    __syncthreads();
    for (int id = threadIdx.x; id < SF_TB * cpp; id += blockDim.x) {
        const int tl = id / cpp, c0 = (id - tl * cpp) * 8, t = t0 + tl;
        if (t >= Tb) continue;
        float wk[9][8], bb[8], sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; e++) { bb[e] = bias[c0 + e]; sc[e] = bs[c0 + e]; sh[e] = bt[c0 + e]; }
#pragma unroll
        for (int k = 0; k < 9; k++)
#pragma unroll
            for (int e = 0; e < 8; e++) wk[k][e] = w[k * C0 + c0 + e];
        float x0[3], x1[3], x2[3];                               // rows f - 1, f, f + 1 of the patch at times t - 1 .. t + 1
#pragma unroll
        for (int j = 0; j < 3; j++) { x0[j] = patch[0 * PW + tl + j]; x1[j] = patch[1 * PW + tl + j]; }
        for (int f = 0; f < F; f++) {
#pragma unroll
            for (int j = 0; j < 3; j++) x2[j] = patch[(f + 2) * PW + tl + j];
            float r[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float a = bb[e];
#pragma unroll
                for (int j = 0; j < 3; j++) a += x0[j] * wk[j][e];
#pragma unroll
                for (int j = 0; j < 3; j++) a += x1[j] * wk[3 + j][e];
#pragma unroll
                for (int j = 0; j < 3; j++) a += x2[j] * wk[6 + j][e];
                r[e] = fmaxf(a, 0.f) * sc[e] + sh[e];           // conv -> ReLU -> BN  (ResNetSE34V2.py:184-186)
            }
            const long o = (((long)b * F + f) * Wout + t) * C0 + c0;      // output map rows are Wout >= Tmax positions wide
            if (odt == DT_BF16) {
                uint4 pk;
                pk.x = tobf(r[0]) | ((unsigned)tobf(r[1]) << 16); pk.y = tobf(r[2]) | ((unsigned)tobf(r[3]) << 16);
                pk.z = tobf(r[4]) | ((unsigned)tobf(r[5]) << 16); pk.w = tobf(r[6]) | ((unsigned)tobf(r[7]) << 16);
                *(uint4*)((unsigned short*)out + o) = pk;
            } else {
                *(float4*)((float*)out + o) = make_float4(r[0], r[1], r[2], r[3]);
                *(float4*)((float*)out + o + 4) = make_float4(r[4], r[5], r[6], r[7]);
            }
#pragma unroll
            for (int j = 0; j < 3; j++) { x0[j] = x1[j]; x1[j] = x2[j]; }
        }
    }
}
void launch_spk_front(const float* mels, int Tmax, const int* lens, int F, const float* mean, const float* rstd,
                      const float* w, const float* bias, const float* bn_scale, const float* bn_shift, int C0,
                      void* out, int o_dt, int B, int Wout, hipStream_t s) {
    const size_t lds = (size_t)(F + 2) * (SF_TB + 2) * sizeof(float);                 // C0 % 8 == 0
    hipLaunchKernelGGL(k_spk_front, dim3((Tmax + SF_TB - 1) / SF_TB, B), dim3(256), lds, s, mels, Tmax, lens, F, mean, rstd, w, bias,
                       bn_scale, bn_shift, C0, out, o_dt, Wout);
}

// pool_bias (optional): the partial sums were taken BEFORE the convolution's per-channel bias (the pool fused into conv2d_persist_kernel): the
// mean over the valid positions carries it once
__global__ void k_se_fc(const float* partial, int S, int H, const int* W, const float* w1, const float* b1, const float* w2, const float* b2, int C, int Cr, float* scale,
                        const float* pool_bias) {
    extern __shared__ float hbuf[];                           // [Cr] hidden + [C] mean + [256] partial folds
    const int b = blockIdx.x;
    float* m = hbuf + Cr;
    float* red = m + C;
    const float cnt = (float)H * (float)W[b];
    // fold the S partial sums in a FIXED order (deterministic) with all 256 threads: thread (g, c) takes partials g, g + G, ... of channel
    // c (four independent loads in flight: the fused pool of the persistent convolution leaves ~200 partials per utterance, and one
    // dependent L2 round trip per partial was 60 us of a 10 us kernel), then channel c's G folds are added in order
    if (C <= (int)blockDim.x) {
        const int G = blockDim.x / C, g = threadIdx.x / C, c = threadIdx.x % C;
        if (g < G) {
            const float* pp = partial + (long)b * S * C + c;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int i = g;
            for (; i + 3 * G < S; i += 4 * G) { a0 += pp[(long)i * C]; a1 += pp[(long)(i + G) * C]; a2 += pp[(long)(i + 2 * G) * C]; a3 += pp[(long)(i + 3 * G) * C]; }
            for (; i < S; i += G) a0 += pp[(long)i * C];
            red[g * C + c] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        if (threadIdx.x < C) {
            float a = 0.f;
            for (int q = 0; q < G; q++) a += red[q * C + threadIdx.x];
            m[threadIdx.x] = a / cnt + (pool_bias ? pool_bias[threadIdx.x] : 0.f);
        }
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float a = 0.f;
            for (int i = 0; i < S; i++) a += partial[((long)b * S + i) * C + c];
            m[c] = a / cnt + (pool_bias ? pool_bias[c] : 0.f);
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < Cr; j += blockDim.x) {
        float a = b1[j];
        for (int c = 0; c < C; c++) a += w1[(long)j * C + c] * m[c];
        hbuf[j] = fmaxf(a, 0.f);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a = b2[c];
        for (int j = 0; j < Cr; j++) a += w2[(long)c * Cr + j] * hbuf[j];
        scale[(long)b * C + c] = 1.0f / (1.0f + expf(-a));
    }
}
void launch_se_fc(const float* partial, int S, int H, const int* W, const float* w1, const float* b1, const float* w2, const float* b2, int C,
                  int Cr, float* scale, int B, hipStream_t s, const float* pool_bias) {
    hipLaunchKernelGGL(k_se_fc, dim3(B), dim3(256), (Cr + C + 256) * sizeof(float), s, partial, S, H, W, w1, b1, w2, b2, C, Cr, scale, pool_bias);
}

// y = relu(x * scale[b][c] + res): 8 channels (16 bytes of bf16) per thread, grid-stride over the map's vectors
// bf16 maps: a thread owns 8 channels (its scale factors stay in registers) of four positions PP = 256 / lpr apart; all eight loads are
// in flight before the first use, 32-bit index arithmetic, one modulo per thread (the column advances incrementally)
__global__ __launch_bounds__(256) void k_se_apply_bf16(const unsigned short* x, const unsigned short* res, unsigned short* y, const float* scale, int H, int Wmax,
                                                       const int* W, int C, int lpr_shift) {
    const int b = blockIdx.y, lpr = 1 << lpr_shift, PP = 256 >> lpr_shift;
    const int total = H * Wmax, Wb = W[b];
    const int c = (threadIdx.x & (lpr - 1)) * 8;
    const long base = (long)b * total * C + c;
    const float4 s0 = *(const float4*)(scale + (long)b * C + c), s1 = *(const float4*)(scale + (long)b * C + c + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    int pos = blockIdx.x * (4 * PP) + (threadIdx.x >> lpr_shift);
    int w = pos % Wmax;
    const int step_w = PP % Wmax;
    uint4 a[4], r[4]; bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int pu = pos + u * PP;
        ok[u] = pu < total && w < Wb;
        const long o = base + (long)(ok[u] ? pu : 0) * C;
        a[u] = *(const uint4*)(x + o); r[u] = *(const uint4*)(res + o);
        w += step_w; if (w >= Wmax) w -= Wmax;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        if (!ok[u]) continue;
        const unsigned au[4] = {a[u].x, a[u].y, a[u].z, a[u].w}, ru[4] = {r[u].x, r[u].y, r[u].z, r[u].w};
        unsigned short ov[8];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            ov[2 * i] = tobf(fmaxf(__uint_as_float(au[i] << 16) * sc[2 * i] + __uint_as_float(ru[i] << 16), 0.f));
            ov[2 * i + 1] = tobf(fmaxf(__uint_as_float(au[i] & 0xffff0000u) * sc[2 * i + 1] + __uint_as_float(ru[i] & 0xffff0000u), 0.f));
        }
        uint4 out;
        out.x = ov[0] | ((unsigned)ov[1] << 16); out.y = ov[2] | ((unsigned)ov[3] << 16);
        out.z = ov[4] | ((unsigned)ov[5] << 16); out.w = ov[6] | ((unsigned)ov[7] << 16);
        *(uint4*)(y + base + (long)(pos + u * PP) * C) = out;
    }
}
__global__ __launch_bounds__(256) void k_se_apply(const void* x, const void* res, void* y, int dt, const float* scale, int H, int Wmax, const int* W, int C) {
    const int b = blockIdx.y, lpr = C >> 3;
    const long nvec = (long)H * Wmax * lpr;
    const long base = (long)b * H * Wmax * C;
    const int Wb = W[b];
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
        const long pos = v / lpr; const int c = (int)(v % lpr) * 8;
        if ((int)(pos % Wmax) >= Wb) continue;
        const long o = base + pos * C + c;
        const float4 s0 = *(const float4*)(scale + (long)b * C + c), s1 = *(const float4*)(scale + (long)b * C + c + 4);
        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        float xv[8], rv[8];
        if (dt == DT_BF16) {
            const uint4 a = *(const uint4*)((const unsigned short*)x + o), r = *(const uint4*)((const unsigned short*)res + o);
            const unsigned au[4] = {a.x, a.y, a.z, a.w}, ru[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                xv[2 * i] = __uint_as_float(au[i] << 16); xv[2 * i + 1] = __uint_as_float(au[i] & 0xffff0000u);
                rv[2 * i] = __uint_as_float(ru[i] << 16); rv[2 * i + 1] = __uint_as_float(ru[i] & 0xffff0000u);
            }
            unsigned short ov[8];
#pragma unroll
            for (int e = 0; e < 8; e++) ov[e] = tobf(fmaxf(xv[e] * sc[e] + rv[e], 0.f));
            uint4 out;
            out.x = ov[0] | ((unsigned)ov[1] << 16); out.y = ov[2] | ((unsigned)ov[3] << 16);
            out.z = ov[4] | ((unsigned)ov[5] << 16); out.w = ov[6] | ((unsigned)ov[7] << 16);
            *(uint4*)((unsigned short*)y + o) = out;
        } else {
            const float4 a0 = *(const float4*)((const float*)x + o), a1 = *(const float4*)((const float*)x + o + 4);
            const float4 r0 = *(const float4*)((const float*)res + o), r1 = *(const float4*)((const float*)res + o + 4);
            *(float4*)((float*)y + o) = make_float4(fmaxf(a0.x * sc[0] + r0.x, 0.f), fmaxf(a0.y * sc[1] + r0.y, 0.f), fmaxf(a0.z * sc[2] + r0.z, 0.f), fmaxf(a0.w * sc[3] + r0.w, 0.f));
            *(float4*)((float*)y + o + 4) = make_float4(fmaxf(a1.x * sc[4] + r1.x, 0.f), fmaxf(a1.y * sc[5] + r1.y, 0.f), fmaxf(a1.z * sc[6] + r1.z, 0.f), fmaxf(a1.w * sc[7] + r1.w, 0.f));
        }
    }
}
void launch_se_apply(const void* x, const void* res, void* y, int dt, const float* scale, int B, int H, int Wmax,
                     const int* W, int C, hipStream_t s) {
    const int lpr = C >> 3;
    if (dt == DT_BF16 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0 && (long)H * Wmax < (1l << 30)) {
        int sh = 0; while ((1 << sh) < lpr) sh++;
        const int per_blk = 4 * (256 >> sh);                                        // positions per block
        hipLaunchKernelGGL(k_se_apply_bf16, dim3((unsigned)((H * Wmax + per_blk - 1) / per_blk), B), dim3(256), 0, s, (const unsigned short*)x,
                           (const unsigned short*)res, (unsigned short*)y, scale, H, Wmax, W, C, sh);
        return;
    }
    const long nvec = (long)H * Wmax * (C >> 3);
    const long blocks = (nvec + 255) / 256;
    hipLaunchKernelGGL(k_se_apply, dim3((unsigned)(blocks < 2048 ? blocks : 2048), B), dim3(256), 0, s, x, res, y, dt, scale, H, Wmax, W, C);
}

// attentive statistics pooling: softmax over time per feature column, weighted mean / std
__global__ void k_asp_pool(const void* x, int xdt, const float* logits, int F, int Wmax, const int* W, int C, float* out, int with_std) {
    const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    const int D = F * C;
    if (j >= D) return;
    const int f = j / C, c = j - f * C, Wb = W[b];
    float m = -INFINITY;
    for (int t = 0; t < Wb; t++) m = fmaxf(m, logits[((long)b * Wmax + t) * D + j]);
    float s = 0.f, sx = 0.f, sxx = 0.f;
    for (int t = 0; t < Wb; t++) {
        const float e = expf(logits[((long)b * Wmax + t) * D + j] - m);
        const float v = ld(x, xdt, (((long)b * F + f) * Wmax + t) * C + c);
        s += e; sx += e * v; sxx += e * v * v;
    }
    const float mu = sx / s;
    const float sg = sqrtf(fmaxf(sxx / s - mu * mu, 1e-5f));      // ResNetSE34V2.py:204 clamp(min=1e-5)
    if (with_std) { out[(long)b * 2 * D + j] = mu; out[(long)b * 2 * D + D + j] = sg; }      // ASP: [mu | sg]
    else out[(long)b * D + j] = mu;                                                              // SAP: the weighted mean only
}
void launch_asp_pool(const void* x, int x_dt, const float* logits, int B, int F, int Wmax, const int* W, int C,
                     float* out, int with_std, hipStream_t s) {
    hipLaunchKernelGGL(k_asp_pool, dim3((F * C + 255) / 256, B), dim3(256), 0, s, x, x_dt, logits, F, Wmax, W, C, out, with_std);
}

__global__ void k_l2norm_rows(float* x, int C) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float* row = x + (long)b * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += row[c] * row[c];
    s = wave_sum(s);
    const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);              // F.normalize eps
    for (int c = lane; c < C; c += 64) row[c] *= inv;
}
void launch_l2norm_rows(float* x, int B, int C, hipStream_t s) { hipLaunchKernelGGL(k_l2norm_rows, dim3(B), dim3(64), 0, s, x, C); }

// ------------------------------------------------------------------------------------------------
// log-mel front end (mels.py:357-395): reflect padding, |STFT| from the DFT GEMM's (re, im) columns, log(clip)
// ------------------------------------------------------------------------------------------------
__global__ void k_reflect_pad(const float* wav, long w_bs, const int* n, float* out, long o_bs, int pad, int out_cols) {
    const int b = blockIdx.y, nb = n[b];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < out_cols; i += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < nb + 2 * pad) {
            int src = i - pad;
            if (src < 0) src = -src;
            if (src >= nb) src = 2 * (nb - 1) - src;
            v = wav[b * w_bs + src];
        }
        out[b * o_bs + i] = v;
    }
}
void launch_reflect_pad(const float* wav, long w_bs, const int* n, float* out, long o_bs, int pad, int B, int out_cols, hipStream_t s) {
    dim3 grid((out_cols + 255) / 256 < 1024 ? (out_cols + 255) / 256 : 1024, B);
    hipLaunchKernelGGL(k_reflect_pad, grid, dim3(256), 0, s, wav, w_bs, n, out, o_bs, pad, out_cols);
}

__global__ void k_stft_mag(const float* spec, int lds_, float* mag, int ldm, int nf, int Tmax, const int* frames) {
    const int b = blockIdx.z, t = blockIdx.y;
    const bool live = t < frames[b];
    const float* sp = spec + ((long)b * Tmax + t) * lds_;
    float* mp = mag + ((long)b * Tmax + t) * ldm;
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < ldm; f += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (live && f < nf) { const float re = sp[f], im = sp[nf + f]; v = sqrtf(re * re + im * im); }
        mp[f] = v;
    }
}
void launch_stft_mag(const float* spec, int lds_, float* mag, int ldm, int nf, int B, int Tmax, const int* frames, hipStream_t s) {
    hipLaunchKernelGGL(k_stft_mag, dim3((ldm + 255) / 256, Tmax, B), dim3(256), 0, s, spec, lds_, mag, ldm, nf, Tmax, frames);
}

__global__ void k_log_clip(float* x, int ldx, int C, float lo, int Tmax, const int* frames) {
    const int b = blockIdx.y;
    const long total = (long)Tmax * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i / C), c = (int)(i % C);
        float* p = x + ((long)b * Tmax + t) * ldx + c;
        *p = t < frames[b] ? logf(fmaxf(*p, lo)) : 0.f;
    }
}
void launch_log_clip(float* x, int ldx, int C, float lo, int B, int Tmax, const int* frames, hipStream_t s) {
    const long total = (long)Tmax * C;
    hipLaunchKernelGGL(k_log_clip, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096), B), dim3(256), 0, s, x, ldx, C, lo, Tmax, frames);
}

// ------------------------------------------------------------------------------------------------
// out[b][n] = bias[n] + dot(x[b][0:K], w[n][0:K]) for a handful of rows b and a long K (speaker-encoder head: 50 x 5120 -> 528):
// one wave per (output column, block of 16 rows), the K axis spread over the lanes (float4).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_fc_rows(const float* x, int ldx, const float* w, int ldw, const float* bias, float* out, int ldo, int B, int N, int K) {
    const int n = blockIdx.x, b0 = blockIdx.y * 16, lane = threadIdx.x;
    const float* wn = w + (long)n * ldw;
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {                 // K % 4 == 0
        const float4 wv = *(const float4*)(wn + k);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int b = b0 + r < B ? b0 + r : B - 1;        // clamp: branch-free loads, surplus rows discarded below
            const float4 xv = *(const float4*)(x + (long)b * ldx + k);
            acc[r] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const float t = wave_sum(acc[r]);
        if (lane == 0 && b0 + r < B) out[(long)(b0 + r) * ldo + n] = t + (bias ? bias[n] : 0.f);
    }
}
// Many rows (config 5: 250 clips per call): FOUR output columns per wave -- the 16 rows' x vectors are fetched once per four columns (the
// one-column kernel re-read them per column: 2.8 GB through L1 / L2 for a 5 MB operand, 115 us).  Same additions in the same order per
// output: bit-identical to k_fc_rows.
__global__ __launch_bounds__(64) void k_fc_rows4(const float* x, int ldx, const float* w, int ldw, const float* bias, float* out, int ldo, int B, int N, int K) {
    const int n0 = blockIdx.x * 4, b0 = blockIdx.y * 16, lane = threadIdx.x;
    const float* wn[4];
#pragma unroll
    for (int c = 0; c < 4; c++) wn[c] = w + (long)(n0 + c < N ? n0 + c : N - 1) * ldw;
    float acc[4][16];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        float4 wv[4];
#pragma unroll
        for (int c = 0; c < 4; c++) wv[c] = *(const float4*)(wn[c] + k);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int b = b0 + r < B ? b0 + r : B - 1;
            const float4 xv = *(const float4*)(x + (long)b * ldx + k);
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c][r] += xv.x * wv[c].x + xv.y * wv[c].y + xv.z * wv[c].z + xv.w * wv[c].w;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float t = wave_sum(acc[c][r]);
            if (lane == 0 && b0 + r < B && n0 + c < N) out[(long)(b0 + r) * ldo + n0 + c] = t + (bias ? bias[n0 + c] : 0.f);
        }
}
void launch_fc_rows(const float* x, int ldx, const float* w, int ldw, const float* bias, float* out, int ldo, int B, int N, int K, hipStream_t s) {
    if (B > 32) hipLaunchKernelGGL(k_fc_rows4, dim3((N + 3) / 4, (B + 15) / 16), dim3(64), 0, s, x, ldx, w, ldw, bias, out, ldo, B, N, K);
    else hipLaunchKernelGGL(k_fc_rows, dim3(N, (B + 15) / 16), dim3(64), 0, s, x, ldx, w, ldw, bias, out, ldo, B, N, K);
}

}  // namespace zvx
