// mfma_util.h -- small device helpers shared by the MFMA kernels of libzvx (bf16 packing, packed-f32 leaky-relu).
#pragma once
#include <hip/hip_runtime.h>

namespace zvx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));      // float pairs: v_pk_add_f32 / v_pk_mul_f32

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};                 // v_cvt_pk_bf16_f32 on gfx950 (round-to-nearest-even)
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ f32x2 unpack_bf16x2(unsigned u) { return (f32x2){__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
__device__ __forceinline__ f32x2 lrelu2(f32x2 v, float slope) { const f32x2 m = v * slope; return (f32x2){fmaxf(v.x, m.x), fmaxf(v.y, m.y)}; }     // 0 <= slope <= 1
__device__ __forceinline__ f32x2 inv_lrelu2(f32x2 y, float inv_slope) { const f32x2 m = y * inv_slope; return (f32x2){fminf(y.x, m.x), fminf(y.y, m.y)}; }  // inv_slope >= 1

// ---- IEEE-half tensors (round 5: the HiFi-GAN vocoder; round 3: the mel decoders).  Same MFMA rate as bf16, 11 significand bits
//      instead of 8.  Stores must not overflow to Inf: a kernel whose 16-bit tensors are half sets MODE.FP16_OVFL once
//      (f16_saturate_mode), after which every f32 -> f16 convert clamps to +-65504 while true Inf / NaN pass through (probed on the
//      hardware: tools/micro/f16ovfl.hip) -- the saturation costs no instruction.  Kernels that have not set the mode use pack_f16x2_sat.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void f16_saturate_mode() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }
__device__ __forceinline__ unsigned pack_f16x2_raw(float lo, float hi) { const f16x2_t v = {(_Float16)lo, (_Float16)hi}; return __builtin_bit_cast(unsigned, v); }   // v_cvt_pk_f16_f32 (RNE)
__device__ __forceinline__ unsigned pack_f16x2_sat(float lo, float hi) { return pack_f16x2_raw(__builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f)); }
__device__ __forceinline__ f32x2 unpack_f16x2(unsigned u) { const f16x2_t v = __builtin_bit_cast(f16x2_t, u); return (f32x2){(float)v.x, (float)v.y}; }
// H = true: IEEE half (the kernel runs in f16_saturate_mode), false: bf16
template <bool H> __device__ __forceinline__ unsigned pack16(float lo, float hi) { if constexpr (H) return pack_f16x2_raw(lo, hi); else return pack_bf16x2(lo, hi); }
template <bool H> __device__ __forceinline__ f32x2 unpack16(unsigned u) { if constexpr (H) return unpack_f16x2(u); else return unpack_bf16x2(u); }
template <bool H> __device__ __forceinline__ f32x16 mfma16(const uint4& a, const uint4& b, const f32x16& c) {
    if constexpr (H) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <bool H> __device__ __forceinline__ f32x16 mfma16(const u32x4& a, const uint4& b, const f32x16& c) { return mfma16<H>(__builtin_bit_cast(uint4, a), b, c); }

}  // namespace zvx
