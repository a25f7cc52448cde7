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

}  // namespace zvx
