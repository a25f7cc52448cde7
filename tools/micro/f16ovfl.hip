// Does MODE.FP16_OVFL (bit 23) make f32 -> f16 converts saturate to +-65504 instead of +-Inf on gfx950?
//   hipcc --offload-arch=gfx950 -O2 tools/micro/f16ovfl.hip -o tools/micro/f16ovfl && tools/micro/f16ovfl
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out, int ovfl) {
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    const int i = threadIdx.x;
    float a = in[2 * i], b = in[2 * i + 1];
    h2 v;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v) : "v"(a), "v"(b));      // gfx950: packed RNE convert
    out[i] = __builtin_bit_cast(unsigned, v);
    h2 w = {(_Float16)a, (_Float16)b};                                              // whatever hipcc picks
    out[64 + i] = __builtin_bit_cast(unsigned, w);
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0");
}
int main() {
    float h[128]; for (int i = 0; i < 128; i++) h[i] = 0.f;
    const float vals[] = {1.f, 65504.f, 65520.f, 70000.f, 1e6f, -1e6f, 3e38f, __builtin_inff(), -__builtin_inff(), __builtin_nanf(""), 65519.f, -65536.f};
    for (int i = 0; i < 12; i++) h[i] = vals[i];
    float* d; unsigned* o; hipMalloc(&d, sizeof h); hipMalloc(&o, 128 * 4);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int ov = 0; ov < 2; ov++) {
        hipLaunchKernelGGL(k, 1, 64, 0, 0, d, o, ov);
        unsigned r[128]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d\n", ov);
        for (int i = 0; i < 6; i++) printf("  in %12g %12g -> pk %04x %04x   cast %04x %04x\n", h[2 * i], h[2 * i + 1], r[i] & 0xffff, r[i] >> 16, r[64 + i] & 0xffff, r[64 + i] >> 16);
    }
    return 0;
}
