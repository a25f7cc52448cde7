// micro-benchmark 2: what slows the conv epilogue's stores?  Start from the fast pattern-1 store kernel (storepat.hip) and add
// the conv kernel's features one by one: V = 250 live VGPRs, L = per-wave LDS transposes before the stores, D = delay loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <bool BIGV, bool LDST, int NMFMA>
__global__ __launch_bounds__(256, 2) void k(uint4* out, const float* seed) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* tile = (char*)out + (size_t)blockIdx.x * 65536;
    f32x16 acc[8];
    if (BIGV || NMFMA) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][e] = seed[(i * 16 + e + lane) & 255];
    }
    if (NMFMA) {
        uint4 a4 = make_uint4(lane, 1, 2, 3), b4 = make_uint4(4, 5, lane, 7);
        for (int it = 0; it < NMFMA; it++)
#pragma unroll
            for (int i = 0; i < 8; i++)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x8, b4), acc[i], 0, 0, 0);
    }
    uint4 pk[16];
    char* stage = sm + wave * 8704;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (LDST) {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int g = 0; g < 4; g++)
                    *(float4*)(stage + (lane & 31) * 272 + (i * 32 + 8 * g + 4 * (lane >> 5)) * 4) =
                        (BIGV || NMFMA) ? make_float4(acc[i * 4 + j][4 * g], acc[i * 4 + j][4 * g + 1], acc[i * 4 + j][4 * g + 2], acc[i * 4 + j][4 * g + 3])
                                        : make_float4(lane, j, g, i);
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
#pragma unroll
        for (int p = 0; p < 4; p++) {
            float4 v0, v1;
            if (LDST) { const int rl = p * 8 + (lane >> 3); v0 = *(const float4*)(stage + rl * 272 + (lane & 7) * 32); v1 = *(const float4*)(stage + rl * 272 + (lane & 7) * 32 + 16); }
            else { v0 = make_float4(lane, j, p, 1); v1 = v0; }
            pk[j * 4 + p] = make_uint4(__float_as_uint(v0.x + v1.x), __float_as_uint(v0.y + v1.y), __float_as_uint(v0.z), __float_as_uint(v1.w));
        }
        if (LDST) __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    const int wr = wave & 1, wc = wave >> 1;
#pragma unroll
    for (int it = 0; it < 16; it++) {
        const int row = wr * 128 + it * 8 + (lane >> 3);
        *(uint4*)(tile + (size_t)row * 256 + wc * 128 + (lane & 7) * 16) = pk[it];
    }
}

template <bool BIGV, bool LDST, int NMFMA>
void run(const char* name, uint4* b, const float* seed, size_t bytes, hipEvent_t e0, hipEvent_t e1) {
    const int grid = (int)(bytes / 65536);
    for (int rep = 0; rep < 2; rep++) {
        CHK(hipEventRecord(e0));
        for (int it = 0; it < 10; it++) hipLaunchKernelGGL((k<BIGV, LDST, NMFMA>), dim3(grid), dim3(256), 76 * 1024, 0, b, seed);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("%-44s %.3f ms/launch -> %.2f TB/s\n", name, ms / 10, bytes / (ms / 10 * 1e-3) / 1e12);
    }
}

int main() {
    const size_t bytes = 470ull << 20;
    uint4* b; float* seed;
    CHK(hipMalloc(&b, bytes)); CHK(hipMemset(b, 0, bytes)); CHK(hipMalloc(&seed, 1024)); CHK(hipMemset(seed, 0, 1024));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    run<false, false, 0>("stores only (76 KiB LDS, 2 WG/CU)", b, seed, bytes, e0, e1);
    run<true, false, 0>("+ 128 live accumulator VGPRs", b, seed, bytes, e0, e1);
    run<false, true, 0>("+ LDS transposes", b, seed, bytes, e0, e1);
    run<true, true, 0>("+ both", b, seed, bytes, e0, e1);
    run<true, true, 24>("+ both + 192 MFMAs per wave before", b, seed, bytes, e0, e1);
    run<true, true, 96>("+ both + 768 MFMAs per wave before", b, seed, bytes, e0, e1);
    return 0;
}
