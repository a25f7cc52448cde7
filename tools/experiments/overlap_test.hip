// dev experiment: can the epilogue's memory traffic hide behind the MFMA loop under the power cap?
// conv-slab kernel with its epilogue cut out (-DZVX_EXP=513: no epilogue, one workgroup per CU) on stream A, a plain copy
// kernel moving the epilogue's bytes (residual read + output write) on stream B: alone, alone, concurrently.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "zvx_kernels.h"
using namespace zvx;
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
int main(int argc, char** argv) {
    const int B = 32, M = 57344, C = 128, taps = argc > 1 ? atoi(argv[1]) : 11, dil = 5, copy_wgs = argc > 2 ? atoi(argv[2]) : 512;
    const size_t nb = (size_t)B * M * C * 2;
    std::vector<unsigned short> hx((size_t)B * M * C), hw((size_t)taps * C * C);
    srand(1);
    for (auto& v : hx) v = f2bf((float)rand() / RAND_MAX * 2.f - 1.f);
    for (auto& v : hw) v = f2bf(((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f);
    unsigned short *dx, *dw, *dwp, *dout, *dres, *dres2; float* db; int* dlen;
    hipMalloc(&dx, nb); hipMalloc(&dout, nb); hipMalloc(&dres, nb); hipMalloc(&dres2, nb);
    hipMalloc(&dw, hw.size() * 2); hipMalloc(&dwp, packed_weight_elems(taps, C, C) * 2); hipMalloc(&db, C * 4); hipMalloc(&dlen, B * 4);
    hipMemcpy(dx, hx.data(), nb, hipMemcpyHostToDevice); hipMemcpy(dres, hx.data(), nb, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice); hipMemset(db, 0, C * 4);
    std::vector<int> len(B, M); hipMemcpy(dlen, len.data(), B * 4, hipMemcpyHostToDevice);
    launch_pack_weights(dw, taps, C, C, dwp, 0);
    GemmArgs a; memset(&a, 0, sizeof a);
    a.X = dx; a.x_bs = (long)M * C; a.ldx = C; a.Wp = dwp; a.W = dw; a.w_ts = (long)C * C; a.ldw = C; a.dtype = DT_BF16;
    a.M = M; a.N = C; a.K = C; a.nbatch = B; a.nheads = 1; a.in_len = dlen; a.out_len = dlen; a.in_len_static = M;
    a.ntaps = taps; for (int i = 0; i < taps; i++) a.dv[i] = (i - taps / 2) * dil;
    a.stride = 1; a.alpha = 1.f; a.bias = db; a.bias_mode = 1; a.out_scale = 1.f; a.act = ACT_LRELU; a.slope = 0.1f;
    a.out = dout; a.o_bs = (long)M * C; a.ldo = C; a.out_dtype = DT_BF16;
    a.res = dres; a.r_bs = (long)M * C; a.ldr = C; a.res_dtype = DT_BF16; a.res_mode = 2; a.res_inv_slope = 10.f;
    gemm_enable_convtile(0);
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int NIT = 60; float ms;
    auto conv = [&]() { launch_gemm(a, sa); };
    auto copy = [&]() { hipLaunchKernelGGL(k_copy, dim3(copy_wgs), dim3(256), 0, sb, (const uint4*)dres, (uint4*)dres2, nb / 16); };
    for (int i = 0; i < 150; i++) { conv(); copy(); } hipDeviceSynchronize();
    hipEventRecord(e0, sa); for (int i = 0; i < NIT; i++) conv(); hipEventRecord(e1, sa); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("conv alone (variant %s)     %8.3f ms\n", gemm_variant_name(gemm_variant_of(a)), ms / NIT);
    hipEventRecord(e0, sb); for (int i = 0; i < NIT; i++) copy(); hipEventRecord(e1, sb); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("copy alone (%zu MB read + %zu MB write, %d workgroups) %8.3f ms = %.2f TB/s\n", nb >> 20, nb >> 20, copy_wgs, ms / NIT, 2.0 * nb / (ms / NIT) / 1e9);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);                       // legacy stream: after everything before it
    for (int i = 0; i < NIT; i++) { conv(); copy(); }
    hipStreamSynchronize(sa); hipStreamSynchronize(sb);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("both streams together       %8.3f ms per pair\n", ms / NIT);
    return 0;
}
