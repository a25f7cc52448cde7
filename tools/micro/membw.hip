// micro-benchmark: HBM write / read / copy bandwidth with the conv epilogue's access pattern (16 B per lane, rows of 256 B)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>   // 0 write, 1 read, 2 copy
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ in, uint4* __restrict__ out, long n16, int per_block) {
    long base = (long)blockIdx.x * per_block;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int i = threadIdx.x; i < per_block; i += 256) {
        long idx = base + i;
        if (idx >= n16) break;
        if (MODE == 0) out[idx] = make_uint4(i, i, i, i);
        if (MODE == 1) { uint4 v = in[idx]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
        if (MODE == 2) out[idx] = in[idx];
    }
    if (MODE == 1 && acc.x == 0x12345678) out[0] = acc;
}

int main() {
    const long bytes = 470L << 20, n16 = bytes / 16;
    uint4 *a, *b;
    CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes));
    CHK(hipMemset(a, 1, bytes)); CHK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int per_block : {4096, 16384, 65536}) {          // 64 KiB, 256 KiB, 1 MiB per workgroup
        int grid = (int)((n16 + per_block - 1) / per_block);
        for (int mode = 0; mode < 3; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                CHK(hipEventRecord(e0));
                for (int it = 0; it < 10; it++) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, a, b, n16, per_block);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, a, b, n16, per_block);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, a, b, n16, per_block);
                }
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("per_block %7d B mode %s: %.3f ms/launch -> %.2f TB/s (%s bytes counted)\n", per_block * 16,
                                mode == 0 ? "write" : mode == 1 ? "read " : "copy ", ms / 10,
                                (mode == 2 ? 2.0 : 1.0) * bytes / (ms / 10 * 1e-3) / 1e12, mode == 2 ? "read+write" : "one-way");
            }
        }
    }
    return 0;
}
