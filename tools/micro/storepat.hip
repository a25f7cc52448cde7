// micro-benchmark: store patterns of the conv epilogue.  Each workgroup (4 waves) writes a 256-row x 256-byte tile (64 KiB).
//   mode 0: every wave instruction writes 1 KiB contiguous (4 full rows)                      -- "full rows"
//   mode 1: every wave instruction writes 8 half-rows (128 B at a 256-B stride); waves 0/1 write the low half of the
//           rows, waves 2/3 the high half                                                     -- current epilogue, BN=128
//   mode 2: like 1, but the two half-row writers are the same wave in consecutive instructions
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(uint4* out, int lds_pad) {
    extern __shared__ char sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* tile = (char*)out + (size_t)blockIdx.x * 65536;
    uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
    if (lds_pad < 0) sm[threadIdx.x] = 1;
#pragma unroll
    for (int it = 0; it < 16; it++) {
        size_t off;
        if (MODE == 0) off = (size_t)(wave * 16 + it) * 1024 + lane * 16;                                   // rows wave*64 + it*4 .. +3
        else if (MODE == 1) { const int wr = wave & 1, wc = wave >> 1; const int row = wr * 128 + it * 8 + (lane >> 3); off = (size_t)row * 256 + wc * 128 + (lane & 7) * 16; }
        else { const int row = wave * 64 + (it >> 1) * 8 + (lane >> 3); off = (size_t)row * 256 + (it & 1) * 128 + (lane & 7) * 16; }
        *(uint4*)(tile + off) = v;
    }
}

int main() {
    const size_t bytes = 470ull << 20;
    const int grid = (int)(bytes / 65536);
    uint4* b; CHK(hipMalloc(&b, bytes)); CHK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int lds : {0, 76 * 1024}) {          // 76 KiB of dynamic LDS -> only 2 workgroups per CU, like the conv kernel
        for (int mode = 0; mode < 3; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                CHK(hipEventRecord(e0));
                for (int it = 0; it < 10; it++) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), lds, 0, b, lds);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), lds, 0, b, lds);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), lds, 0, b, lds);
                }
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("lds %6d B/workgroup, pattern %d: %.3f ms/launch -> %.2f TB/s\n", lds, mode, ms / 10, bytes / (ms / 10 * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
