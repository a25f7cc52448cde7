// micro test: does LDS-DMA (buffer_load_dwordx4 ... lds) reach LDS offsets >= 64 KiB through M0 on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) int i32x4;
__global__ void k(const unsigned* src, unsigned* out, int lds_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 40960; i += 64) ((unsigned*)lds)[i] = 0xdeadbeefu;     // 160 KiB
    __syncthreads();
    const unsigned long long pa = (unsigned long long)src;
    const i32x4 rsrc = {__builtin_amdgcn_readfirstlane((int)(unsigned)pa), __builtin_amdgcn_readfirstlane((int)((pa >> 32) & 0xffff)), 1024, 0x00020000};
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned la = __builtin_amdgcn_readfirstlane(base + lds_off);
    const int voff = lane * 16;
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(la), "v"(voff), "s"(rsrc) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // report: where did word 0 of lane 0 (value 0x1000) land?
    int found = -1, cnt = 0;
    for (int i = 0; i < 40960; i++) if (((unsigned*)lds)[i] != 0xdeadbeefu) { if (found < 0) found = i; cnt++; }
    if (lane == 0) { out[0] = found * 4; out[1] = cnt; out[2] = ((unsigned*)lds)[lds_off / 4]; }
}
int main() {
    unsigned h[256]; for (int i = 0; i < 256; i++) h[i] = 0x1000 + i;
    unsigned *d, *o; hipMalloc(&d, 1024); hipMalloc(&o, 64); hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int off : {0, 32768, 65536 - 1024, 65536, 70 * 1024, 100 * 1024, 158 * 1024}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, d, o, off);
        unsigned r[3]; hipMemcpy(r, o, 12, hipMemcpyDeviceToHost);
        printf("lds_off %6d: first changed byte offset %6d, changed words %d, word at target 0x%x  %s\n", off, (int)r[0], r[1], r[2], (int)r[0] == off && r[1] == 256 ? "OK" : "MISPLACED");
    }
    return 0;
}
