// rsx_bench.hip -- stand-alone timing harness for the resstream cut-out fork (development aid; results of -DRS_EXP != 0 builds are wrong by design).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -I zerovox_amd/csrc [-DRS_EXP=<mask>] tools/micro/rsx_bench.hip -o rsx_bench
//   rsx_bench <C 32|64> <ntaps> <npair> <am 0..3> [B=32] [M=229376 (C=32) / 114688 (C=64)] [iters=20]
//   (the launch sets of the V1 vocoder: C=32: npair 3, am 2 / 3 / 1(+out) for k = 3 / 7 / 11;  C=64: k=3 npair 3 am 2; k=7/11: npair 2 am 0 (out), then npair 1 am 3 / 1)
#include "resstream_exp.hip"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
namespace zvx { size_t packed_weight_elems(int ntaps, int N, int K) { const int nkc = (K + 63) / 64, nt = (N + 31) / 32; return (size_t)nt * nkc * ntaps * 4 * 64 * 8; } }
using namespace zvx;
static unsigned short rbf(unsigned& s, float scale) {
    s = s * 1664525u + 1013904223u;
    float f = ((int)(s >> 8) % 2001 - 1000) * 0.001f * scale;
    unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16);
}
int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 32, nt = argc > 2 ? atoi(argv[2]) : 11, np = argc > 3 ? atoi(argv[3]) : 3, am = argc > 4 ? atoi(argv[4]) : 1;
    const int B = argc > 5 ? atoi(argv[5]) : 32, M = argc > 6 ? atoi(argv[6]) : (C == 32 ? 229376 : 114688), iters = argc > 7 ? atoi(argv[7]) : 20;
    const size_t nel = (size_t)B * M * C, nw = packed_weight_elems(nt, C, C);
    std::vector<unsigned short> hx(nel), hw(nw);
    unsigned seed = 12345;
    for (auto& v : hx) v = rbf(seed, 1.0f);
    for (auto& v : hw) v = rbf(seed, 0.05f);
    std::vector<float> hb(C, 0.01f);
    unsigned short *X, *O, *XS, *W; float* bias; int* len;
    hipMalloc(&X, nel * 2); hipMalloc(&O, nel * 2); hipMalloc(&XS, nel * 2); hipMalloc(&W, nw * 2); hipMalloc(&bias, C * 4); hipMalloc(&len, B * 4);
    hipMemcpy(X, hx.data(), nel * 2, hipMemcpyHostToDevice); hipMemcpy(XS, hx.data(), nel * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), nw * 2, hipMemcpyHostToDevice); hipMemcpy(bias, hb.data(), C * 4, hipMemcpyHostToDevice);
    std::vector<int> hl(B, M); hipMemcpy(len, hl.data(), B * 4, hipMemcpyHostToDevice);
    StreamArgs a; memset(&a, 0, sizeof a);
    a.X = X; a.x_bs = (long)M * C; a.ldx = C; a.C = C; a.ntaps = nt; a.npair = np;
    const int d3[3] = {1, 3, 5};
    for (int p = 0; p < np; p++) { a.W1[p] = W; a.W2[p] = W; a.b1[p] = bias; a.b2[p] = bias; a.dil[p] = np == 1 ? 5 : d3[p]; }
    a.o_bs = (long)M * C; a.ldo = C; a.a_bs = (long)M * C; a.lda = C; a.accum_mode = am;
    if (am == 0 || am == 1) a.out = O;
    if (am) a.accum = XS;
    a.slope1 = 0.1f; a.res_inv_slope = 10.f; a.out_scale = am == 1 ? 1.f / 3 : 1.f; a.slope = 0.1f; a.len = len; a.M = M; a.nbatch = B; a.opt = 3;
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) if (launch_resstream(a, st, false) < 0) { printf("rejected\n"); return 1; }
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; i++) launch_resstream(a, st, false);
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    const double fl = 2.0 * 2.0 * np * (double)B * M * C * C * nt;
    printf("RS_EXP=%d C=%d k=%d pairs=%d am=%d B=%d M=%d: %.4f ms  %.1f TF/s  err=%s\n", RS_EXP, C, nt, np, am, B, M, ms, fl / ms / 1e9, hipGetErrorString(hipGetLastError()));
    return 0;
}
