// ps_bench.hip -- stand-alone timing harness for pairstream.hip (development aid; results of -DPS_EXP != 0 builds are wrong by design).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -I zerovox_amd/csrc [-DPS_EXP=<mask>] [-DPS_PROFILE] tools/micro/ps_bench.hip -o ps_bench
//   ps_bench <ntaps> <dil> [B=32] [M=57344] [iters=40] [am=0]
#include "pairstream_exp.hip"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
using namespace zvx;
static unsigned short rbf(unsigned& s, float scale) {
    s = s * 1664525u + 1013904223u;
    float f = ((int)(s >> 8) % 2001 - 1000) * 0.001f * scale;
    unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16);
}
int main(int argc, char** argv) {
    const int nt = argc > 1 ? atoi(argv[1]) : 11, dil = argc > 2 ? atoi(argv[2]) : 1, B = argc > 3 ? atoi(argv[3]) : 32, M = argc > 4 ? atoi(argv[4]) : 57344;
    const int iters = argc > 5 ? atoi(argv[5]) : 40, am = argc > 6 ? atoi(argv[6]) : 0;
    const int C = 128;
    const size_t nel = (size_t)B * M * C, nw = (size_t)4 * 2 * nt * 4 * 64 * 8;
    std::vector<unsigned short> hx(nel), hw(nw);
    unsigned seed = 12345;
    for (auto& v : hx) v = rbf(seed, 1.0f);
    for (auto& v : hw) v = rbf(seed, 0.05f);
    std::vector<float> hb(256, 0.01f);
    unsigned short *X, *O, *XS, *W1, *W2; float* bias; int* len; long long* prof;
    hipMalloc(&X, nel * 2); hipMalloc(&O, nel * 2); hipMalloc(&XS, nel * 2); hipMalloc(&W1, nw * 2); hipMalloc(&W2, nw * 2); hipMalloc(&bias, 1024); hipMalloc(&len, B * 4); hipMalloc(&prof, 1024);
    hipMemcpy(X, hx.data(), nel * 2, hipMemcpyHostToDevice); hipMemcpy(XS, hx.data(), nel * 2, hipMemcpyHostToDevice);
    hipMemcpy(W1, hw.data(), nw * 2, hipMemcpyHostToDevice); hipMemcpy(W2, hw.data(), nw * 2, hipMemcpyHostToDevice);
    hipMemcpy(bias, hb.data(), 1024, hipMemcpyHostToDevice); hipMemset(prof, 0, 1024);
    std::vector<int> hl(B, M); hipMemcpy(len, hl.data(), B * 4, hipMemcpyHostToDevice);
    PairArgs p; memset(&p, 0, sizeof p);
    p.X = X; p.x_bs = (long)M * C; p.ldx = C; p.W1 = W1; p.W2 = W2; p.b1 = bias; p.b2 = bias + 128; p.C = C; p.ntaps = nt; p.dil = dil;
    p.o_bs = (long)M * C; p.ldo = C; p.a_bs = (long)M * C; p.lda = C; p.accum_mode = am;
    if (am == 0 || am == 1) p.out = O;
    if (am) p.accum = XS;
    p.slope1 = 0.1f; p.res_inv_slope = 10.f; p.out_scale = am == 1 ? 1.f / 3 : 1.f; p.slope = 0.1f; p.len = len; p.M = M; p.nbatch = B; p.prof = prof;
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; i++) if (!launch_pairstream(p, st, false, nullptr, nullptr)) { printf("rejected\n"); return 1; }
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; i++) launch_pairstream(p, st, false, nullptr, nullptr);
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    const double fl = 2.0 * 2.0 * (double)B * M * C * C * nt;
    printf("EXP=%d k=%d d=%d B=%d M=%d am=%d: %.4f ms  %.1f TF/s  err=%s\n", PS_EXP, nt, dil, B, M, am, ms, fl / ms / 1e9, hipGetErrorString(hipGetLastError()));
#ifdef PS_PROFILE
    long long hp[128]; hipMemcpy(hp, prof, 1024, hipMemcpyDeviceToHost);
    printf("   wave 0 total %.1f k cycles -> %.3f GHz effective, %.1f k cycles per 128-row step\n", (hp[0] + hp[1] + hp[2]) / 1e3, (hp[0] + hp[1] + hp[2]) / (ms * 1e6), (hp[0] + hp[1] + hp[2]) / 1e3 / ((M * (double)B / 256 + 10) / 128 + 2));
    for (int w = 0; w < 8; w += 4) printf("   wave %d (role %d ct %d): main %8.1f k  epilogue %8.1f k  barrier %8.1f k cycles\n", w, w >> 2, w & 3, hp[w * 4] / 1e3, hp[w * 4 + 1] / 1e3, hp[w * 4 + 2] / 1e3);
#endif
    return 0;
}
