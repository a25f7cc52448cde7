// fa_bench.hip -- stand-alone timing harness for the fused attention of the FS2 decoder (attention.hip; development aid).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zerovox_amd/csrc [-DFA_EXP=<mask>] [-DFA_PROFILE] tools/micro/fa_bench.hip -o tools/micro/fa_bench
//   fa_bench [B=32] [L=896] [iters=40]
#include "../../zerovox_amd/csrc/attention.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace zvx;
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, L = argc > 2 ? atoi(argv[2]) : 896, iters = argc > 3 ? atoi(argv[3]) : 40;
    const int H = 528, nh = 2, D = 264, Lp = (L + 7) & ~7;
    std::vector<_Float16> hqk((size_t)B * L * 2 * H), hvt((size_t)B * H * Lp);
    unsigned s = 1234;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) % 2001 - 1000) * 0.001f; };
    for (auto& v : hqk) v = (_Float16)(rnd() * 0.5f);
    for (auto& v : hvt) v = (_Float16)rnd();
    void *qk, *vt, *out; int* len; long long* prof;
    hipMalloc(&qk, hqk.size() * 2); hipMalloc(&vt, hvt.size() * 2); hipMalloc(&out, (size_t)B * L * H * 2); hipMalloc(&len, B * 4); hipMalloc(&prof, 256);
    hipMemcpy(qk, hqk.data(), hqk.size() * 2, hipMemcpyHostToDevice); hipMemcpy(vt, hvt.data(), hvt.size() * 2, hipMemcpyHostToDevice); hipMemset(prof, 0, 256);
    std::vector<int> hl(B, L); hipMemcpy(len, hl.data(), B * 4, hipMemcpyHostToDevice);
    FlashArgs fa; memset(&fa, 0, sizeof fa);
    fa.qk = qk; fa.qk_bs = (long)L * 2 * H; fa.ldq = 2 * H; fa.k_off = H; fa.vt = vt; fa.vt_bs = (long)H * Lp; fa.ldv = Lp;
    fa.out = out; fa.o_bs = (long)L * H; fa.ldo = H; fa.len = len; fa.L = L; fa.D = D; fa.nheads = nh; fa.nbatch = B; fa.scale = 1.f / sqrtf((float)D); fa.f16 = 1; fa.prof = prof;
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) if (!launch_flash_attention(fa, st, false)) { printf("rejected\n"); return 1; }
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; i++) launch_flash_attention(fa, st, false);
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    const double fl = 4.0 * B * nh * (double)L * L * D;
    printf("EXP=%d B=%d L=%d: %.1f us  %.1f TF/s (%.3f of 2.5 PF)  err=%s\n", FA_EXP, B, L, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 2500, hipGetErrorString(hipGetLastError()));
#ifdef FA_PROFILE
    long long hp[8]; hipMemcpy(hp, prof, 64, hipMemcpyDeviceToHost);
    const char* nm[8] = {"k-load issue", "S chain", "k-store + v-load issue", "softmax", "PV chain", "v-store", "prologue", "barrier"};
    long long tot = 0; for (int k = 0; k < 8; k++) tot += hp[k];
    const int nt = (L + 63) / 64;
    for (int k = 0; k < 8; k++) printf("   %-24s %9.1f k cycles  (%6.0f per tile)\n", nm[k], hp[k] / 1e3, (double)hp[k] / nt);
    printf("   total %.1f k cycles per workgroup\n", tot / 1e3);
#endif
    return 0;
}
