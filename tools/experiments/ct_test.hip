// dev test: conv-tile kernel vs conv-slab kernel on one 128 -> 128 convolution (bit equality), all epilogue modes.
//   hipcc --offload-arch=gfx950 -O2 -I zerovox_amd/csrc tools/micro/ct_test.hip zerovox_amd/csrc/gemm.o -o tools/micro/ct_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "zvx_kernels.h"
using namespace zvx;
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 3, M = argc > 2 ? atoi(argv[2]) : 1000, taps = argc > 3 ? atoi(argv[3]) : 7, dil = argc > 4 ? atoi(argv[4]) : 3;
    const int C = 128;
    std::vector<unsigned short> hx((size_t)B * M * C), hw((size_t)taps * C * C), hres((size_t)B * M * C), hacc((size_t)B * M * C);
    std::vector<float> hb(C);
    srand(1);
    const bool zeros = getenv("CT_ZEROS") != nullptr;
    for (auto& v : hx) v = zeros ? 0 : f2bf(frand());
    for (auto& v : hw) v = zeros ? 0 : f2bf(frand() * 0.05f);
    for (auto& v : hres) v = f2bf(frand());
    for (auto& v : hacc) v = f2bf(frand());
    for (auto& v : hb) v = frand() * 0.1f;
    std::vector<int> len(B); for (int b = 0; b < B; b++) len[b] = (b == 0 || argc > 5) ? M : 1 + rand() % M;
    unsigned short *dx, *dw, *dwp, *dres, *dacc0, *dacc, *dout; float* db; int* dlen;
    const size_t nb = (size_t)B * M * C * 2;
    hipMalloc(&dx, nb); hipMalloc(&dres, nb); hipMalloc(&dacc0, nb); hipMalloc(&dacc, nb); hipMalloc(&dout, nb);
    hipMalloc(&dw, hw.size() * 2); hipMalloc(&dwp, packed_weight_elems(taps, C, C) * 2); hipMalloc(&db, C * 4); hipMalloc(&dlen, B * 4);
    hipMemcpy(dx, hx.data(), nb, hipMemcpyHostToDevice); hipMemcpy(dres, hres.data(), nb, hipMemcpyHostToDevice);
    hipMemcpy(dacc0, hacc.data(), nb, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(dlen, len.data(), B * 4, hipMemcpyHostToDevice);
    launch_pack_weights(dw, taps, C, C, dwp, 0);
    int nbad = 0;
    for (int mode = 0; mode < 5; mode++) {
        std::vector<unsigned short> o[2], ac[2];
        int vid[2];
        for (int pass = 0; pass < 2; pass++) {
            gemm_enable_convtile(pass);
            hipMemcpy(dacc, dacc0, nb, hipMemcpyDeviceToDevice); hipMemset(dout, 0xee, nb);
            GemmArgs a; memset(&a, 0, sizeof a);
            a.X = dx; a.x_bs = (long)M * C; a.ldx = C; a.Wp = dwp; a.W = dw; a.w_ts = (long)C * C; a.ldw = C; a.dtype = DT_BF16;
            a.M = M; a.N = C; a.K = C; a.nbatch = B; a.nheads = 1; a.in_len = dlen; a.out_len = dlen; a.in_len_static = M;
            a.ntaps = taps; for (int i = 0; i < taps; i++) a.dv[i] = (i - taps / 2) * dil;
            a.stride = 1; a.alpha = 1.f; a.bias = db; a.bias_mode = 1; a.out_scale = 1.f; a.act = ACT_LRELU; a.slope = 0.1f;
            a.out = dout; a.o_bs = (long)M * C; a.ldo = C; a.out_dtype = DT_BF16;
            if (mode >= 1) { a.res = dres; a.r_bs = (long)M * C; a.ldr = C; a.res_dtype = DT_BF16; a.res_mode = 2; a.res_inv_slope = 10.f; }
            if (mode >= 2) { a.accum = dacc; a.a_bs = (long)M * C; a.lda = C; a.accum_dtype = DT_BF16; a.out_scale = 1.f / 3.f; }
            if (mode == 2) { a.accum_mode = 2; a.out = nullptr; }
            if (mode == 3) { a.accum_mode = 3; a.out = nullptr; }
            if (mode == 4) { a.accum_mode = 1; }
            vid[pass] = launch_gemm(a, 0);
            if (hipDeviceSynchronize() != hipSuccess) { printf("mode %d pass %d: device error %s\n", mode, pass, hipGetErrorString(hipGetLastError())); return 1; }
            o[pass].resize((size_t)B * M * C); ac[pass].resize((size_t)B * M * C);
            hipMemcpy(o[pass].data(), dout, nb, hipMemcpyDeviceToHost); hipMemcpy(ac[pass].data(), dacc, nb, hipMemcpyDeviceToHost);
        }
        long bad = 0, first = -1;
        for (size_t i = 0; i < o[0].size(); i++) if (o[0][i] != o[1][i] || ac[0][i] != ac[1][i]) { if (first < 0) first = (long)i; bad++; }
        printf("mode %d: variants %s / %s: %ld mismatching elements", mode, gemm_variant_name(vid[0]), gemm_variant_name(vid[1]), bad);
        if (bad) {
            nbad++;
            const long b = first / ((long)M * C), r = first / C % M, c = first % C;
            printf("  first at (b %ld, row %ld, ch %ld) slab %04x tile %04x; per-row-block map of utterance 0 (rows/32 x ch/32, count):", b, r, c, o[0][first], o[1][first]);
            for (int rb = 0; rb < (M + 31) / 32 && rb < 12; rb++) { printf("\n    rows %4d: ", rb * 32);
                for (int cb = 0; cb < 4; cb++) { int n = 0; for (int rr = rb * 32; rr < rb * 32 + 32 && rr < M; rr++) for (int cc = cb * 32; cc < cb * 32 + 32; cc++) { size_t i = ((size_t)rr) * C + cc; n += o[0][i] != o[1][i] || ac[0][i] != ac[1][i]; } printf("%5d", n); } }
        }
        printf("\n");
    }
    if (argc > 5) {                                                    // timing (+ phase cycles of CT_PROFILE builds)
        unsigned long long* dprof; const int NWG = 512; hipMalloc(&dprof, NWG * 4 * 8 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int mode = 0; mode < 5; mode++) for (int pass = 0; pass < 2; pass++) {
            gemm_enable_convtile(pass);
            GemmArgs a; memset(&a, 0, sizeof a);
            a.X = dx; a.x_bs = (long)M * C; a.ldx = C; a.Wp = dwp; a.W = dw; a.w_ts = (long)C * C; a.ldw = C; a.dtype = DT_BF16;
            a.M = M; a.N = C; a.K = C; a.nbatch = B; a.nheads = 1; a.in_len = dlen; a.out_len = dlen; a.in_len_static = M;
            a.ntaps = taps; for (int i = 0; i < taps; i++) a.dv[i] = (i - taps / 2) * dil;
            a.stride = 1; a.alpha = 1.f; a.bias = db; a.bias_mode = 1; a.out_scale = 1.f; a.act = ACT_LRELU; a.slope = 0.1f;
            a.out = dout; a.o_bs = (long)M * C; a.ldo = C; a.out_dtype = DT_BF16;
            if (mode >= 1) { a.res = dres; a.r_bs = (long)M * C; a.ldr = C; a.res_dtype = DT_BF16; a.res_mode = 2; a.res_inv_slope = 10.f; }
            if (mode >= 2) { a.accum = dacc; a.a_bs = (long)M * C; a.lda = C; a.accum_dtype = DT_BF16; a.out_scale = 1.f / 3.f; }
            if (mode == 2) { a.accum_mode = 2; a.out = nullptr; }
            if (mode == 3) { a.accum_mode = 3; a.out = nullptr; }
            if (mode == 4) { a.accum_mode = 1; }
            hipMemset(dprof, 0, NWG * 4 * 8 * 8);
            a.post_shift = (const float*)dprof;
            for (int i = 0; i < 3; i++) launch_gemm(a, 0);
            hipEventRecord(e0, 0);
            const int NIT = 10;
            for (int i = 0; i < NIT; i++) launch_gemm(a, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= NIT;
            double rows = 0; for (int b = 0; b < B; b++) rows += len[b];
            printf("time mode %d %-10s %8.3f ms  %7.1f TF/s", mode, pass ? "convtile" : "convslab", ms, 2.0 * rows * C * C * taps / ms / 1e9);
            if (pass) {
                std::vector<unsigned long long> hp(NWG * 4 * 8); hipMemcpy(hp.data(), dprof, hp.size() * 8, hipMemcpyDeviceToHost);
                double ph[6] = {0}, nt = 0, rt = 0, ct = 0; int nw = 0;
                for (int w = 0; w < NWG; w++) { const unsigned long long* o = &hp[(size_t)(w * 4 + 0) * 8]; if (!o[6]) continue; nw++; nt += o[6]; for (int k = 0; k < 6; k++) ph[k] += o[k]; rt += o[7] >> 32; ct += o[7] & 0xffffffffu; }
                if (nw) { printf("  | wave0 cycles/tile (x100MHz clock): prime+bar %.0f main %.0f bar %.0f seek+dma %.0f epi %.0f vmwait %.0f (tiles/wg %.1f) shader clock %.0f MHz", ph[0] / nt, ph[1] / nt, ph[2] / nt, ph[3] / nt, ph[4] / nt, ph[5] / nt, nt / nw, ct / rt * 100.0); }
            }
            printf("\n");
        }
    }
    printf(nbad ? "MISMATCH\n" : "ALL BIT-EQUAL\n");
    return nbad != 0;
}
