// attention.hip -- fused scaled-dot-product attention for the FS2 / SCLN mel decoder (fs2.py:47-58, 133-164), bf16 in, f32 state.
//
//     out[q][:] = softmax_k( Q[q].K[k] / sqrt(d)  masked to k < len ) . V[k][:]
//
// One workgroup = one (utterance, head, 128-query tile); its four waves own 32 queries each.  Keys / values stream through
// LDS in 64-key tiles (double-buffered, global loads of tile t+1 in flight while tile t is computed); the [L][L] score
// matrix never exists in HBM (the unfused path moved 2 x 205 MB of f32 scores + 2 x 103 MB of bf16 probabilities per layer
// at B = 32, L = 896).  Online softmax keeps (running max, running sum) per query in registers.
//
// MFMA layout trick: scores are computed TRANSPOSED, S^T = K.Q^T (A = K rows from LDS, B = Q fragments held in registers),
// so a lane owns ONE query and 16 keys per 32-key block -- the row max / row sum are in-register reductions plus one
// exchange between lanes l and l+32 -- and P^T is already the B operand of O^T = V^T.P^T after one v_permlane32_swap per
// 8 keys (no LDS round trip for P).  V arrives transposed from its projection GEMM ([d][key], key-contiguous), so the A
// operand of the second product is a plain ds_read_b128 as well.  d = 264 is padded to 272 (17 k16 steps) for Q.K and to
// 288 (9 row tiles) for the output; one wave per SIMD with the full 512-register file (9 x 16 output accumulators, 17 Q
// fragments, 2 x 16 score accumulators).
#include "mfma_util.h"
#include "zvx_kernels.h"

#include <hip/hip_ext.h>

#include <type_traits>

namespace zvx {

static thread_local hipEvent_t g_fa_ev_start = nullptr, g_fa_ev_stop = nullptr;
void flash_profile_events(hipEvent_t start, hipEvent_t stop) { g_fa_ev_start = start; g_fa_ev_stop = stop; }

// Development switches (tools/micro/fa_bench.hip): FA_PROFILE = per-phase s_memtime totals of workgroup 0 / wave 0 -> a.prof;
// FA_EXP cuts pieces OUT (wrong results by design, only the timing means something): 1 no softmax arithmetic, 2 no staging (global
// loads / LDS stores of the next tile), 4 no second product, 8 no first product
#ifndef FA_EXP
#define FA_EXP 0
#endif
#ifdef FA_PROFILE
#define FA_STAMP(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define FA_STAMP(k) do {} while (0)
#endif
#define FA_BQ 128
#define FA_BK 64
#define FA_NKB (FA_BK / 32)      // 32-key blocks per tile
#define FA_TAU 8.0f              // log2 units the running maximum may outgrow the softmax reference before the accumulators are rescaled

// F16: Q / K / V^T / output (and the probabilities fed to the second product) are IEEE half instead of bf16 (the FS2 decoder in
// the 16-bit mode, like the StyleTTS decoder: same MFMA rate, 8x smaller rounding error)
typedef __attribute__((ext_vector_type(8))) _Float16 fa_f16x8;
template <bool F16>
__device__ __forceinline__ unsigned fa_pack2(float lo, float hi) {
    if (F16) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, (h2){(_Float16)__builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f), (_Float16)__builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f)});
    }
    return pack_bf16x2(lo, hi);
}
template <bool F16>
__device__ __forceinline__ f32x16 fa_mfma(const uint4& a_, const uint4& b_, const f32x16& c_) {
    if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(fa_f16x8, a_), __builtin_bit_cast(fa_f16x8, b_), c_, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), c_, 0, 0, 0);
}

// One chain of N = NACC x SPA matrix steps whose A fragments come from LDS: fragment I sits at `base` + (I / SPA) * ACCB + (I % SPA) * 32
// (a compile-time immediate of its ds_read_b128), feeds accumulator I / SPA together with B operand bop[I % SPA], and is requested PD
// steps ahead of its MFMA into a ring of PD + 1 register sets (counted lgkmcnt: LDS operations complete in order).  Left to hipcc the
// loop was `ds_read; s_waitcnt lgkmcnt(0); v_mfma` with ONE register set -- every matrix step paid a whole LDS round trip (round 4:
// ~130 cycles per 32-cycle MFMA).  The caller guarantees an empty LGKM queue at entry.
template <int I, int N, int PD, int SPA, int ACCB>
__device__ __forceinline__ void fa_read(uint4 (&xf)[PD + 1], unsigned base) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[I % (PD + 1)]) : "v"(base), "n"((I / SPA) * ACCB + (I % SPA) * 32));
}
template <int I, int N, int PD, int SPA, int ACCB>
__device__ __forceinline__ void fa_prefetch(uint4 (&xf)[PD + 1], unsigned base) {
    if constexpr (I < PD && I < N) { fa_read<I, N, PD, SPA, ACCB>(xf, base); fa_prefetch<I + 1, N, PD, SPA, ACCB>(xf, base); }
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
struct FaNoHook { template <int I> __device__ __forceinline__ void operator()(std::integral_constant<int, I>) const {} };
template <int I, int N, int PD, int SPA, int ACCB, bool F16, int NACC, class HOOK>
__device__ __forceinline__ void fa_steps(uint4 (&xf)[PD + 1], unsigned base, f32x16 (&acc)[NACC], const uint4 (&bop)[SPA], const HOOK& hook) {
    if constexpr (I < N) {
        if constexpr (I + PD < N) fa_read<I + PD, N, PD, SPA, ACCB>(xf, base);
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N - 1 - I >= PD ? PD : N - 1 - I) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        acc[I / SPA] = fa_mfma<F16>(xf[I % (PD + 1)], bop[I % SPA], acc[I / SPA]);
        __builtin_amdgcn_sched_barrier(0);
        hook(std::integral_constant<int, I>{});                   // (LDS-DMA requests of the next tile ride in the gaps between the matrix steps)
        fa_steps<I + 1, N, PD, SPA, ACCB, F16, NACC>(xf, base, acc, bop, hook);
    }
}
template <int PD, int SPA, int ACCB, bool F16, int NACC, class HOOK = FaNoHook>
__device__ __forceinline__ void fa_chain(unsigned base, f32x16 (&acc)[NACC], const uint4 (&bop)[SPA], const HOOK& hook = HOOK()) {
    uint4 xf[PD + 1];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    fa_prefetch<0, NACC * SPA, PD, SPA, ACCB>(xf, base);
    fa_steps<0, NACC * SPA, PD, SPA, ACCB, F16, NACC>(xf, base, acc, bop, hook);
}

template <int D, bool F16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void flash_attn_kernel(const FlashArgs a) {
    constexpr int KS = (D + 15) / 16, DP = KS * 16;             // k16 steps / padded depth of Q.K
    constexpr int NDB = (D + 31) / 32, DO = NDB * 32;           // 32-row output tiles / padded depth of the output
    constexpr int KP = DP * 2 + 16;                             // LDS pitch of a K row (bytes): odd number of 16-byte slots
    constexpr int VP = FA_BK * 2 + 16;                          // LDS pitch of a V^T row
    constexpr int KCH = DP / 8, VCH = FA_BK / 8;                // 16-byte chunks per row
    constexpr int KIT = (FA_BK * KCH + 255) / 256, VIT = (DO * VCH + 255) / 256;
    constexpr int KBYTES = FA_BK * KP, VBYTES = DO * VP;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [2][K tile | V^T tile]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l32 = lane & 31, hi = lane >> 5;   // wave: an SGPR (the DMA requests' LDS addresses are scalar arithmetic)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;   // LDS byte address of the window
    // workgroups reach the 8 XCDs round-robin in launch order: remap so that each XCD walks a CONTIGUOUS range of (utterance, head,
    // query tile) -- the <= 7 query tiles of one (utterance, head) then share that XCD's L2 for the 0.95 MB of K / V^T they all stream
    // (round 4: query tile fastest on blockIdx.x put them on 7 different XCDs: 515 MB moved per launch for 121 MB of tensors)
    const int nq = (a.L + FA_BQ - 1) / FA_BQ;
    int wgi;
    {
        const int nwg = gridDim.x, id = blockIdx.x, qn = nwg >> 3, rn = nwg & 7, xcd = id & 7;
        wgi = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (id >> 3);
    }
    const int bh = wgi / nq, b = bh / a.nheads, h = bh - b * a.nheads;
    const int len = a.len ? a.len[b] : a.L;
    const int q0 = (wgi - bh * nq) * FA_BQ;
    if (q0 >= len) return;
    const unsigned short* const Qg = (const unsigned short*)a.qk + (long)b * a.qk_bs + (long)h * D;
    const unsigned short* const Kg = Qg + a.k_off;
    const unsigned short* const Vg = (const unsigned short*)a.vt + (long)b * a.vt_bs + (long)h * D * a.ldv;
    const int q = q0 + wave * 32 + l32;

    // ---- Q fragments (B operand of S^T = K.Q^T): 8 consecutive depth elements of this lane's query per k16 step ----
    uint4 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) {
        const int d0 = 16 * s + 8 * hi;
        qf[s] = make_uint4(0, 0, 0, 0);
        if (q < len && d0 < D) qf[s] = *(const uint4*)(Qg + (long)q * a.ldq + d0);       // D % 8 == 0: a chunk is all inside or all padding
    }

    // ---- staging of one K / V^T tile by LDS-DMA (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write -- round 5: through
    //      registers the 77 KiB of a tile cost ~1000 cycles of LDS store path and ~1200 of vector-memory issue per tile, serialised with
    //      the lone wave's matrix steps).  An image is cut into 1-KiB pieces (64 lanes x 16 bytes, lane-linear in LDS); piece p, lane l is
    //      16-byte slot g = 64 p + l of the padded image: K row g / 35, chunk g % 35 (chunk 33 = the zero pad of the depth, 34 = the row
    //      pad); V^T row g / 9, chunk g % 9 (8 = the row pad).  Pad slots and rows past the tensor get an out-of-range offset (the
    //      buffer descriptor's range check returns zeros).  Wave w owns pieces w, w + 4, ...; the lane offsets are tile-invariant (the
    //      tile's first key goes into the descriptor's base).  ----
    constexpr int KSL = KP / 16, VSL = VP / 16;                 // 16-byte slots per padded row: 35, 9
    constexpr int KPC = FA_BK * KSL / 64, VPC = (D * VSL + 63) / 64;   // pieces per image: 35, 38
    static_assert(FA_BK * KSL % 64 == 0 && KBYTES % 1024 == 0 && VPC * 1024 <= VBYTES, "DMA images");
    constexpr int KPW = (KPC + 3) / 4, VPW = (VPC + 3) / 4, NDMA = KPW + VPW;   // pieces per wave: 9 + 10
    int dk_off[KPW], dv_off[VPW];
#pragma unroll
    for (int i = 0; i < KPW; i++) {
        const int pc = min(wave + 4 * i, KPC - 1), g = pc * 64 + lane, row = g / KSL, sl = g - row * KSL;    // (a surplus piece repeats the last one)
        dk_off[i] = sl < D / 8 ? (row * a.ldq + sl * 8) * 2 : -16;
    }
#pragma unroll
    for (int i = 0; i < VPW; i++) {
        const int pc = min(wave + 4 * i, VPC - 1), g = pc * 64 + lane, row = g / VSL, sl = g - row * VSL;
        dv_off[i] = (sl < FA_BK / 8 && row < D) ? (row * a.ldv + sl * 8) * 2 : -16;
    }
    // descriptors of tile k0: K rows [k0, len) of this (utterance, head) -- rows at or past len are out of range --; V^T keys from k0 on,
    // rows [0, D) through the end of the head's block (keys past len meet probabilities that are exactly zero: finite junk is harmless)
    auto k_rsrc = [&](int k0) {
        const unsigned long long pa = (unsigned long long)(Kg + (long)k0 * a.ldq);
        const long rec = ((long)(len - k0 - 1) * a.ldq + D) * 2;
        return (i32x4){__builtin_amdgcn_readfirstlane((int)(unsigned)pa), __builtin_amdgcn_readfirstlane((int)((pa >> 32) & 0xffff)),
                       __builtin_amdgcn_readfirstlane((int)(rec > 0x7fffffff ? 0x7fffffff : (rec < 0 ? 0 : rec))), 0x00020000};
    };
    auto v_rsrc = [&](int k0) {
        const unsigned long long pa = (unsigned long long)(Vg + k0);
        const long rec = ((long)D * a.ldv - k0) * 2;
        return (i32x4){__builtin_amdgcn_readfirstlane((int)(unsigned)pa), __builtin_amdgcn_readfirstlane((int)((pa >> 32) & 0xffff)),
                       __builtin_amdgcn_readfirstlane((int)(rec > 0x7fffffff ? 0x7fffffff : (rec < 0 ? 0 : rec))), 0x00020000};
    };
    auto dma = [&](const i32x4& rs, int voff, unsigned la) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(la), "v"(voff), "s"(rs) : "memory", "m0");
    };
    // request j (0 .. NDMA - 1) of this wave for the tile whose descriptors are rk / rv, into buffer `buf`: K pieces first
    auto dma_req = [&](auto jc, const i32x4& rk, const i32x4& rv, int buf) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const unsigned b0 = lds_base + (unsigned)buf * (KBYTES + VBYTES);
        if constexpr (j < KPW) dma(rk, dk_off[j], __builtin_amdgcn_readfirstlane(b0 + (unsigned)min(wave + 4 * j, KPC - 1) * 1024u));
        else dma(rv, dv_off[j - KPW], __builtin_amdgcn_readfirstlane(b0 + KBYTES + (unsigned)min(wave + 4 * (j - KPW), VPC - 1) * 1024u));
    };

    f32x16 o[NDB];
#pragma unroll
    for (int i = 0; i < NDB; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) o[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                       // running max (shared by the lane pair) and this lane's partial row sum
    const float sc = a.scale * 1.4426950408889634f;             // exp(x) = exp2(x log2 e)

    const int ntiles = (len + FA_BK - 1) / FA_BK;
#ifdef FA_PROFILE
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    {
        const i32x4 rk = k_rsrc(0), rv = v_rsrc(0);
        static_for<0, NDMA>([&](auto jc) { dma_req(jc, rk, rv, 0); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    FA_STAMP(6);
    for (int t = 0; t < ntiles; t++) {
        const int k0 = t * FA_BK;
        // the next tile's K rows and V^T rows go into the other buffer (last read before the previous tile's closing barrier); their NDMA
        // requests per wave are issued BETWEEN the matrix steps of the score product and of the first half of the second product (one per
        // ~2.7 steps: the CU's vector-memory path takes a 1-KiB request per ~16 cycles, four waves feed it; closer spacing stalls the issue)
        // (the requests are unconditional -- no branch per request: behind an utterance's last tile the K descriptor's range is empty, so
        // its pieces arrive as zeros, and the V^T pieces are in-range junk; both land in the buffer nobody reads any more)
        const i32x4 rk = k_rsrc(k0 + FA_BK), rv = v_rsrc(k0 + FA_BK);
        const int nbuf = (t + 1) & 1;
        // step g of the tile's 34 + 36 matrix steps carries request j when the ramp j = g NDMA / NSPREAD steps up there; the last NDMA-free
        // steps of the second product (and the closing barrier) cover the latency of the last requests
        constexpr int NS1 = FA_NKB * KS, NSPREAD = NS1 + NDB * 2 * FA_NKB / 2;
        auto hook1 = [&](auto ic) __attribute__((always_inline)) {
            constexpr int g = decltype(ic)::value, j = g * NDMA / NSPREAD;
            if constexpr (!(FA_EXP & 2) && g < NSPREAD && ((g + 1) * NDMA) / NSPREAD > j) dma_req(std::integral_constant<int, j>{}, rk, rv, nbuf);
        };
        auto hook2 = [&](auto ic) __attribute__((always_inline)) {
            constexpr int g = NS1 + decltype(ic)::value, j = g * NDMA / NSPREAD;
            if constexpr (!(FA_EXP & 2) && g < NSPREAD && ((g + 1) * NDMA) / NSPREAD > j) dma_req(std::integral_constant<int, j>{}, rk, rv, nbuf);
        };
        const unsigned char* const kb = lds + (t & 1) * (KBYTES + VBYTES);
        const unsigned char* const vb = kb + KBYTES;

        // ---- S^T = K.Q^T: two 32-key blocks x 32 queries ----
        f32x16 s[FA_NKB];
#pragma unroll
        for (int kbk = 0; kbk < FA_NKB; kbk++)
#pragma unroll
            for (int e = 0; e < 16; e++) s[kbk][e] = 0.f;
        FA_STAMP(0);
        if (!(FA_EXP & 8)) fa_chain<3, KS, 32 * KP, F16, FA_NKB>(lds_base + (unsigned)(kb - lds) + (unsigned)(l32 * KP + hi * 16), s, qf, hook1);
        // the other buffer's K half: last read in tile t-1's score product, behind that tile's closing barrier
        FA_STAMP(1);
        FA_STAMP(2);
        // ---- online softmax over this lane's 32 keys (+ the partner lane's 32): key of element (kbk, g, e) = 32 kbk + 8 g + 4 hi + e ----
        // the scores stay unscaled: p = exp2(s * sc - m) is one fma in front of v_exp_f32, the running max is kept in scaled units
        if (k0 + FA_BK > len) {                                     // wave-uniform: only an utterance's last tile has keys to mask
#pragma unroll
            for (int kbk = 0; kbk < FA_NKB; kbk++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int key = k0 + kbk * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
                    if (key >= len) s[kbk][e] = -INFINITY;
                }
        }
        float mt = -INFINITY;
#pragma unroll
        for (int kbk = 0; kbk < FA_NKB; kbk++)
#pragma unroll
            for (int e = 0; e < 16; e++) mt = fmaxf(mt, s[kbk][e]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_cand = fmaxf(m_run, mt * sc);                 // finite: every tile has at least one valid key (sc > 0)
        // The reference point of the exponentials only has to be NEAR the running maximum: p = exp2(s sc - m_run) stays below 2^FA_TAU as long
        // as the true maximum has not grown past m_run + FA_TAU, and softmax is a ratio -- the final division by the row sum makes any common
        // reference exact.  So the reference (and with it the 144 accumulator rescales per lane, each a trip through the accumulation
        // registers: ~1500 cycles of a ~7000-cycle tile) moves only when some query of the wave has outgrown it by more than FA_TAU; with
        // FA_TAU = 8 that is the first tile and almost never again.  f32 row sums and f16 / bf16 probabilities up to 256 are far inside range.
        const bool moved = __builtin_amdgcn_ballot_w64(m_cand > m_run + FA_TAU) != 0;      // wave-uniform (first tile: m_run = -inf)
        const float m_new = moved ? m_cand : m_run;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // first tile: exp2(-inf) = 0; not moved: exactly 1 (unused)
        m_run = m_new;
        float rs = 0.f;
        unsigned pk[FA_NKB][8];                                      // probabilities as bf16 pairs: pk[kbk][2 g + (0: e 0,1 | 1: e 2,3)]
#pragma unroll
        for (int kbk = 0; kbk < FA_NKB; kbk++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kbk][4 * g], sc, -m_new)), p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kbk][4 * g + 1], sc, -m_new));
                const float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kbk][4 * g + 2], sc, -m_new)), p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kbk][4 * g + 3], sc, -m_new));
                rs += (p0 + p1) + (p2 + p3);
                pk[kbk][2 * g] = fa_pack2<F16>(p0, p1); pk[kbk][2 * g + 1] = fa_pack2<F16>(p2, p3);
            }
        if (moved) {
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < NDB; i++)
#pragma unroll
                for (int e = 0; e < 16; e++) o[i][e] *= alpha;
        }
        l_run += rs;
        // ---- P^T as B operand of O^T += V^T.P^T: lanes l / l+32 swap the halves of every 8-key group (v_permlane32_swap) ----
        uint4 pf[2 * FA_NKB];                                            // key steps of 16: step 2 kbk + t covers keys 32 kbk + 16 t .. + 15
#pragma unroll
        for (int kbk = 0; kbk < FA_NKB; kbk++)
#pragma unroll
            for (int tt = 0; tt < 2; tt++) {
                unsigned a0 = pk[kbk][2 * (2 * tt)], a1 = pk[kbk][2 * (2 * tt) + 1];          // group g = 2 tt
                unsigned b0 = pk[kbk][2 * (2 * tt + 1)], b1 = pk[kbk][2 * (2 * tt + 1) + 1];  // group g = 2 tt + 1
                // inline asm gets none of hipcc's VALU wait states: the s_nops keep a VALU result from being consumed by the swap,
                // and the swap's result from being consumed by a VALU op, in the next two slots (stale lanes otherwise)
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
                             : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));                          // a[32:63] <-> b[0:31]
                pf[2 * kbk + tt] = make_uint4(a0, a1, b0, b1);
            }
        // ---- O^T += V^T.P^T ----
        FA_STAMP(3);
        if (!(FA_EXP & 4)) fa_chain<3, 2 * FA_NKB, 32 * VP, F16, NDB>(lds_base + (unsigned)(vb - lds) + (unsigned)(l32 * VP + hi * 16), o, pf, hook2);
        FA_STAMP(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the next tile's pieces have landed (requested a whole softmax + second product ago)
        FA_STAMP(5);
        __syncthreads();
        FA_STAMP(7);
    }
#ifdef FA_PROFILE
    if (a.prof && blockIdx.x == 0 && tid == 0) for (int k = 0; k < 8; k++) a.prof[k] = (long long)tacc[k];
#endif
    // ---- out[q][h D + d] = O[d][q] / l ----
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
    if (q < len) {
        unsigned short* const op = (unsigned short*)a.out + (long)b * a.o_bs + (long)q * a.ldo + (long)h * D;
#pragma unroll
        for (int i = 0; i < NDB; i++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int d = i * 32 + 8 * g + 4 * hi;
                if (d < D) {                                    // D % 4 == 0
                    uint2 w;
                    w.x = fa_pack2<F16>(o[i][4 * g] * inv, o[i][4 * g + 1] * inv);
                    w.y = fa_pack2<F16>(o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv);
                    *(uint2*)(op + d) = w;
                }
            }
    }
}

// returns true when the shape is covered (head depth 264, bf16, 16-byte aligned rows)
bool launch_flash_attention(const FlashArgs& a, hipStream_t stream, bool dry_run) {
    if (a.D != 264 || a.L <= 0 || a.ldq % 8 || a.ldv % 8 || a.ldo % 4 || a.k_off % 8) return false;
    if (dry_run) return true;
    constexpr int D = 264, KS = (D + 15) / 16, DP = KS * 16, DO = (D + 31) / 32 * 32;
    const size_t lds = 2 * ((size_t)FA_BK * (DP * 2 + 16) + (size_t)DO * (FA_BK * 2 + 16));
    const dim3 grid(((a.L + FA_BQ - 1) / FA_BQ) * a.nbatch * a.nheads), block(256);      // 1-D: the kernel deals (utterance, head, query tile) to the XCDs itself
    if (lds > 160 * 1024) return false;
    if (!lds_opt_in(a.f16 ? (const void*)flash_attn_kernel<264, true> : (const void*)flash_attn_kernel<264, false>)) return false;
    if (a.f16) {
        if (g_fa_ev_start) hipExtLaunchKernelGGL((flash_attn_kernel<264, true>), grid, block, lds, stream, g_fa_ev_start, g_fa_ev_stop, 0, a);
        else hipLaunchKernelGGL((flash_attn_kernel<264, true>), grid, block, lds, stream, a);
    } else {
        if (g_fa_ev_start) hipExtLaunchKernelGGL((flash_attn_kernel<264, false>), grid, block, lds, stream, g_fa_ev_start, g_fa_ev_stop, 0, a);
        else hipLaunchKernelGGL((flash_attn_kernel<264, false>), grid, block, lds, stream, a);
    }
    return true;
}


// ------------------------------------------------------------------------------------------------
// Exact-f32 attention for the phoneme encoder (fs2.py:47-58, 133-164).  The encoder feeds discrete decisions (bucket ids,
// durations), so its attention stays in f32 end to end: v_mfma_f32_32x32x2f32 for both products, expf for the softmax.
// One workgroup = (utterance, head, 32 queries); keys / values stream through LDS in tiles of 128 keys, each in two depth
// chunks (128 + 136 of d = 264), online softmax state (running max / sum) per query row in registers of the 8 threads that
// own the row.  Replaces V^T GEMM + score GEMM + softmax + P.V GEMM (+ a memset) of the unfused path: [L][L] scores and
// probabilities never leave the chip, and Q | K | V come from ONE projection GEMM.
//   * MFMA operands: lane (i = l % 32, kh = l / 32) supplies element [i][k] of A resp. [k][i] of B for k = 2 step + kh.  The
//     contraction order is free as long as A and B agree, so one ds_read_b128 at depth 8 j + 4 kh feeds FOUR steps (element s of
//     both halves); row pitches are 4 x odd dwords, which makes those b128 reads conflict-free.
//   * key tiles are fixed ([0,128), [128,256) ...) and masked by the utterance's own length, so an utterance's result does
//     not depend on the batch it travels in.
// ------------------------------------------------------------------------------------------------
template <int IT> struct AfTile { float4 v[IT]; };      // a tile's global loads in flight (registers) between fetch and LDS commit
#define AF_BQ 32
#define AF_BK 128
template <int D>
__global__ __launch_bounds__(256) void attn_f32_kernel(const AttnF32Args a) {
    constexpr int C0 = 128, C1 = D - C0;                        // depth chunks, multiples of 8
    constexpr int QP = D + 4, KP = (C1 > C0 ? C1 : C0) + 4, PP = AF_BK + 4;   // pitches in floats: 268, 140, 132 (all 4 x odd)
    static_assert(C1 % 8 == 0 && C0 % 8 == 0 && (QP / 4) % 2 == 1 && (KP / 4) % 2 == 1 && (PP / 4) % 2 == 1, "chunking / pitches");
    constexpr int NT1 = (C1 + 31) / 32;                         // column tiles of the second chunk (5: the last one 8 wide)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* const Qs = (float*)lds_raw;                          // [32][QP]
    float* const Ks = Qs + AF_BQ * QP;                          // [128][KP]   K chunk, then V chunk
    float* const Ps = Ks + AF_BK * KP;                          // [32][PP]    scores, then probabilities
    float* const As = Ps + AF_BQ * PP;                          // [32] rescale factor of the running output, then 1 / sum

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l32 = lane & 31, kh = lane >> 5;
    const int b = blockIdx.y / a.nheads, h = blockIdx.y - b * a.nheads;
    const int len = a.len ? a.len[b] : a.L;
    const int q0 = blockIdx.x * AF_BQ;
    if (q0 >= len) return;
    const float* const base = a.qkv + (long)b * a.bs + (long)h * D;

    // rows [row0, row0 + NROWS) x columns [0, NCOLS) of a row-major f32 matrix -> LDS tile (rows at or beyond the utterance's length: zeros).  Every
    // load of the tile is in flight before the first LDS store (a load -> store loop pays one L2 round trip per iteration).
    auto tile_fetch = [&](const float* src, int row0, auto nrows_c, auto ncols_c) {
        constexpr int NROWS = decltype(nrows_c)::value, NCOLS = decltype(ncols_c)::value;
        constexpr int C4N = NCOLS / 4, N = NROWS * C4N, IT = (N + 255) / 256;
        AfTile<IT> t;
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const int i = tid + it * 256, r = i / C4N, c4 = i - r * C4N, row = row0 + r;
            t.v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < N && row < len) t.v[it] = *(const float4*)(src + (long)row * a.ld + c4 * 4);     // rows past the utterance's OWN length are zeros: a V row there meets a probability of exactly 0, and 0 x (stale NaN) is not 0 (it read `row < a.L` until round 5)
        }
        return t;
    };
    auto tile_commit = [&](const auto& t, auto nrows_c, auto ncols_c, float* dst, int pitch) {
        constexpr int NROWS = decltype(nrows_c)::value, NCOLS = decltype(ncols_c)::value;
        constexpr int C4N = NCOLS / 4, N = NROWS * C4N, IT = (N + 255) / 256;
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const int i = tid + it * 256, r = i / C4N, c4 = i - r * C4N;
            if (i < N) *(float4*)(dst + r * pitch + c4 * 4) = t.v[it];
        }
    };
    auto load_tile = [&](const float* src, int row0, auto nrows_c, auto ncols_c, float* dst, int pitch) {
        tile_commit(tile_fetch(src, row0, nrows_c, ncols_c), nrows_c, ncols_c, dst, pitch);
    };
    constexpr std::integral_constant<int, AF_BQ> c_bq{};
    constexpr std::integral_constant<int, AF_BK> c_bk{};
    constexpr std::integral_constant<int, D> c_d{};
    constexpr std::integral_constant<int, C0> c_c0{};
    constexpr std::integral_constant<int, C1> c_c1{};
    load_tile(base + a.q_off, q0, c_bq, c_d, Qs, QP);

    f32x16 o0, o1, o2;
#pragma unroll
    for (int e = 0; e < 16; e++) { o0[e] = 0.f; o1[e] = 0.f; o2[e] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;                       // softmax state of row tid / 8 (all 8 threads of a row hold it)
    const int srow = tid >> 3, ssub = tid & 7;
    const float* const qrow = Qs + l32 * QP + 4 * kh;
    const float* const krow = Ks + (wave * 32 + l32) * KP + 4 * kh;
    const float* const prow = Ps + l32 * PP + 4 * kh;

    const int ntiles = (len + AF_BK - 1) / AF_BK;
    for (int kt = 0; kt < ntiles; kt++) {
        const int k0 = kt * AF_BK;
        const bool my_keys = k0 + wave * 32 < len;                                  // wave-uniform: this wave's 32 keys hold a valid one
        // ---- S = Q K^T over the two depth chunks ----
        f32x16 s, s2;                                                               // two independent MFMA chains (a lone wave per SIMD cannot hide the dependent-issue latency)
#pragma unroll
        for (int e = 0; e < 16; e++) { s[e] = 0.f; s2[e] = 0.f; }
        __syncthreads();                                                            // Ks (V of the previous tile) and Ps are free
        load_tile(base + a.k_off, k0, c_bk, c_c0, Ks, KP);
        __syncthreads();
        const auto k1 = tile_fetch(base + a.k_off + C0, k0, c_bk, c_c1);            // in flight under the chunk-0 products
        if (my_keys) {
#pragma unroll
            for (int j = 0; j < C0 / 8; j++) {
                const float4 qa = *(const float4*)(qrow + 8 * j), kb = *(const float4*)(krow + 8 * j);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, kb.x, s, 0, 0, 0);
                s2 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, kb.y, s2, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, kb.z, s, 0, 0, 0);
                s2 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, kb.w, s2, 0, 0, 0);
            }
        }
        __syncthreads();
        tile_commit(k1, c_bk, c_c1, Ks, KP);
        const auto v0 = tile_fetch(base + a.v_off, k0, c_bk, c_c0);                 // V chunk 0: in flight under the chunk-1 products and the softmax
        __syncthreads();
        if (my_keys) {
#pragma unroll
            for (int j = 0; j < C1 / 8; j++) {
                const float4 qa = *(const float4*)(qrow + C0 + 8 * j), kb = *(const float4*)(krow + 8 * j);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, kb.x, s, 0, 0, 0);
                s2 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, kb.y, s2, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, kb.z, s, 0, 0, 0);
                s2 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, kb.w, s2, 0, 0, 0);
            }
            s += s2;
        }
        // scores -> LDS: accumulator e of lane (key l32, kh) is query row 8 (e / 4) + 4 kh + e % 4
#pragma unroll
        for (int e = 0; e < 16; e++) Ps[(8 * (e >> 2) + 4 * kh + (e & 3)) * PP + wave * 32 + l32] = s[e] * a.scale;
        __syncthreads();                                                            // K chunk consumed by every wave, scores complete
        tile_commit(v0, c_bk, c_c0, Ks, KP);
        const auto v1 = tile_fetch(base + a.v_off + C0, k0, c_bk, c_c1);            // V chunk 1: under the softmax and the chunk-0 P.V
        {   // ---- online softmax of row srow over this tile's keys (16 per thread) ----
            float* const pr = Ps + srow * PP + ssub * 16;
            float v[16];
#pragma unroll
            for (int i = 0; i < 4; i++) { const float4 t = *(const float4*)(pr + 4 * i); v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w; }
            const int kbase = k0 + ssub * 16;
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; i++) if (kbase + i < len) m = fmaxf(m, v[i]);
            m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 4, 64));
            const float m_new = fmaxf(m_run, m);                                    // finite: key k0 < len is valid for every row
            const float alpha = m_run == -INFINITY ? 0.f : expf(m_run - m_new);
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++) { v[i] = kbase + i < len ? expf(v[i] - m_new) : 0.f; sum += v[i]; }
            sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 4, 64);
            l_run = alpha * l_run + sum; m_run = m_new;
#pragma unroll
            for (int i = 0; i < 4; i++) *(float4*)(pr + 4 * i) = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
            if (ssub == 0) As[srow] = alpha;
        }
        __syncthreads();
        // ---- O = alpha O + P V ----
        {
            float al[16];
#pragma unroll
            for (int g = 0; g < 4; g++) { const float4 t = *(const float4*)(As + 8 * g + 4 * kh); al[4 * g] = t.x; al[4 * g + 1] = t.y; al[4 * g + 2] = t.z; al[4 * g + 3] = t.w; }
#pragma unroll
            for (int e = 0; e < 16; e++) { o0[e] *= al[e]; o1[e] *= al[e]; o2[e] *= al[e]; }
        }
        // every step of the tile is issued: probabilities beyond the last valid key are exactly zero and V rows there are
        // finite (zeros beyond L), so they add nothing -- and the loop has no branches for the scheduler to stop at
        auto pv_tile = [&](f32x16 acc, int col) {                                   // col: column of the chunk this lane supplies
            const float* const vc = Ks + col + 4 * kh * KP;
            f32x16 acc2;
#pragma unroll
            for (int e = 0; e < 16; e++) acc2[e] = 0.f;
#pragma unroll
            for (int j = 0; j < AF_BK / 8; j++) {
                const float4 pa = *(const float4*)(prow + 8 * j);
                const float* const vj = vc + 8 * j * KP;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.x, vj[0], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.y, vj[KP], acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.z, vj[2 * KP], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.w, vj[3 * KP], acc2, 0, 0, 0);
            }
            return acc + acc2;
        };
        o0 = pv_tile(o0, wave * 32 + l32);                                          // columns [32 w, 32 w + 32) of chunk 0
        __syncthreads();
        tile_commit(v1, c_bk, c_c1, Ks, KP);
        __syncthreads();
        o1 = pv_tile(o1, min(wave * 32 + l32, C1 - 1));                             // columns C0 + [32 w, 32 w + 32)
        if (NT1 > 4 && wave == 3) o2 = pv_tile(o2, min(128 + l32, C1 - 1));         // the 8-column remainder (C0 + 128 ...)
    }
    // ---- out[q][h D + col] = O / sum ----
    __syncthreads();
    if (ssub == 0) As[srow] = 1.0f / l_run;
    __syncthreads();
    float* const op = a.out + (long)b * a.o_bs + (long)h * D;
    auto store_tile = [&](const f32x16& acc, int col) {
        if (col >= D) return;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int r = 8 * (e >> 2) + 4 * kh + (e & 3), q = q0 + r;
            if (q < a.L) {
                const float val = q < len ? acc[e] * As[r] : 0.f;
                op[(long)q * a.ldo + col] = val;
                if (a.planes) {                                                     // [hi | hi | lo] 16-bit planes for the output projection's split-product GEMM
                    unsigned short hi, lo;
                    if (a.planes_f16) {                                             // IEEE-half planes: lo scaled by 2^11 (ops.hip: split2h)
                        const _Float16 hh = (_Float16)__builtin_amdgcn_fmed3f(val, -65504.f, 65504.f);
                        hi = __builtin_bit_cast(unsigned short, hh);
                        lo = __builtin_bit_cast(unsigned short, (_Float16)__builtin_amdgcn_fmed3f((val - (float)hh) * 2048.f, -65504.f, 65504.f));
                    } else {
                        hi = __builtin_bit_cast(unsigned short, (__bf16)val);
                        lo = __builtin_bit_cast(unsigned short, (__bf16)(val - __uint_as_float((unsigned)hi << 16)));
                    }
                    unsigned short* const pp = a.planes + ((long)b * a.L + q) * 3 * a.planes_C + h * D + col;
                    pp[0] = hi; pp[a.planes_C] = hi; pp[2 * a.planes_C] = lo;
                }
            }
        }
    };
    store_tile(o0, wave * 32 + l32);
    store_tile(o1, C0 + wave * 32 + l32);
    if (NT1 > 4 && wave == 3) store_tile(o2, C0 + 128 + l32);
}

bool launch_attention_f32(const AttnF32Args& a, hipStream_t stream, bool dry_run) {
    if (a.D != 264 || a.L <= 0 || a.ld % 4 || a.q_off % 4 || a.k_off % 4 || a.v_off % 4) return false;
    if (dry_run) return true;
    if ((size_t)a.qkv & 15) return false;
    constexpr int D = 264, KP = (D - 128) + 4;
    const size_t lds = sizeof(float) * ((size_t)AF_BQ * (D + 4) + (size_t)AF_BK * KP + (size_t)AF_BQ * (AF_BK + 4) + 32);
    auto kfn = attn_f32_kernel<264>;
    if (!lds_opt_in((const void*)kfn)) return false;
    const dim3 grid((a.L + AF_BQ - 1) / AF_BQ, a.nbatch * a.nheads), block(256);
    if (g_fa_ev_start) hipExtLaunchKernelGGL(kfn, grid, block, lds, stream, g_fa_ev_start, g_fa_ev_stop, 0, a);
    else hipLaunchKernelGGL(kfn, grid, block, lds, stream, a);
    return true;
}

}  // namespace zvx
