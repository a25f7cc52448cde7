// resstream.hip -- a whole HiFi-GAN ResBlock1 (hifigan.py:49-56: three times  x += conv_1(lrelu(conv_d(lrelu(x)))) ) as ONE
// streaming kernel for the narrow stages (C = 32 / 64), where the unfused convolutions are HBM-bound.
//
// Idea (the "LDS-ring-buffered 1-D convolution"): the 2*NPAIR convolutions of the chain are a software pipeline along
// TIME.  A persistent workgroup walks a segment of one utterance in steps of R rows; every convolution ("role") is owned
// by a fixed group of waves that keep ITS weight fragments in registers for the whole launch, reads its input rows from
// an LDS ring written by the previous role and writes its output rows to the next ring:
//
//     HBM --DMA--> [X0 ring] -conv1_0-> [T0 ring] -conv2_0 (+x from X0)-> [X1 ring] -conv1_1-> [T1] -conv2_1 (+X1)-> [X2] ...
//                                                                                      ... -conv2_last (+ xs) --> HBM
//
// Role r works one block behind role r-1 and its block boundaries are shifted left by the cumulated halo H_r, so the rows
// it needs (its block plus h_r rows either side) are exactly the rows role r-1 finished one step earlier: one barrier per
// step, ring r only holds 2R + 2h rows (3R + h1 + h2 for the X rings, which the second-next role re-reads as residual).
// The stage tensor therefore crosses HBM ONCE per ResBlock instead of once per convolution pair, halo rows are computed
// once per SEGMENT (not per tile), and no weight is ever re-fetched.
//
// Numerics are identical to the pair kernels of gemm.hip (same bf16 roundings of the intermediates, same accumulation
// order tap-major / k16-minor), so results are bit-equal to the unfused path.
#include "mfma_util.h"
#include "zvx_kernels.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <type_traits>

namespace zvx {

static thread_local hipEvent_t g_rs_ev_start = nullptr, g_rs_ev_stop = nullptr;
void resstream_profile_events(hipEvent_t start, hipEvent_t stop) { g_rs_ev_start = start; g_rs_ev_stop = stop; }

#define RS_RD 64          // rows per DMA block
#define RS_PF 2           // DMA blocks are requested this many steps before role 0 needs them

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// The MFMA chain of one block with compile-time operand addressing: fragment i = (tap i / KS, k16 slot i % KS) is read from
// `base` + an immediate, PD reads ahead of its MFMA (counted lgkmcnt waits; LDS operations complete in order).
template <int I, int PD, int KS, int DIL, int P>
__device__ __forceinline__ void rs_read(uint4 (&xf)[PD + 1], unsigned base) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[I % (PD + 1)]) : "v"(base), "n"((I / KS) * DIL * P + (I % KS) * 32));
}
template <int I, int PD, int KS, int DIL, int P>
__device__ __forceinline__ void rs_prefetch(uint4 (&xf)[PD + 1], unsigned base) {
    if constexpr (I < PD) { rs_read<I, PD, KS, DIL, P>(xf, base); rs_prefetch<I + 1, PD, KS, DIL, P>(xf, base); }
}
template <int I, int NW, int PD, int KS, int DIL, int P, bool H16>
__device__ __forceinline__ void rs_mma_steps(uint4 (&xf)[PD + 1], const uint4 (&w)[NW], unsigned base, f32x16& acc) {
    if constexpr (I < NW) {
        if constexpr (I + PD < NW) rs_read<I + PD, PD, KS, DIL, P>(xf, base);
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(NW - 1 - I >= PD ? PD : NW - 1 - I) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        acc = mfma16<H16>(w[I], xf[I % (PD + 1)], acc);
        __builtin_amdgcn_sched_barrier(0);
        rs_mma_steps<I + 1, NW, PD, KS, DIL, P, H16>(xf, w, base, acc);
    }
}

// wave -> (role, sub): waves w, w+4, w+8 share a SIMD; the tables give every SIMD a mix of conv1- and conv2-kind roles.
// 12 waves: the conv2 roles carry the long epilogues (residual + two activations) and the last one also the HBM store phase,
// so SIMDs 0 / 2 get {conv2_0, conv2_1, conv1_0 (+ DMA issue)} and SIMDs 1 / 3 get {conv2_2 (+ stores), conv1_1, conv1_2}.
template <int NR, int WPR>
__device__ __forceinline__ void role_of_wave(int w, int& role, int& sub, bool balanced) {
    if (NR == 6 && WPR == 2 && balanced) { role = (int)((0x404023235151ull >> (4 * w)) & 15); sub = (w >> 1) & 1; }
    else if (NR == 6 && WPR == 2) { role = (int)((0x452301453210ull >> (4 * w)) & 15); sub = w >= 6; }
    else if (NR == 4 && WPR == 2) { role = (int)((0x23013210u >> (4 * w)) & 15); sub = w >= 4; }
    else if (NR == 2 && WPR == 4) { role = (int)((0x01011010u >> (4 * w)) & 15); sub = w >> 1; }
    else { role = w % NR; sub = w / NR; }
}

// Issue priority of a wave (s_setprio).  After the step barrier every wave of a SIMD starts its MFMA chain at once; with equal
// priorities the chains interleave, all finish together and the epilogues (VALU) then run with the matrix pipe idle.  With
// distinct priorities the chains run one after the other and the epilogue of an earlier wave overlaps the MFMAs of the next;
// the waves with the long epilogue (conv2: residual + two activations) go first.
template <int NR, int WPR>
__device__ __forceinline__ int prio_of_wave(int w, bool balanced) {
    if (NR == 6 && WPR == 2) return balanced ? 3 - (w >> 2) : (int)((0x122223133131ull >> (4 * w)) & 15);
    return w < 4 ? 2 : 1;
}

// DP: the dilations of the conv1 roles, one hex digit per pair (0x531 = 1, 3, 5): with the dilation a compile-time constant the
// row offset of every tap is an immediate of its ds_read_b128 and the MFMA loop carries no address arithmetic at all.
// ring geometry of a chain: compile-time (kernel) and launch-time (LDS size, shape checks) from the same formulas
template <int C, int NT, int NPAIR, int RSPLIT, int DP>
struct RsGeom {
    static constexpr int R = 32 * RSPLIT, H2 = (NT - 1) / 2;
    static constexpr int dil(int p) { return (DP >> (4 * p)) & 15; }
    static constexpr int h0 = dil(0) * H2;
    static constexpr int dT = 2 * R + 2 * H2;                                                   // T rings: two blocks + conv2's halo
    static constexpr int dX0 = (((RS_PF + 2) * R + RS_RD + h0 + H2 + 2 * h0) + RS_RD - 1) / RS_RD * RS_RD;   // covers the DMA lead, whole 64-row blocks
    static constexpr int dX(int p) {                                                            // X rings of the later pairs (also re-read as residual)
        return p == 0 ? 0 : (2 * R + 2 * dil(p) * H2 > 3 * R + dil(p) * H2 + H2 ? 2 * R + 2 * dil(p) * H2 : 3 * R + dil(p) * H2 + H2);
    }
};

// One role of the chain.  Everything that depends on the role -- which convolution, its dilation, its rings and their sizes, whether it
// issues the DMA or owns the store phase -- is a compile-time constant here: the step loop of a generic body spent as many scalar
// instructions on that bookkeeping (66 per step and wave) as vector instructions on the epilogue (76), and a SIMD issues about one
// instruction per 4 cycles whatever its kind (SQ_ACTIVE_INST_ANY ~ 90 % of the kernel's cycles on the narrow stages).
template <int C, int NT, int NPAIR, int RSPLIT, int AM, bool HAS_OUT, int DP, bool H16, int ROLE>
__device__ __forceinline__ void rs_role(const StreamArgs& a, unsigned char* const lds, const int lane, const int sub, const int wave) {
    using G = RsGeom<C, NT, NPAIR, RSPLIT, DP>;
    constexpr int NR = 2 * NPAIR, NTL = C / 32, WPR = NTL * RSPLIT;
    constexpr int R = 32 * RSPLIT, KS = C / 16, P = 2 * C + 16, CPP = C / 8 + 1, H2 = (NT - 1) / 2, NW = NT * KS;
    constexpr int PPB = CPP;                        // 1-KiB DMA pieces per 64-row block (64 rows x CPP 16-byte slots / 64 lanes)
    constexpr int PPW = (PPB + WPR - 1) / WPR;      // pieces per issuing wave and block (surplus ones repeat the last piece)
    constexpr int NKC = (C + 63) / 64;
    constexpr int role = ROLE, pair = ROLE >> 1, kind = ROLE & 1;
    const int ct = sub % NTL, rs = sub / NTL;
    const int l32 = lane & 31, koff = (lane >> 5) * 16, h4 = 4 * (lane >> 5);
    constexpr bool is_final = ROLE == NR - 1;

    // ---- chain geometry (wave-uniform) ----
    int my_h = 0, my_H = 0, Hsum = 0;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int h = (r & 1) ? H2 : G::dil(r >> 1) * H2;
        Hsum += h;
        if (r <= role) my_H += h;
        if (r == role) my_h = h;
    }
    const int h0 = G::dil(0) * H2;
    const int my_dil = kind ? 1 : G::dil(pair);
    const int NB0 = (Hsum + R - 1) / R;
    // LDS map: X0 | T0 | X1 | T1 | X2 | T2 | stage (final role) | bias table
    int offX = 0, offT = 0, dXp = G::dX0, offXn = 0, dXn = 0;
    {
        int cur = G::dX0 * P;
#pragma unroll
        for (int p = 0; p < NPAIR; p++) {
            if (p == pair) offT = cur;
            cur += G::dT * P;
            if (p + 1 < NPAIR) {
                if (p + 1 == pair) { offX = cur; dXp = G::dX(p + 1); }
                if (p == pair) { offXn = cur; dXn = G::dX(p + 1); }
                cur += G::dX(p + 1) * P;
            }
        }
        offXn = __builtin_amdgcn_readfirstlane(offXn); offX = __builtin_amdgcn_readfirstlane(offX); offT = __builtin_amdgcn_readfirstlane(offT);
        dXp = __builtin_amdgcn_readfirstlane(dXp); dXn = __builtin_amdgcn_readfirstlane(dXn);
    }
    int ring_end = G::dX0 * P + NPAIR * G::dT * P;
#pragma unroll
    for (int p = 1; p < NPAIR; p++) ring_end += G::dX(p) * P;
    unsigned char* const stw = lds + ring_end + sub * (32 * 80);                  // final role, per wave: 32 rows x 32 ch bf16, pitch 80
    // rings of this role
    const int in_off = kind ? offT : offX, Din = kind ? G::dT : dXp;               // operand source
    const int out_off = kind ? offXn : offT, Dout = kind ? dXn : G::dT;            // destination ring (unused by the final role)
    const int res_off = offX, Dres = dXp;                                         // conv2: residual source = the pair's input stream

    // ---- this wave's weights: conv1 <- W1[pair], conv2 <- W2[pair]; packed stream [nt32][chunk][tap][4 k16 slots], 1 KiB fragments ----
    const void* wbase = nullptr;
    const float* bsrc = nullptr;
#pragma unroll
    for (int p = 0; p < NPAIR; p++)
        if (p == pair) { wbase = kind ? a.W2[p] : a.W1[p]; bsrc = kind ? a.b2[p] : a.b1[p]; }
    const uint4* const Wq = (const uint4*)wbase + ((long)ct * NKC * NT * 4) * 64 + lane;
    uint4 w[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) { const int t = i / KS, kk = i % KS; w[i] = Wq[(((kk >> 2) * NT + t) * 4 + (kk & 3)) * 64]; }
    // bias table [NR][C] in LDS (after the final role's stages): 16 values per lane, re-read per block instead of held in registers
    float* const bias_l = (float*)(lds + ring_end + WPR * (32 * 80));
    if (rs == 0 && lane < 32) bias_l[role * C + ct * 32 + lane] = bsrc[ct * 32 + lane];
    __syncthreads();
    // settle the loads here: a compiler-placed wait inside the step loop would also wait for the (hidden) slab DMAs
#pragma unroll
    for (int i = 0; i < NW; i++) asm volatile("" :: "v"(w[i].x));

    // ---- DMA lane offsets (role 0): piece j of a 64-row block = 16-byte slots [64 j, 64 j + 64) of its padded LDS image ----
    const unsigned lds_addr0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int nslots = G::dX0 / RS_RD;                                              // DMA blocks the X0 ring holds
    int dma_off[PPW];                                                              // byte offset of this lane's 16 bytes within a DMA block, per piece; pad slot: out of range -> zeros
#pragma unroll
    for (int n = 0; n < PPW; n++) {
        const int j = sub + WPR * n < PPB ? sub + WPR * n : PPB - 1;
        const int slot = j * 64 + lane, row = slot / CPP, qs = slot % CPP;
        dma_off[n] = qs == CPP - 1 ? -16 : (row * a.ldx + (qs << 3)) * 2;
    }
    const unsigned lane_in = lds_addr0 + in_off + koff + l32 * P;                  // this lane's operand row 0 of the input ring

    const float slope1 = a.slope1, rinv = a.res_inv_slope, oscale = a.out_scale, oslope = a.slope;

#ifdef RS_PROFILE
    unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#define RS_STAMP(k) do { if (a.prof) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[k] += t_ - tlast; tlast = t_; } } while (0)
#else
#define RS_STAMP(k) do {} while (0)
#endif
    const int nsegs = a.nseg * a.nbatch;
    for (int seg = blockIdx.x; seg < nsegs; seg += gridDim.x) {
        const int b = seg / a.nseg, sj = seg - b * a.nseg;
        int len = a.len ? a.len[b] : a.M;                                          // uniform address: scalar load
        len = __builtin_amdgcn_readfirstlane(len);
        const int seg0 = sj * a.S;
        if (seg0 >= len) continue;
        const int seg_end = min(seg0 + a.S, len);
        const int b_last = (seg_end - seg0 + Hsum + R - 1) / R - 1;               // last block of the final role (others run the same count)
        const int nact = b_last + NB0 + 1;                                         // blocks per role
        const int nsteps = nact + NR - 1;                                          // role r works on block i in step i + r
        const int o0 = seg0 - NB0 * R - 2 * h0;                                    // first row of DMA block 0
        const unsigned short* const Xb = (const unsigned short*)a.X + (long)b * a.x_bs;

        // ring positions at this role's first block (independent of seg0: every segment starts from the same image)
        int rd_pos = role == 0 ? rs * 32 : (Din - 2 * my_h + rs * 32) % Din;
        int wr_pos = rs * 32;
        int rs_pos = pair == 0 ? (h0 - H2 + rs * 32) : ((Dres - (G::dil(pair) * H2 + H2) % Dres + rs * 32) % Dres);
        int g_out0 = seg0 - NB0 * R - my_H + rs * 32;                              // global row of this wave's first output row
        f32x16 acc;

        // ---- DMA issue (role 0 waves) ----
        int issued = 0, dslot = 0;
        auto dma_block = [&]() {
            const int g0 = o0 + issued * RS_RD;
            const unsigned long long pa = (unsigned long long)(Xb + (long)g0 * a.ldx);
            int nrec = (len - g0) * a.ldx * 2; if (nrec < 0) nrec = 0;
            const i32x4 rsrc = {__builtin_amdgcn_readfirstlane((int)(unsigned)pa), __builtin_amdgcn_readfirstlane((int)((pa >> 32) & 0xffff)),
                                __builtin_amdgcn_readfirstlane(nrec), 0x00020000};
            const int thr = -g0 * a.ldx * 2;                                       // offsets below this belong to rows before the utterance
            const unsigned la0 = lds_addr0 + offX + dslot * (RS_RD * P);
#pragma unroll
            for (int n = 0; n < PPW; n++) {
                const int j = sub + WPR * n < PPB ? sub + WPR * n : PPB - 1;
                // interior blocks (g0 >= 0) use the per-lane offsets as precomputed; blocks that start before the utterance
                // push the rows before it out of range (zeros)
                const int voff = g0 >= 0 ? dma_off[n] : (dma_off[n] < thr ? -16 : dma_off[n]);
                const unsigned la = __builtin_amdgcn_readfirstlane(la0 + j * 1024);
                asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(la), "v"(voff), "s"(rsrc) : "memory", "m0");
            }
            issued++; dslot++; if (dslot == nslots) dslot = 0;
        };
        auto need = [&](int s) { return (2 * h0 + (s + 1) * R - 1) / RS_RD; };      // last DMA block role 0 reads in step s
        auto wait_landed = [&](int nblocks) {                                      // all but the newest `issued - nblocks` blocks have landed
            const int out = issued - nblocks;
            if (out <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (out == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW) : "memory");
            else if (out == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PPW) : "memory");
            else if (out == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * PPW) : "memory");
        };
        if (role == 0) {
            const int target = need(RS_PF - 1) + 1;
            while (issued < target) dma_block();
            wait_landed(need(0) + 1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

        // final role: the store phase of a block (stage -> xs / output in HBM) runs at the START of the next step, beside the
        // other roles' MFMAs; its xs rows are requested a step early
        bool pend = false; int pend_g0 = 0;
        uint4 xs[2];
        // xs / output rows of this segment through raw buffers whose range IS the segment [seg0, seg_end): rows outside it (the
        // pipeline's lead-in, the tail of the last block) fall out of range -- loads return 0, stores are dropped -- so the store
        // phase carries no row masks and no 64-bit address arithmetic
        const int nvalid = seg_end - seg0;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((unsigned short*)a.accum + (long)b * a.a_bs + (long)seg0 * a.lda), 0, a.accum ? nvalid * a.lda * 2 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)((unsigned short*)a.out + (long)b * a.o_bs + (long)seg0 * a.ldo), 0, a.out ? nvalid * a.ldo * 2 : 0, 0x00020000);
        const int lane_col = (ct * 32 + (lane & 3) * 8) * 2, lane_row = (lane >> 2) - seg0;
        auto row_off = [&](int g0, int h, int ld) { return __mul24(g0 + h * 16 + lane_row, ld * 2) + lane_col; };
        auto store_phase = [&]() {
            uint4 o[2];
#pragma unroll
            for (int h = 0; h < 2; h++) o[h] = *(const uint4*)(stw + (h * 16 + (lane >> 2)) * 80 + (lane & 3) * 16);
#ifdef RS_PROFILE
            asm volatile("" :: "v"(o[0].x), "v"(o[1].x));
            RS_STAMP(7);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            RS_STAMP(8);
#endif
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (AM & 1 || HAS_OUT) {
                    f32x2 t[4] = {unpack16<H16>(o[h].x), unpack16<H16>(o[h].y), unpack16<H16>(o[h].z), unpack16<H16>(o[h].w)};
                    if (AM & 1) {
                        t[0] += unpack16<H16>(xs[h].x); t[1] += unpack16<H16>(xs[h].y);
                        t[2] += unpack16<H16>(xs[h].z); t[3] += unpack16<H16>(xs[h].w);
                        if (AM & 2) o[h] = make_uint4(pack16<H16>(t[0].x, t[0].y), pack16<H16>(t[1].x, t[1].y), pack16<H16>(t[2].x, t[2].y), pack16<H16>(t[3].x, t[3].y));
                    }
                    if (AM & 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[h]), rsA, row_off(pend_g0, h, a.lda), 0, 0);
                    if (HAS_OUT) {
                        if (AM) {
#pragma unroll
                            for (int e = 0; e < 4; e++) t[e] = lrelu2(t[e] * oscale, oslope);
                            o[h] = make_uint4(pack16<H16>(t[0].x, t[0].y), pack16<H16>(t[1].x, t[1].y), pack16<H16>(t[2].x, t[2].y), pack16<H16>(t[3].x, t[3].y));
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[h]), rsO, row_off(pend_g0, h, a.ldo), 0, 0);
                    }
                } else if (AM & 2) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[h]), rsA, row_off(pend_g0, h, a.lda), 0, 0);   // AM == 2: the staged bf16 rows as they are
                }
            }
        };

        // One block of one role.  KIND 0: conv1 (-> T ring), 1: conv2 feeding the next pair (-> X ring), 2: the chain's last conv2.
        // Every LDS read is inline asm with counted lgkmcnt waits (LDS operations complete in order): the B fragments are
        // requested PD ahead of the MFMA that consumes them.
        auto mma_slow = [&]() {
            constexpr int PD = NW <= 6 ? NW : (NW >= 40 ? 4 : 6);
#pragma unroll
            for (int e = 0; e < 16; e++) acc[e] = 0.f;
            const unsigned p0 = rd_pos + l32;
            const unsigned in_addr = lds_addr0 + in_off + koff;
            uint4 xf[PD + 1];
            unsigned ta = 0;                                                       // LDS byte address of this lane's row for the tap being requested
            auto request = [&](int i) {                                            // i is a literal at every call site (fully unrolled)
                if (i % KS == 0) {
                    unsigned pt = p0 + (i / KS) * my_dil;
                    pt = min(pt, pt - (unsigned)Din);                              // pt >= Din -> pt - Din (unsigned wrap trick)
                    ta = in_addr + pt * P;
                }
                switch (i % KS) {                                                  // the offset must be an immediate
                    case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(xf[i % (PD + 1)]) : "v"(ta)); break;
                    case 1: asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(xf[i % (PD + 1)]) : "v"(ta)); break;
                    case 2: asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(xf[i % (PD + 1)]) : "v"(ta)); break;
                    default: asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(xf[i % (PD + 1)]) : "v"(ta)); break;
                }
            };
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // nothing else in the LGKM queue while waits are counted
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < PD; i++) request(i);
#pragma unroll
            for (int i = 0; i < NW; i++) {
                if (i + PD < NW) request(i + PD);
                const int rem = NW - 1 - i;                                        // reads requested after read i
                if (rem >= PD) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(PD) : "memory");
                else if (rem == 5) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
                else if (rem == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                else if (rem == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                else if (rem == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                else if (rem == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                acc = mfma16<H16>(w[i], xf[i % (PD + 1)], acc);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // The same block when its rows (block + halo) do not wrap around the ring: one base address, every tap / k16 slot an
        // immediate offset.  Same reads, same MFMA order as mma_slow.
        auto mma_fast = [&](auto dil_c) {
            constexpr int DIL = decltype(dil_c)::value;
            constexpr int PD = NW <= 6 ? NW : (NW >= 40 ? 4 : 6);
#pragma unroll
            for (int e = 0; e < 16; e++) acc[e] = 0.f;
            const unsigned base = lane_in + (unsigned)(rd_pos * P);
            uint4 xf[PD + 1];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            rs_prefetch<0, PD, KS, DIL, P>(xf, base);
            rs_mma_steps<0, NW, PD, KS, DIL, P, H16>(xf, w, base, acc);
        };
        auto mma_block = [&]() {
            constexpr int D0 = DP & 15, D1 = (DP >> 4) & 15, D2 = (DP >> 8) & 15;
            const int span = 32 + (NT - 1) * my_dil;
            if (rd_pos + span > Din) mma_slow();
            else if (kind) mma_fast(std::integral_constant<int, 1>{});
            else if (pair == 0) mma_fast(std::integral_constant<int, D0>{});
            else if (pair == 1) mma_fast(std::integral_constant<int, NPAIR >= 2 ? D1 : D0>{});
            else mma_fast(std::integral_constant<int, NPAIR >= 3 ? D2 : D0>{});
            rd_pos += R; if (rd_pos >= Din) rd_pos -= Din;
        };
        auto epilogue_block = [&](auto kind_c, auto masked_c) {
            constexpr int KIND = decltype(kind_c)::value;
            constexpr bool MASKED = decltype(masked_c)::value;                    // false: every row of the block lies inside the utterance
            const int g = g_out0 + l32;                                            // this lane's output row
            const bool inside = g >= 0 && g < len;                                 // streams are zero outside the utterance (every conv zero-pads ITS input)
            float4 bq[4];
            uint2 rq[4] = {};
            const unsigned ba = lds_addr0 + (unsigned)((unsigned char*)bias_l - lds) + (role * C + ct * 32 + h4) * 4;
            asm volatile("ds_read_b128 %0, %1" : "=v"(bq[0]) : "v"(ba));
            asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(bq[1]) : "v"(ba));
            asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(bq[2]) : "v"(ba));
            asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(bq[3]) : "v"(ba));
            if (KIND >= 1) {                                                       // x = the pair's input stream, same rows
                unsigned rp = rs_pos + l32; rp = min(rp, rp - (unsigned)Dres);
                const unsigned ra = lds_addr0 + res_off + rp * P + (ct * 32 + h4) * 2;
                asm volatile("ds_read_b64 %0, %1" : "=v"(rq[0]) : "v"(ra));
                asm volatile("ds_read_b64 %0, %1 offset:16" : "=v"(rq[1]) : "v"(ra));
                asm volatile("ds_read_b64 %0, %1 offset:32" : "=v"(rq[2]) : "v"(ra));
                asm volatile("ds_read_b64 %0, %1 offset:48" : "=v"(rq[3]) : "v"(ra));
            }
            unsigned char* dst;
            if (KIND <= 1) { unsigned wp = wr_pos + l32; wp = min(wp, wp - (unsigned)Dout); dst = lds + out_off + wp * P + (ct * 32 + h4) * 2; }
            else dst = stw + l32 * 80 + h4 * 2;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x2 v01 = (f32x2){acc[4 * q], acc[4 * q + 1]} + (f32x2){bq[q].x, bq[q].y};
                f32x2 v23 = (f32x2){acc[4 * q + 2], acc[4 * q + 3]} + (f32x2){bq[q].z, bq[q].w};
                if (KIND >= 1) { v01 += inv_lrelu2(unpack16<H16>(rq[q].x), rinv); v23 += inv_lrelu2(unpack16<H16>(rq[q].y), rinv); }
                if (KIND <= 1) { v01 = lrelu2(v01, slope1); v23 = lrelu2(v23, slope1); }                  // T, or the next pair's activated input
                else if (!AM) { v01 = lrelu2(v01, oslope); v23 = lrelu2(v23, oslope); }                  // no running sum: the output activation is applied here
                uint2 pk;
                pk.x = pack16<H16>(v01.x, v01.y); pk.y = pack16<H16>(v23.x, v23.y);
                if (KIND <= 1 && MASKED && !inside) { pk.x = 0u; pk.y = 0u; }
                *(uint2*)(dst + q * 16) = pk;
            }
            if (KIND == 2) {
                pend = true; pend_g0 = g_out0;
                if (AM & 1) {                                                      // xs rows of this block: requested now, used by the store phase next step
#pragma unroll
                    for (int h = 0; h < 2; h++) xs[h] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, row_off(g_out0, h, a.lda), 0, 0));
                }
            }
            wr_pos += R; if (wr_pos >= Dout && Dout > 0) wr_pos -= Dout;
            rs_pos += R; if (rs_pos >= Dres) rs_pos -= Dres;
            g_out0 += R;
        };

        for (int s = 0; s < nsteps; s++) {
            RS_STAMP(0);
            const int blk = s - role;                                              // this role's block in this step
            if (role == 0) {
                const int target = need(s + RS_PF) + 1;
                while (issued < target) dma_block();
            }
            if (is_final && pend) { store_phase(); pend = false; RS_STAMP(9); }
            RS_STAMP(1);
            if (blk >= 0 && blk < nact) {
                mma_block();
                RS_STAMP(2);
                const bool interior = g_out0 >= 0 && g_out0 + 32 <= len;          // wave-uniform
                if (is_final) epilogue_block(std::integral_constant<int, 2>{}, std::false_type{});
                else if (!kind) { if (interior) epilogue_block(std::integral_constant<int, 0>{}, std::false_type{}); else epilogue_block(std::integral_constant<int, 0>{}, std::true_type{}); }
                else { if (interior) epilogue_block(std::integral_constant<int, 1>{}, std::false_type{}); else epilogue_block(std::integral_constant<int, 1>{}, std::true_type{}); }
            }
            RS_STAMP(3);
            if (role == 0) wait_landed(need(s + 1) + 1);                           // what role 0 reads in the next step has landed
            RS_STAMP(4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            RS_STAMP(5);
            asm volatile("s_barrier" ::: "memory");
            RS_STAMP(6);
        }
        if (is_final && pend) store_phase();
        if (role == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // surplus requests of the tail: landed before the next segment re-uses X0
        asm volatile("s_barrier" ::: "memory");
    }
#ifdef RS_PROFILE
    if (a.prof && blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 12; k++) a.prof[(wave * 16 + k)] = (long long)tacc[k], a.prof[wave * 16 + 14] = role, a.prof[wave * 16 + 15] = sub;
#endif
}

template <int C, int NT, int NPAIR, int RSPLIT, int AM, bool HAS_OUT, int DP, bool H16>
__global__ __launch_bounds__(128 * NPAIR * (C / 32) * RSPLIT) void resstream_kernel(const StreamArgs a) {
    constexpr int NR = 2 * NPAIR, NTL = C / 32, WPR = NTL * RSPLIT;
    if (H16) f16_saturate_mode();                                                  // f32 -> f16 converts clamp to +-65504 (mfma_util.h)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int role, sub;
    role_of_wave<NR, WPR>(wave, role, sub, (a.opt & 2) != 0);
    role = __builtin_amdgcn_readfirstlane(role); sub = __builtin_amdgcn_readfirstlane(sub);
    if (a.opt & 1) {
        const int pr = __builtin_amdgcn_readfirstlane(prio_of_wave<NR, WPR>(wave, (a.opt & 2) != 0));
        if (pr == 3) __builtin_amdgcn_s_setprio(3); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
    }
    if (role == 0) rs_role<C, NT, NPAIR, RSPLIT, AM, HAS_OUT, DP, H16, 0>(a, lds, lane, sub, wave);
    else if (role == 1) rs_role<C, NT, NPAIR, RSPLIT, AM, HAS_OUT, DP, H16, 1>(a, lds, lane, sub, wave);
    else if (NR > 2 && role == 2) rs_role<C, NT, NPAIR, RSPLIT, AM, HAS_OUT, DP, H16, (NR > 2 ? 2 : 0)>(a, lds, lane, sub, wave);
    else if (NR > 2 && role == 3) rs_role<C, NT, NPAIR, RSPLIT, AM, HAS_OUT, DP, H16, (NR > 2 ? 3 : 1)>(a, lds, lane, sub, wave);
    else if (NR > 4 && role == 4) rs_role<C, NT, NPAIR, RSPLIT, AM, HAS_OUT, DP, H16, (NR > 4 ? 4 : 0)>(a, lds, lane, sub, wave);
    else if (NR > 4) rs_role<C, NT, NPAIR, RSPLIT, AM, HAS_OUT, DP, H16, (NR > 4 ? 5 : 1)>(a, lds, lane, sub, wave);
}

// ------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------
static int ncu() { return persistent_cus(); }

template <int C, int NT, int NPAIR, int RSPLIT, int DP>
static bool launch_rs(StreamArgs& a, hipStream_t stream, bool dry_run) {
    constexpr int NTL = C / 32, WPR = NTL * RSPLIT, R = 32 * RSPLIT, P = 2 * C + 16, H2 = (NT - 1) / 2, NR = 2 * NPAIR;
    // ring sizes (rows): see the header comment and RsGeom (the kernel uses the same compile-time values)
    using G = RsGeom<C, NT, NPAIR, RSPLIT, DP>;
    for (int p = 0; p < NPAIR; p++) if (a.dil[p] != G::dil(p)) return false;
    const int h0 = G::h0;
    if (2 * h0 + R > 4 * RS_RD) return false;                                       // DMA lead assumed <= 4 blocks in flight
    for (int p = 0; p < NPAIR; p++) if (a.dil[p] * H2 + H2 > R) return false;       // the residual rows of a block lie within the producer's last two blocks
    a.dT = G::dT;
    a.dX0 = G::dX0;
    size_t rows = a.dX0 + (size_t)NPAIR * a.dT;
    a.dX[0] = 0;
    for (int p = 1; p < NPAIR; p++) { a.dX[p] = G::dX(p); rows += a.dX[p]; }
    const size_t lds = rows * P + (size_t)WPR * 32 * 80 + (size_t)NR * C * 4;
    if (lds > 160 * 1024) return false;
    // segments: about one per CU, never shorter than 2048 rows (pipeline fill and halo are paid per segment)
    long total = 0; (void)total;
    const long rows_all = (long)a.M * a.nbatch;
    const int nwg = ncu();                                                          // one persistent workgroup per CU
    int S = (int)((rows_all + nwg - 1) / nwg);
    // pipeline fill and halo are paid per segment (~6 R + Hsum rows of extra work), so big jobs keep segments >= 2048 rows; a job
    // that cannot fill the chip with those (single requests) takes shorter ones instead -- idle CUs cost more than redundant rows.
    // The result does not depend on the segmentation (every row is computed from the same inputs in the same order).
    if (S < 2048) {
        const int floor_rows = a.seg_min > 0 ? a.seg_min : 256;                     // seg_min > 0: another floor; < 0: round 2's rule (A/B)
        S = a.seg_min < 0 ? 2048 : (S < floor_rows ? floor_rows : S);
    }
    S = (S + R - 1) / R * R;
    a.S = S; a.nseg = (a.M + S - 1) / S;
    const int nsegs = a.nseg * a.nbatch;
    a.flops = 2.0 * 2.0 * NPAIR * (double)rows_all * C * C * NT;
    if (dry_run) return true;
    const dim3 grid(nsegs < nwg ? nsegs : nwg), block(64 * NR * WPR);
    const int am = a.accum ? a.accum_mode : 0;
#define RS_GO1(AM_, HO_, H_) do { auto kfn = resstream_kernel<C, NT, NPAIR, RSPLIT, AM_, HO_, DP, H_>; \
        if (!lds_opt_in((const void*)kfn)) return false; \
        if (g_rs_ev_start) hipExtLaunchKernelGGL(kfn, grid, block, lds, stream, g_rs_ev_start, g_rs_ev_stop, 0, a); \
        else hipLaunchKernelGGL(kfn, grid, block, lds, stream, a); return true; } while (0)
#define RS_GO(AM_, HO_) do { if (a.f16) RS_GO1(AM_, HO_, true); else RS_GO1(AM_, HO_, false); } while (0)
    if (a.out) { if (am == 0) RS_GO(0, true); if (am == 1) RS_GO(1, true); return false; }
    if (am == 2) RS_GO(2, false);
    if (am == 3) RS_GO(3, false);
#undef RS_GO
#undef RS_GO1
    return false;
}

int launch_resstream(StreamArgs a, hipStream_t stream, bool dry_run) {
    if (a.npair < 1 || a.npair > 3 || a.ldx % 8 || (a.out && a.ldo % 8) || (a.accum && a.lda % 8)) return -1;
    if (a.ldx != a.C) return -1;                                                    // DMA image assumes dense rows
    const int am = a.accum ? a.accum_mode : 0;
    if (!a.out && !(am & 2)) return -1;
    if (a.out && am >= 2) return -1;
    for (int p = 0; p < a.npair; p++) if (a.dil[p] < 1 || !a.W1[p] || !a.W2[p] || !a.b1[p] || !a.b2[p]) return -1;
    // the dilations are template constants (HiFi-GAN's ResBlock1 sets: 1, 3, 5 -- config.py / hifigan.py:49-56); other sets
    // take the per-pair path
    int dp = 0;
    for (int p = 0; p < a.npair; p++) { if (a.dil[p] > 15) return -1; dp |= a.dil[p] << (4 * p); }
#define RS_TRY(C_, NT_, NP_, RSP_, DP_) if (a.C == C_ && a.ntaps == NT_ && a.npair == NP_ && dp == DP_) return launch_rs<C_, NT_, NP_, RSP_, DP_>(a, stream, dry_run) ? (C_ == 32 ? 20 : 21) : -1
    // 12 waves (whole ResBlock, 168 registers per wave) where the weight fragments leave room; otherwise 8 waves (256
    // registers): the first two pairs as one chain, the last pair on its own with the rows split over more waves
    RS_TRY(32, 3, 3, 2, 0x531); RS_TRY(32, 7, 3, 2, 0x531); RS_TRY(32, 11, 3, 2, 0x531);
    RS_TRY(64, 3, 3, 1, 0x531);
    RS_TRY(64, 7, 2, 1, 0x31); RS_TRY(64, 7, 1, 2, 0x5);
    RS_TRY(64, 11, 2, 1, 0x31); RS_TRY(64, 11, 1, 2, 0x5);
#undef RS_TRY
    return -1;
}

}  // namespace zvx
