// pairstream.hip -- one HiFi-GAN ResBlock1 iteration (hifigan.py:51-55)
//         xt = lrelu(conv1_dilated(lrelu(x)) + b1);   x' = conv2(xt) + b2 + x
// as ONE streaming launch for the wide stage C = 128 (k = 3 / 7 / 11), where a convolution's weights (32 KiB x k) fit
// neither registers nor LDS and the unfused pair makes five HBM passes over the stage tensor.
//
// Structure (the LDS-ring pipeline of resstream.hip, re-cut for streamed weights):
//   * a persistent 8-wave workgroup per CU walks a SEGMENT of one utterance in steps of R = 128 rows;
//   * waves 0-3 are conv1, waves 4-7 conv2 (waves w and w+4 share a SIMD); every wave owns 32 output channels and ALL 128
//     rows of the step (4 accumulator tiles), so one weight fragment feeds 4 MFMAs and the per-wave weight stream is
//     256 B per MFMA, requested 8 fragments ahead into a register ring (counted vmcnt, no barrier in the K loop);
//   * x rows arrive by LDS-DMA into an XOR-swizzled ring (256-B rows, 16-B slot ^= row & 15: conflict-free ds_read_b128
//     without padding -- the ring has to share 160 KiB with T), one step ahead;  T = lrelu(conv1 + b1) is written by the
//     conv1 waves into a padded ring (272-B rows) that the conv2 waves read one step later;
//   * conv2's epilogue (+ b2, + x from global memory: an L2 / Infinity-Cache hit, x was fetched two steps earlier; running
//     sum; activation; 16-byte row stores built with v_permlane32_swap) is DEFERRED to the start of the next step, so on
//     every SIMD it runs beside the conv1 wave's MFMAs, and the conv1 wave's T epilogue runs beside conv2's MFMA tail;
//   * one s_barrier per step.  No halo rows are recomputed inside a segment; the tensor crosses HBM twice per pair.
// The two roles are separate loop nests (own accumulators, own weight ring): the ring registers are the targets of loads
// hipcc knows nothing about, so no control-flow join may ever make it copy them.
//
// Accumulation order (K-chunk, tap, k16) and every rounding equal the two conv-slab launches this replaces: bit-identical.
#include "mfma_util.h"
#include "zvx_kernels.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <type_traits>

namespace zvx {

#define PS_R 128          // rows per step
#define PS_WD 8           // weight fragments in flight per wave (= MFMA steps per loop iteration)
#define PS_XP 256         // X ring row pitch (swizzled, unpadded)
#define PS_TP 272         // T ring row pitch (256 B + 16 B pad)
#define PS_NDMA 8         // LDS-DMA pieces (4 rows each) per conv1 wave and step

// Development switches (tools/micro/ps_bench.hip): PS_EXP cuts pieces OUT of the kernel (results are wrong by design, only the
// timing means something): 1 no weight refills, 2 no x-fragment LDS reads, 4 no global traffic in conv2's epilogue, 8 no DMA
// in the loop, 16 no T writes, 32 no MFMAs.  PS_PROFILE: per-wave s_memtime totals of workgroup 0 -> a.prof.
#ifndef PS_EXP
#define PS_EXP 0
#endif
#ifndef PS_PD
#define PS_PD 2           // x fragments are requested this many MFMA steps ahead (4 register sets)
#endif
#ifndef PS_PRIO
#define PS_PRIO 1         // bit 0 / bit 1: conv2's / conv1's epilogue runs at raised wave priority (it competes with the partner's MFMA stream for issue slots)
#endif
#ifndef PS_RES_LDS
#define PS_RES_LDS 1      // conv2's residual rows come from the X ring (read one step before the block's main loop, while they are still there)
#endif                    // instead of a second trip to global memory
#ifndef PS_RPF
#define PS_RPF 2          // conv2: the running-sum rows of a block are requested 2: right BEHIND its main loop (a step before its epilogue; round 4),
#endif                    // 1: before its main loop (round 3: the loop's first counted weight wait then also waits for them -- in-order retirement), 0: in its epilogue
#ifndef PS_DMA_TAIL
#define PS_DMA_TAIL 1     // conv1: the X rows of the next block are requested AFTER the block's main loop (in front of its T epilogue, which covers
#endif                    // their latency) instead of between its MFMA steps.  The vector-memory counter retires in order: with the requests
                          // inside the loop, every counted wait for a weight fragment also waited for the DMA requests issued before it -- an HBM
                          // round trip per loop iteration (tools/micro: k = 3 conv1 main loop 660 k -> 319 k cycles without them)
#ifdef PS_PROFILE
#define PS_STAMP(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define PS_STAMP(k) do {} while (0)
#endif

// ROLE 0: conv1 (dilated; X ring -> T ring, issues the X DMA).  ROLE 1: conv2 (T ring -> HBM).
template <int ROLE, int NT, int AM, bool HAS_OUT, bool H16>
__device__ __forceinline__ void ps_role(const PairArgs& a, unsigned char* lds, const int lane, const int ct) {
    constexpr int R = PS_R, H2 = (NT - 1) / 2, NS = 8 * NT, WD = PS_WD;
    // DMA instructions per loop iteration: a fixed number (vmcnt counts stay uniform), enough for the real pieces to be issued
    // BEFORE the last iteration (whose waits then retire them: the rows are in LDS when the step's barrier is reached)
    constexpr int EPI = (ROLE == 0 && !PS_DMA_TAIL) ? (PS_NDMA + NT - 2) / (NT - 1) : 0;
    static_assert(ROLE == 1 || PS_DMA_TAIL || EPI * (NT - 1) >= PS_NDMA, "DMA slots");
    const int l32 = lane & 31, hi = lane >> 5;
    const int dil = a.dil, H1 = dil * H2, DX = a.DX, DT = a.DT, G0 = a.G0;
    // LDS byte addresses are used raw: the dynamic region starts at 0 (no static __shared__ in this kernel)
    const unsigned tbase = (unsigned)DX * PS_XP;              // byte offset of the T ring
    const float* const bias_l = (const float*)(lds + tbase + (unsigned)DT * PS_TP) + ROLE * 128;
    const unsigned scratch_l = tbase + (unsigned)DT * PS_TP + 1024;   // 1 KiB that surplus DMA instructions zero-fill

    // weight stream of this wave's 32-channel tile: NS fragments of 1 KiB in (K-chunk, tap, k16) order, re-read every step
    const unsigned char* const wq = (const unsigned char*)(ROLE ? a.W2 : a.W1) + (long)ct * NS * 1024;
    const unsigned lane16 = lane * 16;
    u32x4 wreg[WD] = {};
    auto wload = [&](int off, int slot) __attribute__((always_inline)) {                    // slot is a literal at every call site
        asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(wreg[slot]) : "v"(lane16), "s"(wq + off) : "memory");
    };
#pragma unroll
    for (int i = 0; i < WD; i++) wload(i * 1024, i);

    const float slope1 = a.slope1, rinv = a.res_inv_slope, oscale = a.out_scale, oslope = a.slope;
    f32x16 acc[4];
#ifdef PS_PROFILE
    unsigned long long tacc[4] = {0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif

    const int nsegs = a.nseg * a.nbatch;
    for (int seg = blockIdx.x; seg < nsegs; seg += gridDim.x) {
        const int b = seg / a.nseg, sj = seg - b * a.nseg;
        int len = a.len ? a.len[b] : a.M;
        len = __builtin_amdgcn_readfirstlane(len);
        const int seg0 = sj * a.S;
        if (seg0 >= len) continue;
        const int seg_end = min(seg0 + a.S, len);
        const int nb = (seg_end - seg0 + 2 * H2 + R - 1) / R;        // blocks per role
        const int c1_0 = seg0 - H2;                                  // first T row of conv1 block 0
        const int xrow0 = c1_0 - H1;                                 // X row at ring position 0
        const unsigned short* const Xb = (const unsigned short*)a.X + (long)b * a.x_bs;

        // ---- LDS-DMA of 4 X rows (one 1-KiB piece) starting at global row grow + 4*pc into ring position pos + 4*pc: lane -> row
        //      lane >> 4, LDS slot lane & 15, SOURCE slot (lane & 15) ^ (position & 15).  Rows outside [0, len) arrive as zeros;
        //      real == false: nothing is fetched, a scratch KiB is zero-filled (the instruction still counts in vmcnt).
        auto dma_rsrc = [&](int grow, bool real) __attribute__((always_inline)) {
            const unsigned long long pa = (unsigned long long)(Xb + (long)grow * a.ldx);
            int nrec = real ? (len - grow) * a.ldx * 2 : 0; if (nrec < 0) nrec = 0;
            return (i32x4){__builtin_amdgcn_readfirstlane((int)(unsigned)pa), __builtin_amdgcn_readfirstlane((int)((pa >> 32) & 0xffff)),
                           __builtin_amdgcn_readfirstlane(nrec), 0x00020000};
        };
        auto dma_piece = [&](const i32x4& rsrc, int grow, int pos, int pc, bool real) __attribute__((always_inline)) {
            int lv = lane;
            asm volatile("" : "+v"(lv));                                       // keep the lane offsets out of the loop-invariant set (registers)
            const int rr = lv >> 4, sl = lv & 15;
            int p = pos + 4 * pc; if (p >= DX) p -= DX;
            const int key = (p + rr) & 15;
            int voff = ((4 * pc + rr) * a.ldx + ((sl ^ key) << 3)) * 2;
            if (grow + 4 * pc + rr < 0) voff = -16;                             // rows before the utterance: out of range -> zeros
            const unsigned la = __builtin_amdgcn_readfirstlane(real ? (unsigned)p * PS_XP : scratch_l);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(la), "v"(voff), "s"(rsrc) : "memory", "m0");
        };

        // ---- ring state (wave-uniform) ----
        int x_rd = 0;                     // conv1: ring position of X row c1_0 + s*R - H1 (tap 0 of the block's first row)
        int x_wr = G0;                    // position the next DMA block is written to (G0 < DX)
        int gx = xrow0 + G0;              // its first global row
        int t_wr = 0;                     // conv1: T ring position of T row c1_0 + s*R
        int t_rd = DT - 2 * H2;           // conv2: T ring position of tap 0 of its block's first row
        int g1 = c1_0;                    // conv1: first T row of the block
        int g2 = seg0 - 2 * H2;           // conv2: first output row of the block whose epilogue is pending / next
        if (ROLE == 0) {
            const i32x4 rs0 = dma_rsrc(xrow0, true);
            for (int pc = ct; pc < G0 / 4; pc += 4) dma_piece(rs0, xrow0, 0, pc, true);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

        // One block: NS steps of (1 weight fragment x 4 row tiles) as NT iterations of 8.  u = K-chunk * NT + tap; every LDS read
        // is inline asm, requested PS_PD steps ahead (4 register sets) with counted lgkmcnt; the weight fragment of a step was requested
        // one iteration earlier (counted vmcnt: the WD - 1 younger fragments + the EPI DMA instructions every iteration issues).
        auto main_loop = [&](bool dma_real) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[j][e] = 0.f;
            unsigned bs[4], nbs[4], pn = 0, hkn = 0;
            // LDS byte offsets of this lane's rows for step group u (= K-chunk * NT + tap), in two parts so that the VALU work can be
            // spread over the gaps between MFMAs: the part common to the 4 row tiles, then one tile each
            auto bases_common = [&](int u) __attribute__((always_inline)) {
                const int kc = u >= NT ? 1 : 0, tap = u - kc * NT;
                if (ROLE == 0) {
                    pn = x_rd + l32 + tap * dil; pn = min(pn, pn - (unsigned)DX);
                    hkn = (((pn ^ hi) & 15u) << 4) ^ ((unsigned)kc << 7);
                } else {
                    pn = t_rd + l32 + tap; pn = min(pn, pn - (unsigned)DT);
                    hkn = tbase + hi * 16 + kc * 128;
                }
            };
            auto bases_tile = [&](int j, unsigned (&o)[4]) __attribute__((always_inline)) {
                unsigned pj = pn + 32 * j;
                if (ROLE == 0) { pj = min(pj, pj - (unsigned)DX); o[j] = (pj << 8) + hkn; }
                else { pj = min(pj, pj - (unsigned)DT); o[j] = pj * PS_TP + hkn; }
            };
            auto rd1 = [&](uint4& xf, unsigned o, int kk) __attribute__((always_inline)) {          // kk is a literal at every call site
                if (PS_EXP & 2) { asm volatile("" : "=v"(xf)); return; }
                if (PS_EXP & 64) {                                              // the read happens, the MFMAs keep consuming a never-written register
                    uint4 dummy;
                    if (ROLE == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(dummy) : "v"(o ^ (unsigned)(kk << 5)));
                    else asm volatile("ds_read_b128 %0, %1" : "=v"(dummy) : "v"(o + kk * 32));
                    asm volatile("" : "=v"(xf));
                    return;
                }
                if (ROLE == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(xf) : "v"(o ^ (unsigned)(kk << 5)));
                else {
                    switch (kk) {
                        case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(xf) : "v"(o)); break;
                        case 1: asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(xf) : "v"(o)); break;
                        case 2: asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(xf) : "v"(o)); break;
                        default: asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(xf) : "v"(o)); break;
                    }
                }
            };
            constexpr int PD = PS_PD;
            static_assert(PD == 1 || PD == 2, "prefetch distance");
            uint4 xs[4][4];                                                    // fragment sets: step i uses set i & 3
            const i32x4 rsd = PS_DMA_TAIL ? (i32x4){0, 0, 0, 0} : dma_rsrc(gx, dma_real);
            bases_common(0);
#pragma unroll
            for (int j = 0; j < 4; j++) bases_tile(j, bs);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // nothing else in the LGKM queue while waits are counted
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < PD; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) rd1(xs[i][j], bs[j], i);
            for (int u2 = 0; u2 < NT; u2++) {
                const int next_off = (u2 + 1 == NT ? 0 : (u2 + 1) * (WD * 1024));
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int u = 2 * u2 + h;
                    int un = u + 1; if (un == 2 * NT) un = 0;
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) {
                        const int i = h * 4 + kk;
                        // this step's weight fragment (requested one iteration ago) and fragment set (PD steps ago) have landed
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)" :: "n"(WD - 1 + EPI), "n"(4 * (PD - 1)) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        // the step's other instructions sit in the gaps BETWEEN its MFMAs: a wave that is alone on its SIMD (its partner
                        // is in an epilogue) otherwise leaves the matrix pipe idle while it issues them
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            if (PS_EXP & 32) asm volatile("" : "+v"(acc[j]) : "v"(wreg[i]), "v"(xs[i & 3][j]));
                            else acc[j] = mfma16<H16>(wreg[i], xs[i & 3][j], acc[j]);
                            __builtin_amdgcn_sched_barrier(0);
                            if (kk + PD < 4) rd1(xs[(i + PD) & 3][j], bs[j], kk + PD); else rd1(xs[(i + PD) & 3][j], nbs[j], kk + PD - 4);
                            if (kk == 0 && j == 3) bases_common(un);
                            if (kk == 1) bases_tile(j, nbs);
                            if (j == 3) {
                                if (!(PS_EXP & 1)) wload(next_off + i * 1024, i);          // the slot just consumed takes the fragment 8 steps on
                                if (ROLE == 0 && !PS_DMA_TAIL && !(PS_EXP & 8)) {
                                    // this iteration's DMA instructions, spread over its steps (pieces past PS_NDMA: surplus, zero-fill the scratch KiB).
                                    // ALL of them are issued before step 7: that step's wait then leaves exactly this iteration's 7 + EPI
                                    // requests outstanding, so after the last iteration every real piece (issued earlier) has landed.
                                    // (With a piece at step 7 -- (8 e + 4) / EPI for EPI = 4, NT = 3 -- the real piece of the previous
                                    // iteration's step 7 could still be in flight at the step barrier: seen as a rare wrong block, ~1 % of runs.)
#pragma unroll
                                    for (int e = 0; e < EPI; e++)
                                        if (i == (7 * e + 3) / EPI) {
                                            const int pidx = u2 * EPI + e;
                                            dma_piece(rsd, gx, x_wr, ct * PS_NDMA + (pidx < PS_NDMA ? pidx : 0), dma_real && pidx < PS_NDMA);
                                        }
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) bs[j] = nbs[j];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the speculative reads behind the last step
            __builtin_amdgcn_sched_barrier(0);
        };

        // conv1: T = lrelu(acc + b1) as bf16 into the T ring; zero outside the utterance (conv2 zero-pads ITS input, hifigan.py:39-44)
        auto epilogue_T = [&]() __attribute__((always_inline)) {
            int lv = lane;
            asm volatile("" : "+v"(lv));
            const int l32 = lv & 31, hi = lv >> 5;
            float4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) bq[q] = *(const float4*)(bias_l + ct * 32 + 8 * q + 4 * hi);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int g = g1 + 32 * j + l32;
                const bool inside = g >= 0 && g < len;
                unsigned wp = t_wr + 32 * j + l32; wp = min(wp, wp - (unsigned)DT);
                unsigned char* const dst = lds + tbase + wp * PS_TP + (ct * 32 + 4 * hi) * 2;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const f32x2 v01 = lrelu2((f32x2){acc[j][4 * q], acc[j][4 * q + 1]} + (f32x2){bq[q].x, bq[q].y}, slope1);
                    const f32x2 v23 = lrelu2((f32x2){acc[j][4 * q + 2], acc[j][4 * q + 3]} + (f32x2){bq[q].z, bq[q].w}, slope1);
                    uint2 pk;
                    pk.x = inside ? pack16<H16>(v01.x, v01.y) : 0u;
                    pk.y = inside ? pack16<H16>(v23.x, v23.y) : 0u;
                    if (!(PS_EXP & 16)) *(uint2*)(dst + q * 16) = pk;
                    else asm volatile("" :: "v"(pk.x), "v"(pk.y), "v"(dst));
                }
            }
        };

        // conv2: y = acc + b2 + x (x = inverse leaky-relu of the activated input, from global memory), running sum, activation.
        // v_permlane32_swap turns the MFMA layout (lane = row, 4-channel quads split over the wave halves) into 8 consecutive
        // channels per lane: 16-byte loads / stores, lanes l and l + 32 adjacent in a row.  All global accesses are raw buffer
        // operations on per-utterance descriptors: rows outside the segment get an out-of-range offset (loads return 0, stores
        // are dropped), so the whole epilogue is branch-free and every load of the block is in flight before the first use.
        int off[4];
        u32x4 rx[4][2], sx[4][2];
        // Residual x of conv2 block n = X rows seg0 - 2 H2 + n R ..: in the X ring from the end of step n - 1 until the DMA of step
        // n + 1 overwrites them, i.e. stable during step n.  They are read THEN (MFMA layout: lane = row, 4-channel quads) and kept in
        // registers until the block's epilogue at step n + 2: two sets alive (rcur: the pending epilogue's, rnext: the next one's).
        uint2 rcur[4][4], rnext[4][4];
        auto residual_reads = [&](int blk) __attribute__((always_inline)) {
            int lv = lane;
            asm volatile("" : "+v"(lv));
            unsigned p0 = (unsigned)((long)blk * R % DX) + H1 - H2 + (lv & 31);
            p0 = min(p0, p0 - (unsigned)DX);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                unsigned pj = p0 + 32 * j; pj = min(pj, pj - (unsigned)DX);
                const unsigned char* row = lds + pj * PS_XP + (lv >> 5) * 8;
#pragma unroll
                for (int q = 0; q < 4; q++) rnext[j][q] = *(const uint2*)(row + (((ct * 4 + q) ^ (pj & 15u)) << 4));
            }
        };
        auto res_rsrc = [&](const void* base, long bs_) __attribute__((always_inline)) {
            return __builtin_amdgcn_make_buffer_rsrc((void*)((unsigned short*)base + (long)b * bs_), 0, len * a.ldx * 2, 0x00020000);
        };
        auto epilogue_loads = [&](int gb) __attribute__((always_inline)) {
            int lv = lane;
            asm volatile("" : "+v"(lv));                                       // nothing below is hoisted out of the step loop
            const int cA = ct * 32 + 8 * (lv >> 5);
            const __amdgpu_buffer_rsrc_t rsX = res_rsrc(a.X, a.x_bs), rsA = res_rsrc(a.accum, a.a_bs);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int g = gb + 32 * j + (lv & 31);
                off[j] = (g >= seg0 && g < seg_end && !(PS_EXP & 4)) ? (g * a.ldx + cA) * 2 : (int)0x80000000;
                if (!PS_RES_LDS) {
                    rx[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rsX, off[j], 0, 0);
                    rx[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rsX, off[j] + 32, 0, 0);
                }
            }
            if (AM & 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    sx[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off[j], 0, 0);
                    sx[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off[j] + 32, 0, 0);
                }
            }
        };
        auto epilogue_out = [&]() __attribute__((always_inline)) {
            int lv = lane;
            asm volatile("" : "+v"(lv));
            float4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) bq[q] = *(const float4*)(bias_l + ct * 32 + 8 * q + 4 * (lv >> 5));
            const __amdgpu_buffer_rsrc_t rsA = res_rsrc(a.accum, a.a_bs), rsO = res_rsrc(a.out, a.o_bs);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v[16];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    v[4 * q] = acc[j][4 * q] + bq[q].x; v[4 * q + 1] = acc[j][4 * q + 1] + bq[q].y;
                    v[4 * q + 2] = acc[j][4 * q + 2] + bq[q].z; v[4 * q + 3] = acc[j][4 * q + 3] + bq[q].w;
                    if (PS_RES_LDS) {
                        const f32x2 r01 = inv_lrelu2(unpack16<H16>(rcur[j][q].x), rinv), r23 = inv_lrelu2(unpack16<H16>(rcur[j][q].y), rinv);
                        v[4 * q] += r01.x; v[4 * q + 1] += r01.y; v[4 * q + 2] += r23.x; v[4 * q + 3] += r23.y;
                    }
                }
#pragma unroll
                for (int pr = 0; pr < 2; pr++) {                                // quad pairs (0,1) -> channels cA .. cA+7, (2,3) -> cA+16 ..
                    float w8[8];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[8 * pr + e]), __float_as_uint(v[8 * pr + 4 + e]), false, false);
                        w8[e] = __uint_as_float(r[0]); w8[4 + e] = __uint_as_float(r[1]);
                    }
                    f32x2 t[4] = {(f32x2){w8[0], w8[1]}, (f32x2){w8[2], w8[3]}, (f32x2){w8[4], w8[5]}, (f32x2){w8[6], w8[7]}};
                    if (!PS_RES_LDS) {
                        const u32x4 rr = rx[j][pr];
                        t[0] += inv_lrelu2(unpack16<H16>(rr.x), rinv); t[1] += inv_lrelu2(unpack16<H16>(rr.y), rinv);
                        t[2] += inv_lrelu2(unpack16<H16>(rr.z), rinv); t[3] += inv_lrelu2(unpack16<H16>(rr.w), rinv);
                    }
                    if (AM & 1) {
                        const u32x4 ss = sx[j][pr];
                        t[0] += unpack16<H16>(ss.x); t[1] += unpack16<H16>(ss.y); t[2] += unpack16<H16>(ss.z); t[3] += unpack16<H16>(ss.w);
                    }
                    if (AM & 2)
                        __builtin_amdgcn_raw_buffer_store_b128((u32x4){pack16<H16>(t[0].x, t[0].y), pack16<H16>(t[1].x, t[1].y), pack16<H16>(t[2].x, t[2].y), pack16<H16>(t[3].x, t[3].y)},
                                                               rsA, off[j] + 32 * pr, 0, 0);
                    if (HAS_OUT) {
                        if (AM) {
#pragma unroll
                            for (int e = 0; e < 4; e++) t[e] *= oscale;
                        }
#pragma unroll
                        for (int e = 0; e < 4; e++) t[e] = lrelu2(t[e], oslope);
                        __builtin_amdgcn_raw_buffer_store_b128((u32x4){pack16<H16>(t[0].x, t[0].y), pack16<H16>(t[1].x, t[1].y), pack16<H16>(t[2].x, t[2].y), pack16<H16>(t[3].x, t[3].y)},
                                                               rsO, off[j] + 32 * pr, 0, 0);
                    }
                }
            }
        };

        // step s: conv1 block s | conv2: epilogue of block s - 2, then main loop of block s - 1
        for (int s = 0; s <= nb + 1; s++) {
            if (ROLE == 0) {
                if (s < nb) {
                    main_loop(s + 1 < nb);                                     // (PS_DMA_TAIL 0: fetches the X rows of block s + 1 on the way)
                    if (PS_DMA_TAIL && !(PS_EXP & 8) && s + 1 < nb) {          // the X rows of block s + 1: this wave's 8 pieces of 4 rows, landed by the step barrier
                        const i32x4 rsd = dma_rsrc(gx, true);
#pragma unroll
                        for (int pc = 0; pc < PS_NDMA; pc++) dma_piece(rsd, gx, x_wr, ct * PS_NDMA + pc, true);
                    }
                    PS_STAMP(0);
                    gx += R; x_wr += R; if (x_wr >= DX) x_wr -= DX;
                    if (PS_PRIO & 2) __builtin_amdgcn_s_setprio(2);
                    epilogue_T();
                    if (PS_PRIO & 2) __builtin_amdgcn_s_setprio(0);
                    PS_STAMP(1);
                    x_rd += R; if (x_rd >= DX) x_rd -= DX;
                    t_wr += R; if (t_wr >= DT) t_wr -= DT;
                    g1 += R;
                }
            } else {
                if (s >= 2) {
                    if (PS_PRIO & 1) __builtin_amdgcn_s_setprio(2);
                    if (!PS_RPF) epilogue_loads(g2);
                    epilogue_out();
                    if (PS_PRIO & 1) __builtin_amdgcn_s_setprio(0);
                    g2 += R;
                    PS_STAMP(1);
                }
                if (PS_RES_LDS) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
#pragma unroll
                        for (int q = 0; q < 4; q++) rcur[j][q] = rnext[j][q];
                    residual_reads(s);                                         // block s: its epilogue is two steps away
                }
                if (s >= 1 && s <= nb) {
                    if (PS_RPF == 1) epilogue_loads(g2);                       // rows of THIS block: in registers long before its epilogue (next step)
                    main_loop(false);
                    if (PS_RPF == 2) epilogue_loads(g2);                       // ... requested here: the step barrier and the partner's MFMAs cover their latency
                    t_rd += R; if (t_rd >= DT) t_rd -= DT;
                    PS_STAMP(0);
                }
            }
            if (PS_DMA_TAIL && ROLE == 0 && s < nb) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this step's DMA pieces (and the ring's first fragments of the next block) have landed
            if (s <= nb) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            PS_STAMP(2);
        }
    }
    // the ring registers stay allocated until the last requests have landed (hipcc would otherwise re-use them)
#pragma unroll
    for (int i = 0; i < WD; i++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(wreg[i]) :: "memory");
#ifdef PS_PROFILE
    if (a.prof && blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 4; k++) a.prof[(ROLE * 4 + ct) * 4 + k] = (long long)tacc[k];
#endif
}

template <int NT, int AM, bool HAS_OUT, bool H16>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void pairstream128_kernel(const PairArgs a) {
    if (H16) f16_saturate_mode();                                             // f32 -> f16 converts clamp to +-65504 (mfma_util.h)
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // biases [b1 | b2] live in LDS behind the rings (16 values per lane, re-read per block instead of held in registers)
    float* const bias_l = (float*)(lds + (unsigned)a.DX * PS_XP + (unsigned)a.DT * PS_TP);
    if (tid < 256) bias_l[tid] = tid < 128 ? a.b1[tid] : a.b2[tid - 128];
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // settled before any hidden load is in flight
    if (wave < 4) ps_role<0, NT, AM, HAS_OUT, H16>(a, lds, lane, wave);
    else ps_role<1, NT, AM, HAS_OUT, H16>(a, lds, lane, wave - 4);
}

static int ps_ncu() { return persistent_cus(); }

bool launch_pairstream(PairArgs a, hipStream_t stream, bool dry_run, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (a.C != 128 || a.ldx != a.C || a.dil < 1 || !a.W1 || !a.W2 || !a.b1 || !a.b2) return false;
    if (!(a.ntaps == 3 || a.ntaps == 7 || a.ntaps == 11)) return false;
    if ((a.out && a.ldo != a.ldx) || (a.accum && a.lda != a.ldx)) return false;     // one row offset serves x, xs and the output
    if ((long)a.M * a.ldx * 2 >= (1l << 31)) return false;                          // 32-bit buffer offsets within an utterance
    const int am = a.accum ? a.accum_mode : 0;
    if (!a.out && !(am & 2)) return false;
    if (a.out && am >= 2) return false;
    const int H2 = (a.ntaps - 1) / 2, H1 = a.dil * H2;
    a.G0 = (PS_R + 2 * H1 + 3) & ~3;
    a.DX = (a.G0 + PS_R + 15) & ~15;
    a.DT = 2 * PS_R + 2 * H2;
    const size_t lds = (size_t)a.DX * PS_XP + (size_t)a.DT * PS_TP + 1024 + 1024;   // rings + bias table + DMA scratch
    if (lds > 160 * 1024 || 4 * PS_NDMA * 4 != PS_R) return false;
    const long rows_all = (long)a.M * a.nbatch;
    const int nwg = ps_ncu();
    // small jobs (one or two utterances): 1024-row segments would leave most CUs without a workgroup and every workgroup with a
    // pipeline fill per handful of steps -- measured crossover against the two conv-slab launches at ~3 utterances of 896 frames
    if (rows_all < (long)nwg * 768 && !a.force) return false;
    int S = (int)((rows_all + nwg - 1) / nwg);
    const int smin = a.force == 2 ? 256 : 1024;                                     // force == 2: two-block segments for single requests (A/B)
    if (S < smin) S = smin;
    S = (S + PS_R - 1) / PS_R * PS_R;
    a.S = S; a.nseg = (a.M + S - 1) / S;
    if (dry_run) return true;
    const int nsegs = a.nseg * a.nbatch;
    const dim3 grid(nsegs < nwg ? nsegs : nwg), block(512);
#define PS_GO1(NT_, AM_, HO_, H_) do { auto kfn = pairstream128_kernel<NT_, AM_, HO_, H_>; \
        if (!lds_opt_in((const void*)kfn)) return false; \
        if (ev_start) hipExtLaunchKernelGGL(kfn, grid, block, lds, stream, ev_start, ev_stop, 0, a); \
        else hipLaunchKernelGGL(kfn, grid, block, lds, stream, a); return true; } while (0)
#define PS_GO(NT_, AM_, HO_) do { if (a.f16) PS_GO1(NT_, AM_, HO_, true); else PS_GO1(NT_, AM_, HO_, false); } while (0)
#define PS_MODE(NT_) do { if (a.out) { if (am == 0) PS_GO(NT_, 0, true); if (am == 1) PS_GO(NT_, 1, true); return false; } \
        if (am == 2) PS_GO(NT_, 2, false); if (am == 3) PS_GO(NT_, 3, false); return false; } while (0)
    if (a.ntaps == 3) PS_MODE(3);
    if (a.ntaps == 7) PS_MODE(7);
    if (a.ntaps == 11) PS_MODE(11);
#undef PS_MODE
#undef PS_GO
#undef PS_GO1
    return false;
}

}  // namespace zvx
