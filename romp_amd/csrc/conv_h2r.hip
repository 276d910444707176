// conv_h2r.hip -- 3x3 stride-1 convolution on the f16x2 split (conv_split.h), fourth generation ("h2r"): the weights of a
// stage live in REGISTERS, only pixels go through LDS.
//
// What rounds 1-2 measured on the first three generations: per 16-channel stage a 128-pixel x 64-channel
// workgroup tile pulls 37 KB of weight slab and 12 KB of pixels through L1 into LDS, reads 54 fragments per wave back out of
// LDS for 54 MFMAs, and pays two barriers; all of it serialised with the matrix work.  Every re-arrangement of that same work
// (LDS-DMA weight rows, whole-stage DMA pipelines, producer / consumer waves) ran into the same ~40 us per 0.3-GFLOP layer.
// This generation changes the work itself:
//   * a wave owns ONE 32-channel slice of the output and P 32-pixel blocks.  Its slice's weight fragments of the stage
//     (9 taps x 2 pieces x 16 bytes per lane = 72 VGPRs) are loaded straight from global memory into registers -- the packed
//     layout [tap][cin/16][piece][k-half][cout] already is the MFMA A-operand order, a wave reads two contiguous 512-byte
//     runs per fragment -- and are re-used for all P blocks: no weight slab in LDS, no weight fragment reads, a third fewer
//     LDS bytes per MFMA.  The registers of tap t are re-loaded for the NEXT stage as soon as tap t's last MFMA has been
//     issued, so the prefetch costs no registers of its own and has a whole stage to land;
//   * LDS holds nothing but two pixel stage buffers (haloed tile x 16 channels, pre-split H2 units copied by LDS-DMA in the
//     rotated layout described at the descriptors below) and the per-wave epilogue staging tiles: 46-67 KB, two workgroups per CU with room for
//     a kernel of another HRNet branch stream;
//   * a stage is 9 x P / 2 units of [4 fragment reads, 6 MFMAs on alternating accumulators] per wave, the reads software-
//     pipelined two units ahead through a register ring, the DMA pieces of the next stage and the weight reloads issued at the
//     tap ends; ONE barrier per stage; the stage body is one branch-free scheduling region (a workgroup's final stage
//     re-fetches its own inputs instead of testing for "no next stage"; the in-image test of a DMA piece is a mask and a select
//     -- a short-circuit && made hipcc wrap the address arithmetic into nested exec-mask branches that cut the MFMA stream
//     into one region per tap).
// Measured and dropped (round 3, profiles/r03_h2r_notes.md): a three-buffer ring with the pixel DMA two stages ahead, weight
// loads as inline asm with hand-counted vmcnt (hipcc copies asm-loaded registers whose load is still in flight whenever a
// tied wait has more than one site) and the residual prefetched before the last stage -- not faster than this form in any class.
// Wave w of the 4: channel slice w % NS, pixel group w / NS; workgroup tile = (4 / NS) x P x 32 pixels x NS x 32 channels.
#include "conv_split.h"

namespace romp {

// KSUB (round 4): 16-channel SUB-stages per barrier interval.  KSUB = 2 ("ck32"): a stage is 32 input channels = two sub-stage
// buffers side by side, the weights of sub-stage 1 re-loaded tap by tap under sub-stage 0 (still 72 registers), ONE barrier and one
// full memory drain per 32 channels instead of two -- the stage skeleton (drain + barrier + prefetch set-up) was ~1 000 of a stage's
// ~5 500 cycles in the phase traces.
template <int P, int NS, int TW, int KSUB = 1>
struct RCfg {
    static constexpr int NWV = 4, CK = 16;
    static constexpr int PG = NWV / NS;                        // pixel groups (waves along the pixel dimension)
    using C = ConvCfg<3, 1, P, NS, TW, CK, PG>;                // TH = PG * P * (32 / TW) rows, NW = NS * 32 channels
    static constexpr int CG = (C::HC + 3) / 4;                 // 4-pixel column groups per haloed row
    static constexpr int RSU = CG * 16;                        // 16-byte units per haloed row
    static constexpr int NI = ((C::HR * RSU + 63) / 64 + NWV - 1) / NWV;   // DMA pieces (wave-instructions of 1 KiB) per wave and stage
    static constexpr int NA_I = NI * NWV;                      // ... per stage: piece k of wave w is instruction k * 4 + w; units past
                                                               // the haloed tile are zero-filled padding (no "is there a piece" branch)
    static constexpr int SUB_BYTES = NA_I * 1024;              // one 16-channel sub-stage
    static constexpr int STAGE_BYTES = KSUB * SUB_BYTES;
    static constexpr int SS_BYTES = NS * 256;                  // per slot: [slice][scale 32 | shift 32] floats
    // epilogue staging tiles of the LDS-transposed epilogue (float32 outputs; H2 outputs take the direct one), one per wave: in the
    // stage buffer the item has just consumed when that is big enough (after a barrier: every wave done reading it), else a region
    // of their own
    static constexpr bool EPI_ALIAS = STAGE_BYTES >= NWV * EPI_WAVE;
    static constexpr int OFF_E = 2 * STAGE_BYTES;
    static constexpr int OFF_S = OFF_E + (EPI_ALIAS ? 0 : NWV * EPI_WAVE);
    static constexpr int LDS_BYTES = OFF_S + 2 * SS_BYTES + 16;
    static constexpr int G = P >= 2 ? 2 : 1;                   // blocks per unit: a unit = 2G fragment reads + 3G MFMAs (accumulators alternate)
    static constexpr int PFU = 2;                              // fragment reads run PFU units ahead of their MFMAs
    static_assert(NS == 1 || NS == 2 || NS == 4, "channel slices per workgroup");
    static_assert(KSUB * NI <= KSUB * 9, "at most one DMA piece per tap end");
    static_assert((C::HR - 1) * RSU * 16 + RSU * 16 < 65536, "fragment read offsets are ds_read immediates");
};

typedef __attribute__((address_space(3))) void lds_void_r;
typedef const __attribute__((address_space(1))) void glb_void_r;

struct RStage {                 // wave-uniform description of one stage's sources
    const float* in;            // image + group + chunk base of the pixel tensor
    const uint4* wg;            // group + chunk + this wave's channel slice of the split weights
    int iy0, ix0;               // tile origin (may be negative: zero padding)
    int c0;
};

template <int P, int NS, int TW, int KSUB = 1>
__global__ __launch_bounds__(256, 2) void conv_h2r_kernel(ConvParams p) {
    conv_args_now(p);
    if (p.dbg & 32) return;                            // ablation: launch cost only
    using X = RCfg<P, NS, TW, KSUB>;
    using C = typename X::C;
    using frag = f16x8;
    constexpr int NWV = X::NWV, PG = X::PG, G = X::G, PFU = X::PFU;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sBuf = reinterpret_cast<char*>(smem);
    char* sSb = sBuf + X::OFF_S;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = wave % NS, pg = wave / NS;                  // channel slice, pixel group
    const int li = lane & 31, lh = lane >> 5;
    const int q = p.n_queues == 8 ? (blockIdx.x & 7) : 0;
    const int n_chunks = (p.cin_pad >> 4) / KSUB;              // stages: KSUB x 16 input channels each
    const int cin16 = p.cin_pad >> 4;
    char* sE = sBuf + X::OFF_E + wave * EPI_WAVE;

    const int nwg_q = gridDim.x / p.n_queues;
    const int j_cur0 = blockIdx.x / p.n_queues;
    if (j_cur0 >= p.per_queue) return;

    // ---- per-lane DMA descriptors of this wave's pieces (the same for every stage): (row, col, unit) of the 16-byte unit a lane
    // fetches.  LDS unit of (row, col c, unit w of the chunk) = row * RSU + (c >> 2) * 16 + (c & 3) + 4 * ((w + (c >> 2)) & 3):
    // dense for the DMA (lane i writes unit i), conflict-free for the fragment reads.
    int d_rc[X::NI];                                           // row | col << 8 | inside-the-tile << 16 | unit w << 17
#pragma unroll
    for (int k = 0; k < X::NI; ++k) {
        const int U = (k * NWV + wave) * 64 + lane;
        const int row = U / X::RSU, r = U % X::RSU;
        const int cg = r >> 4, r16 = r & 15;
        const int col = cg * 4 + (r16 & 3), w = ((r16 >> 2) - cg) & 3;          // unit w = piece (w & 1) of octet (w >> 1)
        d_rc[k] = row | (col << 8) | ((row < C::HR && col < C::HC) ? 1 << 16 : 0) | (w << 17);
    }
    const int cold = (p.dbg & 1) ? 0 : 1;                      // ablation bit 1: every DMA piece reads the zero page (no HBM traffic)
    const int wcold = (p.dbg & 64) ? 0 : 1;                    // ablation bit 64: every weight fragment load re-reads the FIRST tap's 2 KB of the
                                                               // wave's slice (same instruction stream, L1 hits: no weight traffic from L2)

    auto make_desc = [&](const Item& it, int c0) {
        RStage d;
        d.in = p.in + (size_t)it.b * p.H * p.W * p.in_cs + p.in_co + it.g * p.in_gs + c0;
        d.wg = p.wh + (size_t)it.g * (9 * cin16 * 4 * p.cout_pad) + wcold * (c0 >> 4) * 4 * p.cout_pad + it.n0 + sl * 32;
        d.iy0 = it.ty * C::TH - p.pad_h;
        d.ix0 = it.tx * TW - p.pad_w;
        d.c0 = c0;
        return d;
    };
    auto issue_piece = [&](int ks, const RStage& d, int buf) {             // piece ks = sub-stage ks / NI, piece ks % NI of it
        const int sub = ks / X::NI, k = ks % X::NI;
        const int i = k * NWV + wave;                                      // wave-uniform
        int rc = d_rc[k];
        asm volatile("" : "+v"(rc));                                       // (opaque: keeps the per-piece address parts from being hoisted into VGPRs)
        const int row = rc & 255, col = (rc >> 8) & 255, w = (rc >> 17) & 3;
        const int iy = d.iy0 + row, ix = d.ix0 + col;
        const int ok = ((rc >> 16) & 1) & (int)((unsigned)iy < (unsigned)p.H) & (int)((unsigned)ix < (unsigned)p.W) &
                       (int)(d.c0 + sub * 16 + (w >> 1) * 8 < p.cin_valid) & cold;
        const unsigned long long a_in = (unsigned long long)(d.in + ((iy * p.W + ix) * p.in_cs + sub * 16 + w * 4));
        const unsigned long long a = ok ? a_in : (unsigned long long)p.zero;
        __builtin_amdgcn_global_load_lds((glb_void_r*)a, (lds_void_r*)(sBuf + buf * X::STAGE_BYTES + sub * X::SUB_BYTES + i * 1024), 16, 0, 0);
    };
    // scale | shift of an item: wave s < NS fetches slice s, one dword per lane
    auto issue_ss = [&](const Item& it, int slot) {
        if (wave >= NS) return;
        const float* src = (lane < 32 ? p.scale_h : p.shift) + it.g * p.cout_pad + it.n0 + wave * 32 + (lane & 31);
        __builtin_amdgcn_global_load_lds((glb_void_r*)src, (lds_void_r*)(sSb + slot * X::SS_BYTES + wave * 256), 4, 0, 0);
    };
    // weight fragments of one tap: lane (li, lh) holds channel li of the slice, k-half lh
    frag wreg[9][2];
    const unsigned w_lane = (unsigned)(lh * p.cout_pad + li);
    const unsigned w_tap = (unsigned)(wcold * cin16 * 4 * p.cout_pad), w_pc = (unsigned)(2 * p.cout_pad);   // unit strides of a tap / a piece
    // `wp` walks the taps in order (a running per-lane pointer: two strides in SGPRs instead of eighteen hoisted offsets)
    auto load_w = [&](const uint4*& wp, int tap) {
        wreg[tap][0] = __builtin_bit_cast(frag, wp[0]);
        wreg[tap][1] = __builtin_bit_cast(frag, wp[w_pc]);
        wp += w_tap;
    };
    // ---- fragment addresses of block 0 of this wave: pixel (row, col + dx), unit w = 2 * lh + piece; block j and tap row dy
    // add the constant (j * RPB + dy) * RSU * 16
    int xa[3][2];
    {
        const int prow = pg * P * C::RPB + li / TW, pcol = li % TW;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const int col = pcol + dx, w = lh * 2 + pc;
                xa[dx][pc] = (prow * X::RSU + (col >> 2) * 16 + (col & 3) + 4 * ((w + (col >> 2)) & 3)) * 16;
            }
    }

    int tr_n = 0;
    constexpr int tr_wpw = NWV;
    ROMP_TRACE(1);
    Item cur = decode_item(p, q, j_cur0, C::NW);
    const uint4* cwg;                                          // weights of the stage being computed (sub-stage 0)
    {
        const RStage d0 = make_desc(cur, 0);
#pragma unroll
        for (int k = 0; k < KSUB * X::NI; ++k) issue_piece(k, d0, 0);
        issue_ss(cur, 0);
        const uint4* wp0 = d0.wg + w_lane;
        cwg = d0.wg;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) load_w(wp0, tap);
    }
    ROMP_TRACE(2);

    f32x16 acc[P][1];
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][0][r] = 0.f;

    int j_next = j_cur0 + nwg_q;
    Item nxt = cur;
    bool have_next = j_next < p.per_queue;
    if (have_next) nxt = decode_item(p, q, j_next, C::NW);
    int ch = 0, buf = 0, slot = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    ROMP_TRACE(4);

#pragma unroll 1
    while (true) {
        const bool last = ch + 1 == n_chunks;
        // the stage to prefetch; a workgroup's final stage re-fetches itself (harmless, keeps the stage body branch-free)
        const RStage nd = make_desc(last ? (have_next ? nxt : cur) : cur, last ? (have_next ? 0 : ch * 16 * KSUB) : (ch + 1) * 16 * KSUB);
        const int nbuf = buf ^ 1;
        ROMP_TRACE(10);
        if (!(p.dbg & 8)) {
            const char* sA = sBuf + buf * X::STAGE_BYTES;
            constexpr int UPT = P / G, NUSUB = 9 * UPT, NUNIT = KSUB * NUSUB;   // units per tap, per sub-stage, per stage
            frag xf[PFU + 1][G][2];
            auto read_x = [&](int u) {
                const int sub = u / NUSUB, tap = (u % NUSUB) / UPT, j0 = (u % UPT) * G;
                const int dy = tap / 3, dx = tap % 3;
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc)
                        xf[u % (PFU + 1)][g][pc] =
                            *reinterpret_cast<const frag*>(sA + sub * X::SUB_BYTES + xa[dx][pc] + ((j0 + g) * C::RPB + dy) * (X::RSU * 16));
            };
            // a tap's registers are re-loaded right after its last MFMA of a sub-stage: with the weights of the NEXT sub-stage (the
            // same stage's second 16 channels: one chunk = 4 * cout_pad units further on), or of the next stage's first
            const uint4* wp = (KSUB == 2 ? cwg + wcold * 4 * p.cout_pad : nd.wg) + w_lane;
            const uint4* wp2 = nd.wg + w_lane;
#pragma unroll
            for (int u = 0; u < PFU; ++u) read_x(u);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NUNIT; ++u) {
                const int sub = u / NUSUB, tap = (u % NUSUB) / UPT, j0 = (u % UPT) * G;
                if (u + PFU < NUNIT) read_x(u + PFU);
                const frag (&x)[G][2] = xf[u % (PFU + 1)];
                // h1w2 + h2w1 + h1w1 (smallest terms first), product-major so that consecutive MFMAs hit different accumulators
#pragma unroll
                for (int g = 0; g < G; ++g) acc[j0 + g][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[tap][1], x[g][0], acc[j0 + g][0], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < G; ++g) acc[j0 + g][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[tap][0], x[g][1], acc[j0 + g][0], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < G; ++g) acc[j0 + g][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[tap][0], x[g][0], acc[j0 + g][0], 0, 0, 0);
                const bool tap_end = u % UPT == UPT - 1;
                const int te = sub * 9 + tap;                      // tap end number of the stage: DMA piece `te` of the next stage rides along
                if (tap_end) {                                     // tap done: its registers take the next sub-stage's weights
                    if (KSUB == 2 && sub == 1) load_w(wp2, tap); else load_w(wp, tap);
                    if (te < KSUB * X::NI) issue_piece(te, nd, nbuf);
                }
                // the order inside the unit: its look-ahead reads, its MFMAs, the memory issues of a tap end; units stay in order
                if (u + PFU < NUNIT) __builtin_amdgcn_sched_group_barrier(0x100, 2 * G, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * G, 0);
                if (tap_end) {
                    if (te < KSUB * X::NI) __builtin_amdgcn_sched_group_barrier(0x010, 3, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        cwg = nd.wg;
        ROMP_TRACE(11);
        if (last) {
            if (have_next) issue_ss(nxt, slot ^ 1);
            if (!(p.dbg & 4)) {
                Item ce = cur;
                ce.n0 += sl * 32;
                // (the lane index goes through an opaque move: otherwise hipcc hoists every lane-derived address part of the epilogue
                // out of the stage loop and holds them in VGPRs across the MFMA stages)
                int lane_e = lane;
                asm volatile("" : "+v"(lane_e));
                const float* sc_e = reinterpret_cast<const float*>(sSb + slot * X::SS_BYTES) + sl * 64;
                // H2 out (+ H2 residual): the direct epilogue (conv_common.h; ROMP_CONV_DEBUG=512 keeps the LDS-transposed one: A/B runs)
                if (p.out_h2 && p.vec_io && (!p.res || p.res_h2) && !(p.dbg & 512)) conv_epilogue_h2direct<3, 1, P, TW, PG>(p, ce, acc, sc_e, pg, lane_e & 31, lane_e >> 5);
                else {
                    char* se = sE;
                    if (X::EPI_ALIAS) {                            // (a workgroup-uniform branch: every wave meets at this barrier)
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        se = sBuf + buf * X::STAGE_BYTES + wave * EPI_WAVE;
                    }
                    conv_epilogue<3, 1, P, 1, TW, 16, PG>(p, ce, acc, sc_e, se, pg, lane_e & 31, lane_e >> 5);
                }
            }
#pragma unroll
            for (int j = 0; j < P; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][0][r] = 0.f;
            ROMP_TRACE(14);
            if (!have_next) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the self re-fetch must not outlive the workgroup's LDS
                break;
            }
            cur = nxt;
            slot ^= 1;
            ch = 0;
            j_next += nwg_q;
            have_next = j_next < p.per_queue;
            if (have_next) nxt = decode_item(p, q, j_next, C::NW);
        } else {
            ++ch;
        }
        buf ^= 1;
        // this wave's DMA pieces of the next stage have landed (and its weight registers); every wave is done reading the buffer
        // the stage after next will overwrite
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ROMP_TRACE(12);
    }
}

#define ROMP_CONV_VARIANT_H2R(P, NS, TW)                                                               \
    { 3, 1, P, NS, TW, 16, conv_h2r_kernel<P, NS, TW>, RCfg<P, NS, TW>::LDS_BYTES,                     \
      RCfg<P, NS, TW>::C::TH, 0, 0, 8, 256 }
#define ROMP_CONV_VARIANT_H2R32(P, NS, TW)                                                             \
    { 3, 1, P, NS, TW, 32, conv_h2r_kernel<P, NS, TW, 2>, RCfg<P, NS, TW, 2>::LDS_BYTES,               \
      RCfg<P, NS, TW, 2>::C::TH, 0, 0, 8, 256 }

static ConvVariant kVariantsH2r[] = {
    ROMP_CONV_VARIANT_H2R(2, 1, 16), ROMP_CONV_VARIANT_H2R(2, 2, 16), ROMP_CONV_VARIANT_H2R(2, 4, 16), ROMP_CONV_VARIANT_H2R(1, 4, 16),
    ROMP_CONV_VARIANT_H2R(2, 1, 32), ROMP_CONV_VARIANT_H2R(2, 2, 32), ROMP_CONV_VARIANT_H2R(1, 1, 16),
    ROMP_CONV_VARIANT_H2R32(2, 4, 16), ROMP_CONV_VARIANT_H2R32(2, 2, 16), ROMP_CONV_VARIANT_H2R32(2, 2, 32), ROMP_CONV_VARIANT_H2R32(2, 1, 16),
    // (round 4, measured and dropped: P = 4 -- four pixel blocks per wave, the same 72 weight registers per stage feeding twice the
    // MFMAs, one workgroup per CU with up to 512 registers: 42.6 vs 41.3 us on 128 -> 128 @32^2, 51.5 vs 43.2 us on 64 -> 64 @64^2;
    // the in-order wave alone on its SIMD hides less of its own latencies than two half-size ones: profiles/r04_s2_notes.md)
};
ConvVariant* conv_variants_h2r(int* n) { *n = (int)(sizeof(kVariantsH2r) / sizeof(kVariantsH2r[0])); return kVariantsH2r; }

}  // namespace romp
