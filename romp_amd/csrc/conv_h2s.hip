// conv_h2s.hip -- 3x3 STRIDE-2 convolution on the f16x2 split (conv_split.h) in the register-weight form of conv_h2r.hip ("h2s").
//
// Why a kernel of its own (round 4).  HRNet's stride-2 convs (stem conv2, the transitions, the fuse-layer chains model.py:198-221,
// the head's first conv) are HBM-bound by the roofline: 36-57 FLOP per byte against a ridge of 104.  Until now they ran on the
// stride-1 kernels' bodies with S = 2: pixels staged through registers (nine 16-byte loads per thread and stage), fragment reads at
// a two-pixel stride, one output-channel slice of 32 or 64 per work item so that a 32 -> 64 conv read its input twice and the three
// sibling convs of a fuse layer three times more: 94-160 TFLOP/s, 2.3-2.7 TB/s, measured / algorithmic traffic 1.2-1.4
// (profiles/r03_pmc_traffic_by_op.json).  Here:
//   * the haloed input tile goes to LDS by LDS-DMA (no staging registers), DE-INTERLEAVED into its four (row, column) PARITY PLANES on
//     the way -- the DMA writes LDS units in lane order whatever global address a lane names, so the de-interleave is free.  Tap
//     (dy, dx) of output pixel (r, c) is pixel (r + dy / 2, c + dx / 2) of plane (dy & 1, dx & 1): inside a plane every fragment read
//     is the stride-1 `ds_read_b128` of conv_h2r.hip, in the same rotated, conflict-free unit layout;
//   * a workgroup covers up to 128 output channels (NS = 4 channel slices of 32, one per wave; NS = 3 for the 96-channel merged convs:
//     three computing waves, the fourth only moves pixels -- the kernel is HBM-bound, the idle matrix pipe costs nothing): together
//     with plan.py merging the sibling convs of a fuse layer into one op (romp_op.relu_from) the input tensor is read ONCE;
//   * weights of a stage in registers, re-loaded tap by tap for the next stage, as in conv_h2r.hip.
// Wave w of the 4: channel slice w % NS, pixel group w / NS (NS = 3: wave 3 idle); workgroup tile = PG x P x 32 output pixels (TW = 16:
// 2 rows of 16 per block) x NS x 32 channels; LDS = two stage buffers of 4 planes x (TH + 1) rows x 80 units + the epilogue tiles.
#include "conv_split.h"

namespace romp {

template <int P, int NS, int TW>
struct SCfg {
    static constexpr int NWV = 4, CK = 16;
    static constexpr int PG = NS == 3 ? 1 : NWV / NS;          // pixel groups (waves along the pixel dimension)
    using C = ConvCfg<3, 2, P, NS, TW, CK, PG>;                // TH = PG * P * (32 / TW) output rows, NW = NS * 32 channels
    static constexpr int PR = C::TH + 1;                       // rows of a parity plane (odd-row planes use TH of them)
    static constexpr int PC = TW + 1;                          // columns of a parity plane (odd-column planes use TW)
    static constexpr int CG = (PC + 3) / 4;                    // 4-pixel column groups per plane row
    static constexpr int RSU = CG * 16;                        // 16-byte units per plane row
    static constexpr int PLANE_U = PR * RSU;                   // units per plane
    static constexpr int NI = ((4 * PLANE_U + 63) / 64 + NWV - 1) / NWV;   // DMA pieces (wave-instructions of 1 KiB) per wave and stage
    static constexpr int NA_I = NI * NWV;
    static constexpr int STAGE_BYTES = NA_I * 1024;
    static constexpr int SS_BYTES = NS * 256;                  // per slot: [slice][scale 32 | shift 32] floats
    static constexpr int OFF_E = 2 * STAGE_BYTES;              // epilogue staging tiles, one per wave
    static constexpr int OFF_S = OFF_E + NWV * EPI_WAVE;
    static constexpr int LDS_BYTES = OFF_S + 2 * SS_BYTES + 16;
    static constexpr int G = P >= 2 ? 2 : 1;                   // blocks per unit: a unit = 2G fragment reads + 3G MFMAs (accumulators alternate)
    static constexpr int PFU = 2;                              // fragment reads run PFU units ahead of their MFMAs
    static_assert(NS >= 1 && NS <= 4, "channel slices per workgroup");
    static_assert(NI <= 18, "at most two DMA pieces per tap");
    static_assert(4 * PLANE_U * 16 < 65536, "fragment read offsets are ds_read immediates");
};

typedef __attribute__((address_space(3))) void lds_void_s;
typedef const __attribute__((address_space(1))) void glb_void_s;

struct SStage {                 // wave-uniform description of one stage's sources
    const float* in;            // image + group + chunk base of the pixel tensor
    const uint4* wg;            // group + chunk + this wave's channel slice of the split weights
    int iy0, ix0;               // input coordinates of the halo origin (may be negative: zero padding)
    int c0;
};

template <int P, int NS, int TW>
__global__ __launch_bounds__(256, 2) void conv_h2s_kernel(ConvParams p) {
    conv_args_now(p);
    if (p.dbg & 32) return;                            // ablation: launch cost only
    using X = SCfg<P, NS, TW>;
    using C = typename X::C;
    using frag = f16x8;
    constexpr int NWV = X::NWV, PG = X::PG, G = X::G, PFU = X::PFU;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sBuf = reinterpret_cast<char*>(smem);
    char* sSb = sBuf + X::OFF_S;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool active = wave < NS * PG;                        // (NS = 3: the fourth wave only moves pixels)
    const int sl = active ? wave % NS : 0, pg = active ? wave / NS : 0;   // channel slice, pixel group
    const int li = lane & 31, lh = lane >> 5;
    const int q = p.n_queues == 8 ? (blockIdx.x & 7) : 0;
    const int n_chunks = p.cin_pad >> 4;
    const int cin16 = p.cin_pad >> 4;
    char* sE = sBuf + X::OFF_E + wave * EPI_WAVE;

    const int nwg_q = gridDim.x / p.n_queues;
    const int j_cur0 = blockIdx.x / p.n_queues;
    if (j_cur0 >= p.per_queue) return;

    // ---- per-lane DMA descriptors of this wave's pieces (the same for every stage).  LDS unit U of a stage buffer = plane (py, px),
    // plane row, and inside the row the rotated layout of conv_h2r.hip: unit of (col c, unit w of the chunk) = (c >> 2) * 16 + (c & 3) +
    // 4 * ((w + (c >> 2)) & 3).  It holds unit w of input pixel (2 row + py, 2 col + px) of the haloed tile.
    int d_rc[X::NI];                                           // halo row | halo col << 8 | inside-the-tile << 16 | unit w << 17
#pragma unroll
    for (int k = 0; k < X::NI; ++k) {
        const int U = (k * NWV + wave) * 64 + lane;
        const int plane = U / X::PLANE_U, rem = U % X::PLANE_U;
        const int prow = rem / X::RSU, r = rem % X::RSU;
        const int cg = r >> 4, r16 = r & 15;
        const int col = cg * 4 + (r16 & 3), w = ((r16 >> 2) - cg) & 3;          // unit w = piece (w & 1) of octet (w >> 1)
        const int hr = 2 * prow + (plane >> 1), hc = 2 * col + (plane & 1);
        d_rc[k] = hr | (hc << 8) | ((plane < 4 && hr < C::HR && hc < C::HC) ? 1 << 16 : 0) | (w << 17);
    }
    const int cold = (p.dbg & 1) ? 0 : 1;                      // ablation bit 1: every DMA piece reads the zero page (no HBM traffic)

    auto make_desc = [&](const Item& it, int c0) {
        SStage d;
        d.in = p.in + (size_t)it.b * p.H * p.W * p.in_cs + p.in_co + it.g * p.in_gs + c0;
        d.wg = p.wh + (size_t)it.g * (9 * cin16 * 4 * p.cout_pad) + (c0 >> 4) * 4 * p.cout_pad + it.n0 + sl * 32;
        d.iy0 = it.ty * C::TH * 2 - p.pad_h;
        d.ix0 = it.tx * TW * 2 - p.pad_w;
        d.c0 = c0;
        return d;
    };
    auto issue_piece = [&](int k, const SStage& d, int buf) {
        const int i = k * NWV + wave;                                      // wave-uniform
        int rc = d_rc[k];
        asm volatile("" : "+v"(rc));                                       // (opaque: keeps the per-piece address parts from being hoisted into VGPRs)
        const int row = rc & 255, col = (rc >> 8) & 255, w = (rc >> 17) & 3;
        const int iy = d.iy0 + row, ix = d.ix0 + col;
        const int ok = ((rc >> 16) & 1) & (int)((unsigned)iy < (unsigned)p.H) & (int)((unsigned)ix < (unsigned)p.W) &
                       (int)(d.c0 + (w >> 1) * 8 < p.cin_valid) & cold;
        const unsigned long long a_in = (unsigned long long)(d.in + ((iy * p.W + ix) * p.in_cs + w * 4));
        const unsigned long long a = ok ? a_in : (unsigned long long)p.zero;
        __builtin_amdgcn_global_load_lds((glb_void_s*)a, (lds_void_s*)(sBuf + buf * X::STAGE_BYTES + i * 1024), 16, 0, 0);
    };
    // scale | shift of an item: wave s < NS fetches slice s, one dword per lane
    auto issue_ss = [&](const Item& it, int slot) {
        if (wave >= NS) return;
        const float* src = (lane < 32 ? p.scale_h : p.shift) + it.g * p.cout_pad + it.n0 + wave * 32 + (lane & 31);
        __builtin_amdgcn_global_load_lds((glb_void_s*)src, (lds_void_s*)(sSb + slot * X::SS_BYTES + wave * 256), 4, 0, 0);
    };
    // weight fragments of one tap: lane (li, lh) holds channel li of the slice, k-half lh
    frag wreg[9][2];
    const unsigned w_lane = (unsigned)(lh * p.cout_pad + li);
    const unsigned w_tap = (unsigned)(cin16 * 4 * p.cout_pad), w_pc = (unsigned)(2 * p.cout_pad);   // unit strides of a tap / a piece
    auto load_w = [&](const uint4*& wp, int tap) {
        wreg[tap][0] = __builtin_bit_cast(frag, wp[0]);
        wreg[tap][1] = __builtin_bit_cast(frag, wp[w_pc]);
        wp += w_tap;
    };
    // ---- fragment addresses of block 0 of this wave inside a plane: output pixel (row, col) -> plane pixel (row, col + dxs), dxs = dx >> 1;
    // unit w = 2 * lh + piece.  Tap (dy, dx) and block j add the constant ((plane * PR + j * RPB + (dy >> 1)) * RSU) * 16
    int xa[2][2];
    {
        const int prow = pg * P * C::RPB + li / TW, pcol = li % TW;
#pragma unroll
        for (int dxs = 0; dxs < 2; ++dxs)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const int col = pcol + dxs, w = lh * 2 + pc;
                xa[dxs][pc] = (prow * X::RSU + (col >> 2) * 16 + (col & 3) + 4 * ((w + (col >> 2)) & 3)) * 16;
            }
    }

    int tr_n = 0;
    constexpr int tr_wpw = NWV;
    ROMP_TRACE(1);
    Item cur = decode_item(p, q, j_cur0, C::NW);
    {
        const SStage d0 = make_desc(cur, 0);
#pragma unroll
        for (int k = 0; k < X::NI; ++k) issue_piece(k, d0, 0);
        issue_ss(cur, 0);
        if (active) {
            const uint4* wp0 = d0.wg + w_lane;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) load_w(wp0, tap);
        }
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
        const SStage nd = make_desc(last ? (have_next ? nxt : cur) : cur, last ? (have_next ? 0 : ch * 16) : (ch + 1) * 16);
        const int nbuf = buf ^ 1;
        ROMP_TRACE(10);
        if (!active) {                                             // the pixel mover of an NS = 3 workgroup
#pragma unroll
            for (int k = 0; k < X::NI; ++k) issue_piece(k, nd, nbuf);
        } else if (!(p.dbg & 8)) {
            const char* sA = sBuf + buf * X::STAGE_BYTES;
            constexpr int UPT = P / G, NUNIT = 9 * UPT;            // units per tap, per stage
            frag xf[PFU + 1][G][2];
            auto read_x = [&](int u) {
                const int tap = u / UPT, j0 = (u % UPT) * G;
                const int dy = tap / 3, dx = tap % 3;
                const int plane = (dy & 1) * 2 + (dx & 1);
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc)
                        xf[u % (PFU + 1)][g][pc] =
                            *reinterpret_cast<const frag*>(sA + xa[dx >> 1][pc] + ((plane * X::PR + (j0 + g) * C::RPB + (dy >> 1)) * X::RSU) * 16);
            };
            const uint4* wp = nd.wg + w_lane;
#pragma unroll
            for (int u = 0; u < PFU; ++u) read_x(u);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NUNIT; ++u) {
                const int tap = u / UPT, j0 = (u % UPT) * G;
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
                if (tap_end) {                                     // tap done: its registers take the next stage's weights; up to two DMA pieces ride along
                    load_w(wp, tap);
                    if (tap < X::NI) issue_piece(tap, nd, nbuf);
                    if (tap + 9 < X::NI) issue_piece(tap + 9, nd, nbuf);
                }
                // the order inside the unit: its look-ahead reads, its MFMAs, the memory issues of a tap end; units stay in order
                if (u + PFU < NUNIT) __builtin_amdgcn_sched_group_barrier(0x100, 2 * G, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * G, 0);
                if (tap_end) {
                    if (tap + 9 < X::NI) __builtin_amdgcn_sched_group_barrier(0x010, 4, 0);
                    else if (tap < X::NI) __builtin_amdgcn_sched_group_barrier(0x010, 3, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        ROMP_TRACE(11);
        if (last) {
            if (have_next) issue_ss(nxt, slot ^ 1);
            if (active && !(p.dbg & 4)) {
                Item ce = cur;
                ce.n0 += sl * 32;
                int lane_e = lane;
                asm volatile("" : "+v"(lane_e));               // (see conv_h2r.hip: keeps the epilogue's address parts out of the stage loop)
                const float* sc_e = reinterpret_cast<const float*>(sSb + slot * X::SS_BYTES) + sl * 64;
                if (p.out_h2 && p.vec_io && (!p.res || p.res_h2) && !(p.dbg & 512)) conv_epilogue_h2direct<3, 2, P, TW, PG>(p, ce, acc, sc_e, pg, lane_e & 31, lane_e >> 5);
                else conv_epilogue<3, 2, P, 1, TW, 16, PG>(p, ce, acc, sc_e, sE, pg, lane_e & 31, lane_e >> 5);
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

#define ROMP_CONV_VARIANT_H2S(P, NS, TW)                                                               \
    { 3, 2, P, NS, TW, 16, conv_h2s_kernel<P, NS, TW>, SCfg<P, NS, TW>::LDS_BYTES,                     \
      SCfg<P, NS, TW>::C::TH, 0, 0, 9, 256 }

// (P = 2, NS = 2 needs 2 x 48 KB of stage buffers: one workgroup per CU; on offer for the deep-channel layers all the same)
static ConvVariant kVariantsH2s[] = {
    ROMP_CONV_VARIANT_H2S(2, 4, 16), ROMP_CONV_VARIANT_H2S(2, 3, 16), ROMP_CONV_VARIANT_H2S(1, 2, 16), ROMP_CONV_VARIANT_H2S(2, 2, 16),
    ROMP_CONV_VARIANT_H2S(1, 4, 16), ROMP_CONV_VARIANT_H2S(1, 1, 16),
    // (measured and dropped: P = 4 -- 8 x 16-pixel tiles, one workgroup per CU: 60 vs 49 us on the merged 32 -> 128 conv)
};
ConvVariant* conv_variants_h2s(int* n) { *n = (int)(sizeof(kVariantsH2s) / sizeof(kVariantsH2s[0])); return kVariantsH2s; }

}  // namespace romp
