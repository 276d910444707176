// conv_h2g.hip -- 1x1 convolution (stride 1 or 2) on the f16x2 split as a streamed-K GEMM ("h2g", round 6): the Bottleneck
// 1x1 convs of ResNet-50's layers 2-4 (romp/lib/models/resnet_50.py:64-78, basic_modules.py Bottleneck = simple_romp/romp/
// model.py:85-123: C -> 4C, 4C -> C and the strided 1x1 `downsample`, C = 128 .. 512) and every other 1x1 with >= 32 input channels.
//
// What the generic kernels (conv_split.h) made of this class (profiles/r05_bench_resnet50.json): 0.29-0.35 of HBM and 0.13-0.18 of
// the matrix roof at once, the stride-2 downsamples 0.12 / 0.09 -- half of ResNet-50's forward.  They are 3x3 kernels run with one
// tap: a 16-channel stage is 3 MFMAs per block between two barriers, the weight slab goes through LDS, a 64-channel-wide work item
// re-reads every pixel Cout / 64 times, and the stride-2 form stages the full-resolution tile to use a quarter of it.
// Here a 1x1 conv is what it is, Out[pixels x Cout] = In[pixels x Cin] . W, with K = Cin streamed in stages of 32 * KSUB channels:
//   * a wave owns ONE 32-channel slice of the output and P 32-pixel blocks, as in conv_h2r.hip: its weight fragments of a stage
//     (2 * KSUB chunks x 2 pieces x 16 bytes per lane) live in REGISTERS, loaded straight from the packed layout
//     [cin/16][piece][k-half][cout] (the MFMA A-operand order: two contiguous 512-byte runs per fragment); chunk c's registers
//     are re-loaded for the next stage right after chunk c's last MFMA;
//   * pixels go through LDS by LDS-DMA in FULL 128-byte lines: a DMA piece (one wave instruction, 1 KB) is 8 pixels x the eight
//     16-byte H2 units of 32 channels -- eight consecutive lanes cover one pixel's whole line (round 5 traced the stride-2 3x3
//     layers' stage time to half-used lines).  LDS unit of (tile pixel t, unit w) = t * 8 + ((w + t) & 7): dense for the DMA (lane i
//     writes unit i and simply NAMES the rotated global unit), conflict-free for the fragment reads (16 consecutive pixels' same
//     unit cover every bank exactly twice);
//   * stride 2 reads only the pixels it uses (every other pixel of every other row: whole 128-byte lines each) -- no haloed tile;
//   * two stage buffers, ONE barrier per stage, the next stage's DMA pieces and weight reloads issued at the chunk ends inside the
//     MFMA stream (sched_group_barrier), the stage body one branch-free region; persistent workgroups on the per-XCD queues with
//     the channel slices of a pixel tile adjacent in queue order, so a tile comes from HBM once and from that XCD's L2 afterwards;
//   * H2 outputs (+ H2 residual) leave through the direct epilogue (conv_common.h: permlane32 swaps, 16-byte units), float32
//     outputs through the LDS-transposed one.
// Wave w of the 4: channel slice w % NS, pixel group w / NS; workgroup tile = (4 / NS) x P x 32 pixels x NS x 32 channels.
#include "conv_split.h"

namespace romp {

template <int KS, int P, int NS, int TW, int S, int KSUB>
struct GCfg {
    static constexpr int NWV = 4;
    static constexpr int TAPS = KS * KS;                       // KS = 2: one output parity of a ConvTranspose2d(k4, s2, p1) (resnet_plan.py): K = 4 taps x Cin
    static constexpr int PG = NWV / NS;                        // pixel groups (waves along the pixel dimension)
    using C = ConvCfg<KS, S, P, NS, TW, 16, PG>;               // TH = PG * P * (32 / TW) output rows, NW = NS * 32 channels
    static constexpr int NPIX = PG * P * 32;                   // output pixels of a tile
    static constexpr int NI = NPIX / 8 / NWV;                  // DMA pieces (8 pixels x 128 bytes) per wave and 32-channel sub-stage
    static constexpr int SUB_BYTES = NPIX * 128;               // one 32-channel sub-stage
    static constexpr int STAGE_BYTES = KSUB * SUB_BYTES;
    static constexpr int NCH = 2 * KSUB;                       // 16-channel chunks (MFMA K steps x 3 products) per stage
    static constexpr int SS_BYTES = NS * 256;                  // per slot: [slice][scale 32 | shift 32] floats
    static constexpr bool EPI_ALIAS = STAGE_BYTES >= NWV * EPI_WAVE;       // (float32 outputs) staging tiles inside the consumed stage buffer
    static constexpr int OFF_E = 2 * STAGE_BYTES;
    static constexpr int OFF_S = OFF_E + (EPI_ALIAS ? 0 : NWV * EPI_WAVE);
    static constexpr int LDS_BYTES = OFF_S + 2 * SS_BYTES + 16;
    static constexpr int G = P >= 2 ? 2 : 1;                   // blocks per unit: a unit = 2G fragment reads + 3G MFMAs
    static constexpr int PFU = 2;                              // fragment reads run PFU units ahead of their MFMAs
    static constexpr int PPE = KSUB * NI / NCH;                // DMA pieces issued per chunk end (= NI / 2)
    static_assert(NS == 1 || NS == 2 || NS == 4, "channel slices per workgroup");
    static_assert(KS == 1 || (KS == 2 && S == 1) || KS == 3, "1x1, the 2x2 parity conv, or 3x3 with one (tap, channel run) per stage");
    static_assert(NPIX % 32 == 0 && NI >= 2 && NI % 2 == 0, "the same number of whole pieces at every chunk end");
    static_assert((KSUB - 1) * SUB_BYTES + (P - 1) * 4096 < 65536, "fragment read offsets are ds_read immediates");
};

typedef __attribute__((address_space(3))) void lds_void_g;
typedef const __attribute__((address_space(1))) void glb_void_g;

struct GStage {                 // wave-uniform description of one stage's sources
    const float* in;            // image + group + first channel of the stage
    const uint4* wg;            // group + tap + chunk + this wave's channel slice of the split weights
    int oy0, ox0;               // output tile origin
    int dy, dx;                 // input pixel of output pixel (oy, ox): (oy * S + dy, ox * S + dx)  (1x1: 0, 0; 2x2: tap - padding)
    int c0;
};

template <int KS, int P, int NS, int TW, int S, int KSUB>
__global__ __launch_bounds__(256, 2) void conv_h2g_kernel(ConvParams p) {
    conv_args_now(p);
    if (p.dbg & 32) return;                            // ablation: launch cost only
    using X = GCfg<KS, P, NS, TW, S, KSUB>;
    using C = typename X::C;
    using frag = f16x8;
    constexpr int NWV = X::NWV, PG = X::PG, G = X::G, PFU = X::PFU, NCH = X::NCH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sBuf = reinterpret_cast<char*>(smem);
    char* sSb = sBuf + X::OFF_S;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = wave % NS, pg = wave / NS;                  // channel slice, pixel group
    const int li = lane & 31, lh = lane >> 5;
    const int q = p.n_queues == 8 ? (blockIdx.x & 7) : 0;
    const int spt = p.cin_pad / (32 * KSUB);                   // stages per tap
    const int n_stages = X::TAPS * spt;                        // stage st = tap st / spt, channels (st % spt) * 32 * KSUB ..: the packed weights' own order
    const int cin16 = p.cin_pad >> 4;
    char* sE = sBuf + X::OFF_E + wave * EPI_WAVE;

    const int nwg_q = gridDim.x / p.n_queues;
    const int j_cur0 = blockIdx.x / p.n_queues;
    if (j_cur0 >= p.per_queue) return;

    // ---- per-lane DMA descriptors of this wave's pieces (the same for every sub-stage): piece i = k * 4 + wave holds tile pixels
    // 8 i .. 8 i + 7; lane l writes LDS unit i * 64 + l = t * 8 + (l & 7) with t = 8 i + (l >> 3), and fetches the unit that belongs
    // there: w = ((l & 7) - t) & 7  (unit w = piece (w & 1) of octet (w >> 1) of the 32 channels)
    int d_rc[X::NI];                                           // output row | col << 8 | unit w << 17
#pragma unroll
    for (int k = 0; k < X::NI; ++k) {
        const int t = (k * NWV + wave) * 8 + (lane >> 3);
        const int w = ((lane & 7) - t) & 7;
        d_rc[k] = (t / TW) | ((t % TW) << 8) | (w << 17);
    }
    const int cold = (p.dbg & 1) ? 0 : 1;                      // ablation bit 1: every DMA piece reads the zero page (no HBM traffic)

    auto make_desc = [&](const Item& it, int st) {
        GStage d;
        const int tap = KS == 1 ? 0 : st / spt, c0 = (KS == 1 ? st : st % spt) * (32 * KSUB);
        d.in = p.in + (size_t)it.b * p.H * p.W * p.in_cs + p.in_co + it.g * p.in_gs + c0;
        d.wg = p.wh + (size_t)it.g * (X::TAPS * cin16 * 4 * p.cout_pad) + (size_t)st * (2 * KSUB * 4) * p.cout_pad + it.n0 + sl * 32;
        d.oy0 = it.ty * C::TH;
        d.ox0 = it.tx * TW;
        d.dy = KS == 1 ? 0 : tap / KS - p.pad_h;
        d.dx = KS == 1 ? 0 : tap % KS - p.pad_w;
        d.c0 = c0;
        return d;
    };
    auto issue_piece = [&](int ks, const GStage& d, int buf) {             // piece ks = sub-stage ks / NI, piece ks % NI of it
        const int sub = ks / X::NI, k = ks % X::NI;
        const int i = k * NWV + wave;                                      // wave-uniform
        int rc = d_rc[k];
        asm volatile("" : "+v"(rc));                                       // (opaque: keeps the per-piece address parts from being hoisted into VGPRs)
        const int row = rc & 255, col = (rc >> 8) & 255, w = (rc >> 17) & 7;
        const int oy = d.oy0 + row, ox = d.ox0 + col;
        const int iy = oy * S + d.dy, ix = ox * S + d.dx;
        int ok = (int)(oy < p.Ho) & (int)(d.c0 + sub * 32 + (w >> 1) * 8 < p.cin_valid) & cold;
        if (KS != 1) ok &= (int)((unsigned)iy < (unsigned)p.H) & (int)((unsigned)ix < (unsigned)p.W);      // (the taps' zero padding)
        const unsigned long long a_in = (unsigned long long)(d.in + ((iy * p.W + ix) * p.in_cs + sub * 32 + w * 4));
        const unsigned long long a = ok ? a_in : (unsigned long long)p.zero;
        __builtin_amdgcn_global_load_lds((glb_void_g*)a, (lds_void_g*)(sBuf + buf * X::STAGE_BYTES + sub * X::SUB_BYTES + i * 1024), 16, 0, 0);
    };
    // scale | shift of an item: wave s < NS fetches slice s, one dword per lane
    auto issue_ss = [&](const Item& it, int slot) {
        if (wave >= NS) return;
        const float* src = (lane < 32 ? p.scale_h : p.shift) + it.g * p.cout_pad + it.n0 + wave * 32 + (lane & 31);
        __builtin_amdgcn_global_load_lds((glb_void_g*)src, (lds_void_g*)(sSb + slot * X::SS_BYTES + wave * 256), 4, 0, 0);
    };
    // weight fragments of one 16-channel chunk: lane (li, lh) holds channel li of the slice, k-half lh
    frag wreg[NCH][2];
    const unsigned w_lane = (unsigned)(lh * p.cout_pad + li);
    const unsigned w_chunk = (unsigned)(4 * p.cout_pad), w_pc = (unsigned)(2 * p.cout_pad);   // unit strides of a chunk / a piece
    auto load_w = [&](const uint4*& wp, int c) {
        wreg[c][0] = __builtin_bit_cast(frag, wp[0]);
        wreg[c][1] = __builtin_bit_cast(frag, wp[w_pc]);
        wp += w_chunk;
    };
    // ---- fragment addresses of block 0 of this wave: tile pixel t0 = (pg * P) * 32 + li, unit w = 4 * (chunk & 1) + 2 * lh + piece;
    // block j adds j * 4096 (32 pixels; the rotation (w + t) & 7 does not change), sub-stage (chunk >> 1) adds SUB_BYTES
    int xa[2][2];
    {
        const int t0 = pg * P * 32 + li;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const int w = 4 * kc + 2 * lh + pc;
                xa[kc][pc] = (t0 * 8 + ((w + t0) & 7)) * 16;
            }
    }

    int tr_n = 0;
    constexpr int tr_wpw = NWV;
    ROMP_TRACE(1);
    Item cur = decode_item(p, q, j_cur0, C::NW);
    {
        const GStage d0 = make_desc(cur, 0);
#pragma unroll
        for (int k = 0; k < KSUB * X::NI; ++k) issue_piece(k, d0, 0);
        issue_ss(cur, 0);
        const uint4* wp0 = d0.wg + w_lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c) load_w(wp0, c);
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
        const bool last = ch + 1 == n_stages;
        // the stage to prefetch; a workgroup's final stage re-fetches itself (harmless, keeps the stage body branch-free)
        const GStage nd = make_desc(last ? (have_next ? nxt : cur) : cur, last ? (have_next ? 0 : ch) : ch + 1);
        const int nbuf = buf ^ 1;
        ROMP_TRACE(10);
        if (!(p.dbg & 8)) {
            const char* sA = sBuf + buf * X::STAGE_BYTES;
            constexpr int UPT = P / G, NUNIT = NCH * UPT;          // units per chunk, per stage
            frag xf[PFU + 1][G][2];
            auto read_x = [&](int u) {
                const int c = u / UPT, j0 = (u % UPT) * G;
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc)
                        xf[u % (PFU + 1)][g][pc] = *reinterpret_cast<const frag*>(sA + (c >> 1) * X::SUB_BYTES + xa[c & 1][pc] + (j0 + g) * 4096);
            };
            const uint4* wp = nd.wg + w_lane;                      // chunk c's registers take the next stage's chunk c at its end
#pragma unroll
            for (int u = 0; u < PFU && u < NUNIT; ++u) read_x(u);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NUNIT; ++u) {
                const int c = u / UPT, j0 = (u % UPT) * G;
                if (u + PFU < NUNIT) read_x(u + PFU);
                const frag (&x)[G][2] = xf[u % (PFU + 1)];
                // h1w2 + h2w1 + h1w1 (smallest terms first), product-major so that consecutive MFMAs hit different accumulators
#pragma unroll
                for (int g = 0; g < G; ++g) acc[j0 + g][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[c][1], x[g][0], acc[j0 + g][0], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < G; ++g) acc[j0 + g][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[c][0], x[g][1], acc[j0 + g][0], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < G; ++g) acc[j0 + g][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[c][0], x[g][0], acc[j0 + g][0], 0, 0, 0);
                const bool chunk_end = u % UPT == UPT - 1;
                if (chunk_end) {                                   // chunk done: its registers take the next stage's weights
                    load_w(wp, c);
#pragma unroll
                    for (int e = 0; e < X::PPE; ++e) issue_piece(c * X::PPE + e, nd, nbuf);
                }
                // the order inside the unit: its look-ahead reads, its MFMAs, the memory issues of a chunk end; units stay in order
                if (u + PFU < NUNIT) __builtin_amdgcn_sched_group_barrier(0x100, 2 * G, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * G, 0);
                if (chunk_end) __builtin_amdgcn_sched_group_barrier(0x010, 2 + X::PPE, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        ROMP_TRACE(11);
        if (last) {
            if (have_next) issue_ss(nxt, slot ^ 1);
            // The next stage's DMA pieces and weights (issued inside the stage body above) are waited for HERE, in front of the epilogue:
            // the item then ends on a barrier alone and its output stores stay in flight into the next item.  (Measured against the
            // full drain behind the epilogue, same box, all ten ResNet-50 shapes: equal to +-1 %, profiles/r06_h2g_drain_ab.txt --
            // with two workgroups per CU the other one covers the store round trip either way.  Kept: it never waits for a store.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(p.dbg & 4)) {
                Item ce = cur;
                ce.n0 += sl * 32;
                // (the lane index goes through an opaque move: otherwise hipcc hoists every lane-derived address part of the epilogue
                // out of the stage loop and holds them in VGPRs across the MFMA stages)
                int lane_e = lane;
                asm volatile("" : "+v"(lane_e));
                const float* sc_e = reinterpret_cast<const float*>(sSb + slot * X::SS_BYTES) + sl * 64;
                if (p.out_h2 && p.vec_io && (!p.res || p.res_h2) && !(p.dbg & 512)) conv_epilogue_h2direct<KS, S, P, TW, PG>(p, ce, acc, sc_e, pg, lane_e & 31, lane_e >> 5);
                else {
                    char* se = sE;
                    if (X::EPI_ALIAS) {                            // (a workgroup-uniform branch: every wave meets at this barrier)
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        se = sBuf + buf * X::STAGE_BYTES + wave * EPI_WAVE;
                    }
                    conv_epilogue<KS, S, P, 1, TW, 16, PG>(p, ce, acc, sc_e, se, pg, lane_e & 31, lane_e >> 5);
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
        if (ch == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (an item's last stage: drained in front of its epilogue)
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ROMP_TRACE(12);
    }
}

#define ROMP_CONV_VARIANT_H2G(KS, P, NS, TW, S, KSUB)                                                                \
    { KS, S, P, NS, TW, 32 * KSUB, conv_h2g_kernel<KS, P, NS, TW, S, KSUB>, GCfg<KS, P, NS, TW, S, KSUB>::LDS_BYTES, \
      GCfg<KS, P, NS, TW, S, KSUB>::C::TH, 0, 0, 11, 256 }

static ConvVariant kVariantsH2g[] = {
    // 128 pixels x 128 channels, 64 x 128, 256 x 64, 128 x 64 (64-channel outputs: layer1's conv1, HRNet's transition-free 1x1s)
    ROMP_CONV_VARIANT_H2G(1, 4, 4, 16, 1, 2), ROMP_CONV_VARIANT_H2G(1, 4, 4, 32, 1, 2), ROMP_CONV_VARIANT_H2G(1, 2, 4, 16, 1, 2),
    ROMP_CONV_VARIANT_H2G(1, 4, 2, 16, 1, 1), ROMP_CONV_VARIANT_H2G(1, 2, 2, 16, 1, 2), ROMP_CONV_VARIANT_H2G(1, 2, 1, 16, 1, 2),
    // 32-channel stages (Cin = 32 / 96 / 160 ..: not a multiple of 64)
    ROMP_CONV_VARIANT_H2G(1, 2, 4, 16, 1, 1), ROMP_CONV_VARIANT_H2G(1, 2, 2, 16, 1, 1), ROMP_CONV_VARIANT_H2G(1, 2, 1, 16, 1, 1),
    // the strided `downsample` convs (resnet_50.py:64-78)
    ROMP_CONV_VARIANT_H2G(1, 4, 4, 16, 2, 2), ROMP_CONV_VARIANT_H2G(1, 2, 4, 16, 2, 2), ROMP_CONV_VARIANT_H2G(1, 2, 2, 16, 2, 2),
    // the 2x2 parity convs of the three ConvTranspose2d(k4, s2, p1) layers (resnet_50.py:80-120): K = 4 taps x Cin, matrix-bound
    ROMP_CONV_VARIANT_H2G(2, 4, 4, 16, 1, 2), ROMP_CONV_VARIANT_H2G(2, 2, 4, 16, 1, 2), ROMP_CONV_VARIANT_H2G(2, 2, 2, 16, 1, 2), ROMP_CONV_VARIANT_H2G(2, 4, 2, 16, 1, 1),
    // 3x3 in the same form -- no haloed tile: stage = (tap, 32 * KSUB channels), every tap's pixels fetched on their own as full 128-byte
    // lines (9 instead of ~4.6 / ~1.7 fetched pixels per output pixel, out of L2, against half the weight bytes per MFMA of
    // conv_h2s / conv_h2r at P = 4).  VERDICT r5 #4's experiment: the stride-2 class on full lines; stride 1 rides along for the tuner
    ROMP_CONV_VARIANT_H2G(3, 4, 4, 16, 2, 2), ROMP_CONV_VARIANT_H2G(3, 2, 4, 16, 2, 2), ROMP_CONV_VARIANT_H2G(3, 2, 4, 16, 2, 1), ROMP_CONV_VARIANT_H2G(3, 2, 2, 16, 2, 2),
    ROMP_CONV_VARIANT_H2G(3, 2, 2, 16, 2, 1), ROMP_CONV_VARIANT_H2G(3, 2, 1, 16, 2, 1),
    ROMP_CONV_VARIANT_H2G(3, 2, 4, 16, 1, 2), ROMP_CONV_VARIANT_H2G(3, 4, 4, 16, 1, 2), ROMP_CONV_VARIANT_H2G(3, 2, 4, 16, 1, 4), ROMP_CONV_VARIANT_H2G(3, 2, 2, 16, 1, 2),
};
ConvVariant* conv_variants_h2g(int* n) { *n = (int)(sizeof(kVariantsH2g) / sizeof(kVariantsH2g[0])); return kVariantsH2g; }

}  // namespace romp
