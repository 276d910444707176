// conv_h2c.h -- a whole 64- or 32-channel BasicBlock (simple_romp/romp/model.py:54-83) in ONE kernel on the f16x2 split, in the
// ROW-PIPELINED form:   y = relu(bn2(conv3x3(relu(bn1(conv3x3(x))))) + x),   C -> C -> C channels, stride 1, H2 tensors in and out.
// Written for the 64-channel @64^2 class (the network's largest once conv_h2b.hip had fused the 32-channel blocks: 66 launches,
// 2.75 ms of a 12.2 ms forward at B = 32, 0.29 of the matrix roof as separate convs), then instantiated for 32 channels too, where
// it runs two workgroups per CU and replaced conv_h2b.hip's kernel in batch plans.  Same idea as there -- x read once (haloed),
// y written once, the intermediate m never leaves the CU, both convs' weights register-resident -- with the geometry the larger
// weights force:
//   * a wave owns a CHANNEL GROUP of 16 output channels of both convs: its share of the split weights is 2 x 72 registers per 32
//     input channels (C = 64: 288 per wave, one workgroup per CU, ONE wave per SIMD; a wave that owned pixels instead would
//     need all 2 x 576.  C = 32: 144 per wave, two workgroups per CU).  M = 16 channels means v_mfma_f32_16x16x32_f16: A = 16
//     channels x 32 input channels (weights, registers), B = 32 input channels x 16 pixels (an LDS fragment), D = 4 consecutive
//     channels of one pixel per lane.  C = 64: four channel groups, every wave walks all rows of the tile; C = 32: two channel groups
//     x two ROW GROUPS (m rows 0..4 / 5..9 in conv1, output rows 0..3 / 4..7 in conv2);
//   * tiles of 8 x 16 output pixels: 12 x 20 input halo (C = 64: 60 KB in LDS), 10 x 18 halo of m (48 KB);
//   * a pixel block is 16 pixels of ONE ROW, and input rows are walked top to bottom: the fragment of (input row R, column shift
//     dx) feeds the output rows R, R - 1, R - 2 (dy = 0, 1, 2) -- 2 LDS reads per 9 MFMAs -- and a row of m (of y) is complete two
//     input rows later, so its hand-over (finish) rides under the MFMAs of the following row: no block slots, no tail but the
//     last row.  The two edge columns of m (10 rows x {0, 17}) are two extra blocks, done first;
//   * LDS in PLANES: plane (octet o, piece) holds one 16-byte unit per pixel, planes a multiple of 16 units apart: the 16 lanes of
//     a ds_read_b128 group then read 16 different pixels' units at consecutive unit addresses whatever octets they are after
//     (bank group = unit address mod 16): conflict-free for every tap; one base register + immediates address every fragment;
//   * scale / shift of a lane's 4 channels live in registers (no tables); the residual is parked per lane in LDS as in
//     conv_h2b.hip; the next tile's halo arrives by raw-buffer LDS-DMA under conv2; the last output row of a tile is finished
//     under the next tile's first MFMAs.
// ConvParams as used here: in = x (H2), res = x, out = y (H2); w3 = conv1's weights REPACKED per channel group ([group C/16][tap 9]
// [k-chunk C/32][piece 2][lane 64] 16-byte units, plan.pack_h2_wave16), wh = conv2's; scale / w = conv1's f16x2 scale and shift
// (C floats each), scale_h / shift = conv2's; the geometry fields as for a conv.
#pragma once
#include "conv_split.h"
#include "conv_fuse.h"
#include <string.h>

namespace romp {

// Geometry for C channels.  C = 64: four channel groups of 16, each wave all rows (one workgroup per CU: 288 weight registers per
// wave).  C = 32 (the two-waves-per-SIMD form of the 32-channel block): two channel groups x two ROW groups -- wave (cg, t) does
// m rows 5 t .. 5 t + 4 of conv1 and output rows 4 t .. 4 t + 3 of conv2 -- 144 weight registers per wave, 71 KB of LDS: two
// workgroups per CU, each other's MFMAs covering each other's side work.
template <int C, bool STRIP = false>
struct RCfg {
    static constexpr int TH = 8, TW = 16;
    static constexpr int IR = TH + 4, IC = TW + 4;             // input halo 12 x 20
    static constexpr int MR = TH + 2, MC = TW + 2;             // intermediate halo 10 x 18
    static constexpr int XPL = IR * IC;                        // units per input plane: 240 (= 0 mod 16)
    static constexpr int MPL = 192;                            // units per m plane: 180 used, padded to 0 mod 16
    static constexpr int NPL = C / 4;                          // planes: C / 8 octets x {high, low}; plane = 2 * octet + piece
    static constexpr int NKC = C / 32;                         // 32-input-channel chunks (one MFMA's K)
    static constexpr int NCG = C / 16, NRG = 4 / NCG;          // channel groups of 16 output channels; row groups
    // STRIP (round 6, the halo-carrying form): a workgroup walks RUNS of vertically consecutive tiles; m rows 8, 9 of a tile ARE m rows
    // 0, 1 of the tile below it, so every tile but a run's first copies them (LDS to LDS) and its conv1 produces m rows R0 = 2 .. 9 only,
    // from input rows 2 .. 11: 8 rows of 16 + ONE edge block (rows 2 .. 9 x columns {0, 17}: all 16 lanes of it used) instead of 10 + 2.
    // A run's first tile computes m rows 0, 1 and their four edge pixels in a plain prologue in front of the same main loop.
    static constexpr int R0 = STRIP ? 2 : 0;                   // first m row (= first input row) of the main conv1
    static constexpr int MRT = STRIP ? TH : MR;                // m rows the main conv1 produces
    static constexpr int MRW = MRT / NRG, THW = TH / NRG;      // m rows / output rows per wave
    static constexpr int IRW = MRW + 2, MRW2 = THW + 2;        // input rows a wave walks in conv1; m rows in conv2
    static constexpr int NEB = STRIP ? 1 : 2 / NRG;            // edge blocks per wave (STRIP, C = 32: row group 0's waves only)
    static constexpr int NPIECE = NPL * XPL / 64;              // DMA pieces (64 units): 60 / 30
    static constexpr int NI = (NPIECE + 3) / 4;                // per wave: 15 / 8 (the last one only for waves 0, 1 when C = 32)
    static constexpr int OFF_M = NPL * XPL * 16;
    static constexpr int OFF_R = OFF_M + NPL * MPL * 16;       // residual parking, [wave][row THW][piece 2][lane] x 8 bytes
    static constexpr int LDS_BYTES = OFF_R + 4 * 2 * THW * 64 * 8 + 64;
    static constexpr int WG_PER_CU = C == 32 ? 2 : 1;
    static_assert(C == 32 || C == 64, "32 or 64 channels");
    static_assert(XPL % 16 == 0 && MPL % 16 == 0 && MPL >= MR * MC, "planes a multiple of 16 units apart");
    static_assert(NPL * XPL % 64 == 0, "whole DMA pieces");
    static_assert(LDS_BYTES * WG_PER_CU <= 160 * 1024, "LDS of a CU");
};

typedef float f32x4c __attribute__((ext_vector_type(4)));

// DBG: timing knock-outs (env ROMP_CONV_DEBUG, wrong outputs): 1 no halo DMA, 2 no hand-over / parking, 4 no finish, 8 no MFMA --
// instantiated only in a developer build (-DROMP_BBLOCK_KNOCKOUTS: python -m romp_amd.build with extra_flags; scripts/bblock_bench.py),
// five more copies of a fully unrolled kernel are most of this file's compile time.  The product kernel (DBG = 0) ALWAYS counts the
// values it clamps at +-65504 on their way into fp16 pieces (round 6: conv_common.h sat_track_pk, one v_pk_maximum3_f16 per four
// values; rounds 4-5 had a separate "checked" instantiation that cost 1.7-2 % of the job, profiles/r06_guard_cost*.txt)
template <int C, int DBG, bool STRIP = false>
__global__ __launch_bounds__(256, RCfg<C>::WG_PER_CU) void bblockr_kernel(ConvParams p) {
    using X = RCfg<C, STRIP>;
    using frag = f16x8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sBuf = reinterpret_cast<char*>(smem);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void_f*)sBuf;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wv % X::NCG, tg = wv / X::NCG;              // this wave's channel group and row group
    const int px = lane & 15, q = lane >> 4;                   // B / D operand: pixel px of the block; A: channel px; k-quarter (D: channel quad) q
    // Every kernel argument the set-up reads, asked for in ONE scalar-memory round trip (hipcc otherwise loads each where its first
    // use is: five dependent s_load / s_waitcnt rounds in front of the first memory request), and this wave's weights -- the longest
    // transfer of the set-up, 72 / 36 KB -- requested at once, before the table arithmetic and the first halo: a wave's entry -> first
    // DMA took 3.5 us and the weights another 3.9 us behind it (profiles/r06s_trace3.txt).
    // Weights: 16 output channels x C input channels x 9 taps x 2 pieces of each conv.
    asm volatile("" :: "s"(p.trace), "s"(p.w3), "s"(p.wh), "s"(p.n_queues), "s"(p.per_queue), "s"(p.run_len), "s"(p.tiles_x), "s"(p.tiles_y),
                 "s"(p.H), "s"(p.W), "s"(p.Ho), "s"(p.Wo), "s"(p.in_cs), "s"(p.in), "s"(p.in_co), "s"(p.in_bytes), "s"(p.scale), "s"(p.w),
                 "s"(p.scale_h), "s"(p.shift), "s"(p.act_scale));
    frag w1[9][X::NKC][2], w2[9][X::NKC][2];                   // [tap][k-chunk of 32 input channels][piece]
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int kc = 0; kc < X::NKC; ++kc)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                w1[tap][kc][pc] = __builtin_bit_cast(frag, p.w3[(((cg * 9 + tap) * X::NKC + kc) * 2 + pc) * 64 + lane]);
                w2[tap][kc][pc] = __builtin_bit_cast(frag, p.wh[(((cg * 9 + tap) * X::NKC + kc) * 2 + pc) * 64 + lane]);
            }
    asm volatile("" ::: "memory");
    int tr_n = 0;                                              // phase stamps (ROMP_CONV_TRACE=1): 1 entry, 4 set-up done, per tile 11 conv1 MFMAs,
    constexpr int tr_wpw = 4;                                  // 13 last hand-over, 12 barrier, 17 conv2 MFMAs, 15 barrier
    ROMP_TRACE(1);
    const int qx = p.n_queues == 8 ? (blockIdx.x & 7) : 0;
    const int nwg_q = gridDim.x / p.n_queues;
    const int j0 = blockIdx.x / p.n_queues;
    if (j0 >= p.per_queue) return;
    // work items: tiles (every nwg_q-th of the queue), or -- STRIP -- RUNS of run_len vertically consecutive tiles: item t of the
    // queue's contiguous range is (image, segment of the column, tile column), tile columns fastest (the workgroups of an XCD work
    // on neighbouring columns of the same rows at the same time)
    const int n_items = (p.per_queue - j0 + nwg_q - 1) / nwg_q;
    const int n_mine = STRIP ? n_items * p.run_len : n_items;
    const int segs = STRIP ? p.tiles_y / p.run_len : 1;
    auto tile_of = [&](int k, int run, int kr) __attribute__((always_inline)) {
        if (!STRIP) return decode_item(p, qx, j0 + k * nwg_q, 32);
        int t = qx * p.per_queue + j0 + run * nwg_q;
        Item r;
        r.g = 0; r.n0 = 0;
        r.tx = t % p.tiles_x; t /= p.tiles_x;
        r.ty = (t % segs) * p.run_len + kr;
        r.b = t / segs;
        return r;
    };

    // scale / shift of this lane's 4 channels (16 wv + 4 q ..), PRE-MULTIPLIED by 2^act_shift: m and y are produced in the scaled
    // domain the H2 pieces live in (ReLU commutes with the positive factor; the residual's pieces are x * 2^act_shift already)
    f32x4c s1, b1, s2, b2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 16 * cg + 4 * q + i;
        s1[i] = p.scale[c] * p.act_scale;   b1[i] = p.w[c] * p.act_scale;
        s2[i] = p.scale_h[c] * p.act_scale; b2[i] = p.shift[c] * p.act_scale;
    }

    // ---- halo DMA: piece 4 k + wv is this wave's k-th; a lane's unit U = 64 piece + lane -> plane U / 240, pixel U % 240
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    constexpr bool PACKED = STRIP && C == 32;
    int d_rc[X::NI], d_off[X::NI];                             // row | col << 8;  byte offset from the halo origin
#pragma unroll
    for (int k = 0; k < X::NI; ++k) {
        const int U = (k * 4 + wv) * 64 + lane;
        const int plane = U / X::XPL, r = U % X::XPL;
        const int row = r / X::IC, col = r % X::IC;
        // (PACKED, the 32-channel strip form: ONE register per piece -- row | col << 8 | plane << 16 -- and the offset re-derived at each
        // use, ~8 full-rate VALU a piece under conv2's MFMAs: at 128 + 128 registers the kernel otherwise reloads spilled table entries
        // between its DMA pieces and waits on each for the piece before it to land.  The 64-channel form keeps the table: with
        // v_mul_lo_u32 -- quarter rate -- the re-derivation cost its conv2 8 %, profiles/r06s_trace.txt)
        d_rc[k] = row | (col << 8) | (PACKED ? plane << 16 : 0);
        d_off[k] = PACKED ? 0 : ((row * p.W + col) * p.in_cs + (plane >> 1) * 8 + (plane & 1) * 4) * 4;
    }
    i32x4_t rsrc;                                              // the input tensor as a raw buffer: offsets beyond num_records read zeros
    {
        const unsigned long long base = (unsigned long long)(p.in + p.in_co);
        rsrc[0] = (int)(unsigned)base;
        rsrc[1] = (int)(unsigned)(base >> 32) & 0xffff;
        rsrc[2] = (int)p.in_bytes;
        rsrc[3] = 0x00020000;
    }
    auto is_interior = [&](const Item& it) __attribute__((always_inline)) { return it.ty > 0 && it.ty < p.tiles_y - 1 && it.tx > 0 && it.tx < p.tiles_x - 1; };
    auto fetch_piece = [&](const Item& it, bool valid, bool interior, bool first, int kk) __attribute__((always_inline)) {
        if (DBG & 1) return;
        if (!valid || kk * 4 + wv >= X::NPIECE) return;        // (uniform)
        const int iy0 = it.ty * X::TH - 2, ix0 = it.tx * X::TW - 2;
        const int origin = ((it.b * p.H + iy0) * p.W + ix0) * p.in_cs * 4;     // may be "negative": the sum with d_off is not
        const unsigned dst = lds0 + (unsigned)((kk * 4 + wv) * 1024);
        int voff = d_off[kk] + origin;
        if (PACKED) {
            int rc = d_rc[kk];
            asm volatile("" : "+v"(rc));
            const unsigned row = rc & 255, col = (rc >> 8) & 255, plane = (unsigned)rc >> 16;
            const unsigned pix = __umul24(row, (unsigned)p.W) + col;           // (v_mad_u32_u24: every factor far below 2^24)
            voff = (int)((__umul24(pix, (unsigned)p.in_cs) + (plane >> 1) * 8 + (plane & 1) * 4) * 4) + origin;
        }
        if (!interior) {
            int rc = d_rc[kk];
            asm volatile("" : "+v"(rc));
            const int iy = iy0 + (rc & 255), ix = ix0 + ((rc >> 8) & 255);
            const int ok = (int)((unsigned)iy < (unsigned)p.H) & (int)((unsigned)ix < (unsigned)p.W);
            voff = ok ? voff : (int)0x80000000;
        }
        if (STRIP && !first) {                                 // a carrying tile starts at input row 2: rows 0, 1 are not fetched (zeros land)
            int rc = d_rc[kk];
            asm volatile("" : "+v"(rc));
            voff = (rc & 255) >= 2 ? voff : (int)0x80000000;
        }
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsrc), "s"(dst) : "memory");
    };

    // ---- LDS addresses (bytes).  Input plane unit (plane, row, col) = plane * 240 + row * 20 + col; m: OFF_M + plane * 192 + row * 18 + col
    // fragment of a row block: lane (px, q) reads octet 4 kc + q, piece pc of the pixel dx columns right of its own: base + immediates
    // (conv1's blocks are m columns 1..16 = input columns 1 + px + dx; conv2's are output columns px = m columns px + dx)
    const int xa = (2 * q * X::XPL + (X::R0 + tg * X::MRW) * X::IC + 1 + px) * 16;   // + ((8 kc + pc) * 240 + Rl * 20 + dx) * 16, Rl: the wave's local row
    const int ma = X::OFF_M + (2 * q * X::MPL + tg * X::THW * X::MC + px) * 16;   // + ((8 kc + pc) * 192 + Rl * 18 + dx) * 16
    // the two edge blocks of conv1: m pixels (row, col in {0, 17}); E0 rows 0..7 (lane px -> row px / 2, col 17 (px & 1)), E1 rows 8, 9 (px < 4)
    const int e_row = px >> 1, e_col = (px & 1) * 17;
    const bool e1_act = px < 4;
    const int xe0 = (2 * q * X::XPL + (X::R0 + e_row) * X::IC + e_col) * 16;               // + ((8 kc + pc) * 240 + dy * 20 + dx) * 16  (STRIP: THE edge block, m rows 2 .. 9)
    const int xe1 = (2 * q * X::XPL + (e1_act ? 8 + e_row : 8) * X::IC + (e1_act ? e_col : 0)) * 16;
    // hand-over stores: lane (px, q) holds channels 16 wv + 4 q .. + 3 = half (q & 1) of octet 2 wv + q / 2
    const int mo = 2 * cg + (q >> 1);
    const int hs = X::OFF_M + (2 * mo * X::MPL + (X::R0 + tg * X::MRW) * X::MC + 1 + px) * 16 + (q & 1) * 8;   // row block: + (pc * 192 + rl * 18) * 16
    const int hse0 = X::OFF_M + (2 * mo * X::MPL + (X::R0 + e_row) * X::MC + e_col) * 16 + (q & 1) * 8;   // + pc * 192 * 16  (E1: + 8 * 18 * 16)
    // the residual x of output pixel (r, px) in the input halo: pixel (r + 2, px + 2), same octet half
    const int ra = (2 * mo * X::XPL + (2 + tg * X::THW) * X::IC + 2 + px) * 16 + (q & 1) * 8;   // + (pc * 240 + rl * 20) * 16
    char* sR = sBuf + X::OFF_R + (wv * 2 * X::THW * 64 + lane) * 8;   // this lane's parking slots: + (rl * 2 + pc) * 512

    Item it = tile_of(0, 0, 0);
#pragma unroll
    for (int kk = 0; kk < X::NI; ++kk) fetch_piece(it, true, false, true, kk);
    ROMP_TRACE(2);
    // a "use" of every weight register in front of the tile loop: hipcc waits for these loads HERE, once; the halo DMAs (invisible to
    // it) are covered by the explicit wait.  conv1's weights are pinned to AGPRs (MFMA reads them there), conv2's stay in VGPRs.
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int kc = 0; kc < X::NKC; ++kc)
            asm volatile("" : "+a"(w1[tap][kc][0]), "+a"(w1[tap][kc][1]), "+v"(w2[tap][kc][0]), "+v"(w2[tap][kc][1]));
    ROMP_TRACE(3);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    ROMP_TRACE(4);

    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    auto pack_hi = [&](float a, float b) __attribute__((always_inline)) {
        const f32x2_t v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
    };
    // micro-steps [lo, hi) of N for MFMA number g of G: spread evenly
    auto share = [](int g, int G, int N, int& lo, int& hi) __attribute__((always_inline)) { lo = g * N / G; hi = (g + 1) * N / G; };
#define SIDE_PIN() __builtin_amdgcn_sched_barrier(0)

    f32x4c acc2[X::THW];                                        // (the last row's outlives its tile: finished under the next tile's first MFMAs)
    unsigned sat_pk = 0u;                                       // per-half maximum of the high pieces formed: a half == 0x7BFF iff a value was clamped (conv_common.h sat_track_pk)
    Item itp = it;
    int run = 0, kr = 0;                                       // (STRIP) this tile's run and its place in it
#pragma unroll 1
    for (int k = 0; k < n_mine; ++k) {
        const bool has_next = k + 1 < n_mine;
        const bool first = !STRIP || kr == 0;                  // (STRIP) a run's first tile: no m rows to take over
        const int kr_n = STRIP ? (kr + 1 == p.run_len ? 0 : kr + 1) : 0, run_n = STRIP ? run + (kr_n == 0) : 0;
        const bool next_first = !STRIP || kr_n == 0;
        const Item itn = has_next ? tile_of(k + 1, run_n, kr_n) : it;
        const bool next_interior = is_interior(itn);

        // ---- the finish of an output row: bn2 + x + ReLU in the scaled domain, split; lanes (px, q) and (px, q ^ 1) trade halves
        // (v_permlane16_swap) so that each stores one whole 16-byte unit.  9 micro-steps.
        uint2 e_rh, e_rl;
        float ev[4];
        unsigned eh[2], el[2];
        int e_o = 0;
        constexpr int FIN_N = 9;
        auto fin_micro = [&](const Item& tl, bool live, int r, int t) __attribute__((always_inline)) {
            if (DBG & 4) return;
            switch (t) {
            case 0:
                e_rh = *reinterpret_cast<const uint2*>(sR + (r * 2 + 0) * 512);
                e_rl = *reinterpret_cast<const uint2*>(sR + (r * 2 + 1) * 512);
                break;
            case 1: case 2: case 3: case 4: {
                const int e = t - 1;
                const unsigned wh = e < 2 ? e_rh.x : e_rh.y, wl = e < 2 ? e_rl.x : e_rl.y;
                const float v = fmaf(acc2[r][e], s2[e], b2[e]);
                ev[e] = (e & 1) ? add_pieces_relu<1>(v, wh, wl, H2_MAX) : add_pieces_relu<0>(v, wh, wl, H2_MAX);
                break; }
            case 5:
                eh[0] = pack_hi(ev[0], ev[1]); eh[1] = pack_hi(ev[2], ev[3]);
                if (live) sat_pk = sat_track_pk(sat_pk, eh[0], eh[1]);      // (not live: the first tile's pass over a row that does not exist)
                break;
            case 6: el[0] = h2_low_pair(eh[0], ev[0], ev[1]); el[1] = h2_low_pair(eh[1], ev[2], ev[3]); break;
            case 7: {
                // lanes (px, q even) and (px, q odd) hold channels .. + 0..3 and .. + 4..7 of an octet: after the swaps the even one
                // holds the octet's 8 high pieces, the odd one its 8 low pieces
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                const u32x2_t a = __builtin_amdgcn_permlane16_swap(eh[0], el[0], false, false);
                const u32x2_t b = __builtin_amdgcn_permlane16_swap(eh[1], el[1], false, false);
                eh[0] = a[0]; el[0] = a[1]; eh[1] = b[0]; el[1] = b[1];
                const int oy = tl.ty * X::TH + tg * X::THW + r, ox = tl.tx * X::TW + px;
                e_o = (oy * p.out_rs + ox * p.out_cs) + (16 * cg + 4 * q);
                break; }
            default: {
                float* o = p.out + (size_t)tl.b * p.out_bs + p.out_co + (unsigned)e_o;
                if (live) *reinterpret_cast<uint4*>(o) = make_uint4(eh[0], eh[1], el[0], el[1]);
                break; }
            }
        };
        // ---- residual parking: 16 copies (row r, piece pc) from the input halo, read at step i, written three steps later
        uint2 pk[4];
        constexpr int PARK_N = 2 * X::THW + 3;
        auto park_micro = [&](int t) __attribute__((always_inline)) {
            if (DBG & 2) return;
            if (t < 2 * X::THW) pk[t % 4] = *reinterpret_cast<const uint2*>(sBuf + ra + ((t & 1) * X::XPL + (t >> 1) * X::IC) * 16);
            if (t >= 3) *reinterpret_cast<uint2*>(sR + (t - 3) * 512) = pk[(t - 3) % 4];
        };
        // ---- the hand-over of a block of m (4 channels of one pixel per lane): bn1 + ReLU, zero outside the image (conv2's padding),
        // split, into the m planes.  6 micro-steps.  `inside`: is this lane's m pixel inside the image
        float hv[4];
        unsigned hh[2], hl[2];
        constexpr int HAND_N = 6;
        auto hand_micro = [&](const f32x4c& a, bool inside, bool act, int addr, int t) __attribute__((always_inline)) {
            if (DBG & 2) return;
            switch (t) {
            case 0: case 1: {
#pragma unroll
                for (int e = 2 * t; e < 2 * t + 2; ++e) {
                    const float v = h2_sat(fmaxf(fmaf(a[e], s1[e], b1[e]), 0.f));
                    hv[e] = inside ? v : 0.f;
                }
                break; }
            case 2:
                hh[0] = pack_hi(hv[0], hv[1]); hh[1] = pack_hi(hv[2], hv[3]);
                sat_pk = sat_track_pk(sat_pk, hh[0], hh[1]);       // (pixels outside the image are zeros)
                break;
            case 3: hl[0] = h2_low_pair(hh[0], hv[0], hv[1]); hl[1] = h2_low_pair(hh[1], hv[2], hv[3]); break;
            case 4: if (act) *reinterpret_cast<uint2*>(sBuf + addr) = make_uint2(hh[0], hh[1]); break;
            default: if (act) *reinterpret_cast<uint2*>(sBuf + addr + X::MPL * 16) = make_uint2(hl[0], hl[1]); break;
            }
        };
        const int iy_m0 = it.ty * X::TH - 1 + X::R0 + tg * X::MRW;   // image row of this wave's first m row
        // this wave's edge block(s): block e covers m rows 8 e + (px >> 1) (e = 1: lanes px < 4 only), columns {0, 17}
        auto edge_no = [&](int i) __attribute__((always_inline)) { return X::NEB == 2 ? i : tg; };   // (uniform)
        const int ix_e = it.tx * X::TW - 1 + e_col;             // image column of this lane's edge-block pixel

        // ================= 1. conv1
        f32x4c accE[X::NEB], acc1[X::MRW];
#pragma unroll
        for (int e = 0; e < X::NEB; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i) accE[e][i] = 0.f;
#pragma unroll
        for (int r = 0; r < X::MRW; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc1[r][i] = 0.f;
        if constexpr (!STRIP) {
            // fragment reads run PF units (a unit = the MFMAs fed by one fragment pair) ahead of their MFMAs.  Units 0 .. NUE - 1: the
            // edge block(s), unit (tap * NKC + kc) * NEB + e; then the row-block fragments of the wave's input rows,
            // NUE + ((Rl * 3 + dx) * NKC + kc)
            constexpr int PF = C == 32 ? 2 : 3, NUE = 9 * X::NKC * X::NEB, NU = NUE + X::IRW * 3 * X::NKC, RING = PF + X::NEB;
            frag xf[RING][2];
            const int xe = (edge_no(0) == 0 ? xe0 : xe1);
            auto read_x = [&](int u) __attribute__((always_inline)) {
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    if (u < NUE) {
                        const int e = u % X::NEB, kc = (u / X::NEB) % X::NKC, tap = u / (X::NEB * X::NKC);
                        const int base = X::NEB == 2 ? (e ? xe1 : xe0) : xe;
                        xf[u % RING][pc] = *reinterpret_cast<const frag*>(sBuf + base + ((8 * kc + pc) * X::XPL + (tap / 3) * X::IC + tap % 3) * 16);
                    } else {
                        const int v = u - NUE, Rl = v / (3 * X::NKC), dx = (v / X::NKC) % 3, kc = v % X::NKC;
                        xf[u % RING][pc] = *reinterpret_cast<const frag*>(sBuf + xa + ((8 * kc + pc) * X::XPL + Rl * X::IC + dx) * 16);
                    }
                }
            };
#pragma unroll
            for (int u = 0; u < PF; ++u) read_x(u);
            // (a) the edge block(s): no row reuse.  Under them: the previous tile's last output row, then this tile's residual parking
            constexpr int GE = NUE * 3, NE = FIN_N + PARK_N;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int kc = 0; kc < X::NKC; ++kc) {
                    const int u = (tap * X::NKC + kc) * X::NEB;    // units u .. u + NEB - 1: with two blocks their MFMAs take turns
#pragma unroll
                    for (int e = 0; e < X::NEB; ++e)
                        if (u + e + PF < NU) read_x(u + e + PF);
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int e = 0; e < X::NEB; ++e) {
                            const frag (&x)[2] = xf[(u + e) % RING];
                            if (!(DBG & 8))
                                accE[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[tap][kc][pr == 0 ? 1 : 0], x[pr == 1 ? 1 : 0], accE[e], 0, 0, 0);
                            const int g = ((tap * X::NKC + kc) * 3 + pr) * X::NEB + e;
                            int lo, hi;
                            share(g, GE, NE, lo, hi);
#pragma unroll
                            for (int t = lo; t < hi; ++t) { if (t < FIN_N) fin_micro(itp, k > 0, X::THW - 1, t); else park_micro(t - FIN_N); }
                            SIDE_PIN();
                        }
                }
            // (b) the row blocks, input row after input row (Rl: local to the wave's row group).  Under input row Rl: the hand-over of
            // the edge block(s) (Rl < NEB), then of m row Rl - 3 (complete since input row Rl - 1)
#pragma unroll
            for (int Rl = 0; Rl < X::IRW; ++Rl) {
                const int dy_lo = Rl - (X::MRW - 1) > 0 ? Rl - (X::MRW - 1) : 0, dy_hi = Rl < 2 ? Rl : 2;     // m rows Rl - dy in [0, MRW)
                const int nv = dy_hi - dy_lo + 1;
                const int G = 3 * X::NKC * nv * 3;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int kc = 0; kc < X::NKC; ++kc) {
                        const int u = NUE + (Rl * 3 + dx) * X::NKC + kc;
                        if (u + PF < NU) read_x(u + PF);
                        const frag (&x)[2] = xf[u % RING];
#pragma unroll
                        for (int pr = 0; pr < 3; ++pr)          // (products outside, rows inside: consecutive MFMAs on different accumulators)
#pragma unroll
                            for (int dy = dy_lo; dy <= dy_hi; ++dy) {
                                const int tap = dy * 3 + dx;
                                if (!(DBG & 8))
                                    acc1[Rl - dy] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[tap][kc][pr == 0 ? 1 : 0], x[pr == 1 ? 1 : 0], acc1[Rl - dy], 0, 0, 0);
                                const int g = (((dx * X::NKC + kc) * 3) + pr) * nv + (dy - dy_lo);
                                int lo, hi;
                                share(g, G, HAND_N, lo, hi);
#pragma unroll
                                for (int t = lo; t < hi; ++t) {
                                    if (Rl < X::NEB) {
                                        const int e = edge_no(Rl);
                                        const bool act = e == 0 || e1_act;
                                        const int iy = it.ty * X::TH - 1 + 8 * e + e_row;
                                        const bool in = act && (unsigned)iy < (unsigned)p.Ho && (unsigned)ix_e < (unsigned)p.Wo;
                                        hand_micro(accE[Rl], in, act, hse0 + e * (8 * X::MC * 16), t);
                                    } else if (Rl >= 3) {
                                        const int rl = Rl - 3;
                                        // (only a wave's first and last m row can lie outside the image: above it / below it)
                                        hand_micro(acc1[rl], rl == 0 ? iy_m0 >= 0 : true, true, hs + rl * X::MC * 16, t);
                                    }
                                }
                                SIDE_PIN();
                            }
                    }
            }
            ROMP_TRACE(11);
            if (DBG & 2) {                                     // (knock-out builds: keep every MFMA)
#pragma unroll
                for (int r = 0; r < X::MRW; ++r) asm volatile("" :: "v"(acc1[r]));
#pragma unroll
                for (int e = 0; e < X::NEB; ++e) asm volatile("" :: "v"(accE[e]));
            }
#pragma unroll
            for (int t = 0; t < HAND_N; ++t) hand_micro(acc1[X::MRW - 1], (unsigned)(iy_m0 + X::MRW - 1) < (unsigned)p.Ho, true, hs + (X::MRW - 1) * X::MC * 16, t);
            ROMP_TRACE(13);
        } else {
            // ======== the halo-carrying form (RCfg): prologue (a run's first tile) or copies (the others), ONE edge block, m rows 2 .. 9
            // units (a unit = the MFMAs fed by one fragment pair, read PF units ahead): 0 .. NUE - 1 the edge block's, tap * NKC + kc; then
            // the row units NUE + (Rl * 3 + dx) * NKC + kc.  NAP: accumulators of a block without row reuse (C = 64, one wave per SIMD: one
            // per product, consecutive MFMAs independent; C = 32: the SIMD's other wave fills the gaps, registers are short)
            constexpr int PF = C == 32 ? 2 : 3, NUE = 9 * X::NKC, NU = NUE + X::IRW * 3 * X::NKC, NAP = C == 32 ? 1 : 3;
            const bool do_edge = X::NRG == 1 || tg == 0;       // (uniform) C = 32: the edge block belongs to row group 0's waves
            // ---- 0. a run's first tile: m rows 0, 1 (C = 32: row `tg`) and their four edge pixels (C = 32: row group 1's waves), written
            // plainly (once per run; no side work to place: the tile's own side work rides under the edge block and the rows below)
            if (first) {
                constexpr int PRW = 2 / X::NRG;                // prologue rows per wave
                // (addresses derived HERE from the main loop's bases, behind an opaque copy: hoisted out of the tile loop they cost the
                // 32-channel kernel, 128 + 128 registers at two waves per SIMD, spills of its DMA tables)
                int xp = xa, hp = hs;
                asm volatile("" : "+v"(xp), "+v"(hp));
                xp += (tg * PRW - (X::R0 + tg * X::MRW)) * X::IC * 16;
                hp += (tg * PRW - (X::R0 + tg * X::MRW)) * X::MC * 16;
                int xq = xe0, hq = hse0;                      // the main edge block's pixel (R0 + e_row, e_col) -> (e_row, e_col): lanes px < 4 are m rows 0, 1
                asm volatile("" : "+v"(xq), "+v"(hq));
                xq -= X::R0 * X::IC * 16;
                hq -= X::R0 * X::MC * 16;
                // units: (input row Ri, dx, kc) of the row blocks, then (tap, kc) of the edge pixels' block (C = 32: both row groups' waves
                // compute it -- no divergent schedule -- row group 1's hand it over); fragments one unit ahead, pinned unit by unit
                // (scheduled freely every fragment read goes first: 100 registers)
                constexpr int NPR = (PRW + 2) * 3 * X::NKC, NP = NPR + 9 * X::NKC;
                f32x4c ap[PRW][NAP], ae[NAP];
#pragma unroll
                for (int pr = 0; pr < NAP; ++pr) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) ae[pr][i] = 0.f;
#pragma unroll
                    for (int r = 0; r < PRW; ++r)
#pragma unroll
                        for (int i = 0; i < 4; ++i) ap[r][pr][i] = 0.f;
                }
#ifdef ROMP_BBLOCK_PFP1                                        // (A/B build: one unit ahead at both channel counts, the first form)
                constexpr int PFP = 1;
#else
                constexpr int PFP = C == 64 ? 2 : 1;            // units ahead (64 channels: one wave per SIMD, a unit can be 3 MFMAs = 48 clocks)
#endif
                frag xb[PFP + 1][2];
                auto read_p = [&](int u) __attribute__((always_inline)) {
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        if (u < NPR) {
                            const int Ri = u / (3 * X::NKC), dx = (u / X::NKC) % 3, kc = u % X::NKC;
                            xb[u % (PFP + 1)][pc] = *reinterpret_cast<const frag*>(sBuf + xp + ((8 * kc + pc) * X::XPL + Ri * X::IC + dx) * 16);
                        } else {
                            const int kc = (u - NPR) % X::NKC, tap = (u - NPR) / X::NKC;
                            xb[u % (PFP + 1)][pc] = *reinterpret_cast<const frag*>(sBuf + xq + ((8 * kc + pc) * X::XPL + (tap / 3) * X::IC + tap % 3) * 16);
                        }
                    }
                };
#pragma unroll
                for (int u = 0; u < PFP; ++u) read_p(u);
                SIDE_PIN();
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    if (u + PFP < NP) read_p(u + PFP);
                    const frag (&x)[2] = xb[u % (PFP + 1)];
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr) {
                        if (u < NPR) {
                            const int Ri = u / (3 * X::NKC), dx = (u / X::NKC) % 3, kc = u % X::NKC;
#pragma unroll
                            for (int r = 0; r < PRW; ++r) {
                                const int dy = Ri - r;
                                if (dy >= 0 && dy <= 2)
                                    ap[r][pr % NAP] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[dy * 3 + dx][kc][pr == 0 ? 1 : 0], x[pr == 1 ? 1 : 0], ap[r][pr % NAP], 0, 0, 0);
                            }
                        } else {
                            const int kc = (u - NPR) % X::NKC, tap = (u - NPR) / X::NKC;
                            ae[pr % NAP] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[tap][kc][pr == 0 ? 1 : 0], x[pr == 1 ? 1 : 0], ae[pr % NAP], 0, 0, 0);
                        }
                    }
                    SIDE_PIN();
                }
#pragma unroll
                for (int r = 0; r < PRW; ++r) {
                    f32x4c a = ap[r][0];
#pragma unroll
                    for (int pr = 1; pr < NAP; ++pr)
#pragma unroll
                        for (int i = 0; i < 4; ++i) a[i] += ap[r][pr][i];
#pragma unroll
                    for (int t = 0; t < HAND_N; ++t)
                        hand_micro(a, (unsigned)(it.ty * X::TH - 1 + tg * PRW + r) < (unsigned)p.Ho, true, hp + r * X::MC * 16, t);
                    SIDE_PIN();
                }
                {
                    f32x4c a = ae[0];
#pragma unroll
                    for (int pr = 1; pr < NAP; ++pr)
#pragma unroll
                        for (int i = 0; i < 4; ++i) a[i] += ae[pr][i];
                    const bool mine = e1_act && (X::NRG == 1 || tg == 1);
                    const bool in = (unsigned)(it.ty * X::TH - 1 + e_row) < (unsigned)p.Ho && (unsigned)ix_e < (unsigned)p.Wo;
#pragma unroll
                    for (int t = 0; t < HAND_N; ++t)
                        hand_micro(a, in, mine, hq, t);
                    SIDE_PIN();
                }
            }
            // ---- the take-over of m rows 8, 9 as rows 0, 1 (every tile but a run's first), each wave the units it hands over itself
            // further down (its channel group's 4 planes): LDS operations of a wave execute in order, no barrier.  Interior: 4 planes x 2
            // rows x columns 1 .. 16 = 128 units, two per lane (C = 32: by row group 1's waves, who produce rows 6 .. 9); the edge
            // columns: 16 units, lanes 0 .. 15 of the wave that owns the edge block
            // unit addresses: interior  OFF_M + ((4 cg + (lane >> 5)) * 192 + (8 + ((lane >> 4) & 1)) * 18 + 1 + (lane & 15)) * 16, the second unit 2
            // planes on; edge  OFF_M + ((4 cg + (lane >> 2)) * 192 + (8 + ((lane >> 1) & 1)) * 18 + 17 * (lane & 1)) * 16  (computed in the step)
            // ONE step under one uniform branch, reads then writes: a value that lives across the conditional steps of a schedule this
            // tight was spilled to scratch (and came back behind an s_waitcnt vmcnt(0)); the step's LDS round trip is exposed, once a tile
            constexpr int CP_UP = 8 * X::MC * 16;
            auto copy_step = [&](bool interior, bool edge) __attribute__((always_inline)) {
                if (first) return;
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const int cpI = X::OFF_M + ((4 * cg + (ln >> 5)) * X::MPL + (8 + ((ln >> 4) & 1)) * X::MC + 1 + (ln & 15)) * 16;
                const int cpE = X::OFF_M + ((4 * cg + ((ln >> 2) & 3)) * X::MPL + (8 + ((ln >> 1) & 1)) * X::MC + 17 * (ln & 1)) * 16;
                uint4 c0, c1, ce;
                if (interior) {
                    c0 = *reinterpret_cast<const uint4*>(sBuf + cpI);
                    c1 = *reinterpret_cast<const uint4*>(sBuf + cpI + 2 * X::MPL * 16);
                }
                if (edge && lane < 16) ce = *reinterpret_cast<const uint4*>(sBuf + cpE);
                if (interior) {
                    *reinterpret_cast<uint4*>(sBuf + cpI - CP_UP) = c0;
                    *reinterpret_cast<uint4*>(sBuf + cpI + 2 * X::MPL * 16 - CP_UP) = c1;
                }
                if (edge && lane < 16) *reinterpret_cast<uint4*>(sBuf + cpE - CP_UP) = ce;
            };
            constexpr int NSE = FIN_N + PARK_N + 1;
            auto edge_side = [&](int t) __attribute__((always_inline)) {
                if (t < FIN_N) fin_micro(itp, k > 0, X::THW - 1, t);
                else if (t < FIN_N + PARK_N) park_micro(t - FIN_N);
                else copy_step(X::NRG == 1 || tg == 1, do_edge);
            };
            frag xf[PF + 1][2];
            auto read_x = [&](int u) __attribute__((always_inline)) {
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    if (u < NUE) {
                        const int kc = u % X::NKC, tap = u / X::NKC;
                        xf[u % (PF + 1)][pc] = *reinterpret_cast<const frag*>(sBuf + xe0 + ((8 * kc + pc) * X::XPL + (tap / 3) * X::IC + tap % 3) * 16);
                    } else {
                        const int v = u - NUE, Rl = v / (3 * X::NKC), dx = (v / X::NKC) % 3, kc = v % X::NKC;
                        xf[u % (PF + 1)][pc] = *reinterpret_cast<const frag*>(sBuf + xa + ((8 * kc + pc) * X::XPL + Rl * X::IC + dx) * 16);
                    }
                }
            };
            // ---- (a) the edge block.  Under it: the previous tile's last output row, this tile's residual parking, the copies
            f32x4c accP[NAP];
#pragma unroll
            for (int pr = 0; pr < NAP; ++pr)
#pragma unroll
                for (int i = 0; i < 4; ++i) accP[pr][i] = 0.f;
            {
                constexpr int GE = NUE * 3;
#pragma unroll
                for (int u = 0; u < PF; ++u) read_x(u);
#pragma unroll
                for (int u = 0; u < NUE; ++u) {
                    const int kc = u % X::NKC, tap = u / X::NKC;
                    read_x(u + PF);
                    const frag (&x)[2] = xf[u % (PF + 1)];
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr) {
                        accP[pr % NAP] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[tap][kc][pr == 0 ? 1 : 0], x[pr == 1 ? 1 : 0], accP[pr % NAP], 0, 0, 0);
                        int lo, hi;
                        share(u * 3 + pr, GE, NSE, lo, hi);
#pragma unroll
                        for (int t = lo; t < hi; ++t) edge_side(t);
                        SIDE_PIN();
                    }
                }
            }
            // ---- (b) the row blocks, input row after input row.  Under input row 0: the edge block's hand-over; under row Rl >= 3: m row Rl - 3
            const bool in_e = (unsigned)(it.ty * X::TH - 1 + X::R0 + e_row) < (unsigned)p.Ho && (unsigned)ix_e < (unsigned)p.Wo;
#pragma unroll
            for (int Rl = 0; Rl < X::IRW; ++Rl) {
                const int dy_lo = Rl - (X::MRW - 1) > 0 ? Rl - (X::MRW - 1) : 0, dy_hi = Rl < 2 ? Rl : 2;     // m rows Rl - dy in [0, MRW)
                const int nv = dy_hi - dy_lo + 1;
                const int G = 3 * X::NKC * nv * 3;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int kc = 0; kc < X::NKC; ++kc) {
                        const int u = NUE + (Rl * 3 + dx) * X::NKC + kc;
                        if (u + PF < NU) read_x(u + PF);
                        const frag (&x)[2] = xf[u % (PF + 1)];
#pragma unroll
                        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                            for (int dy = dy_lo; dy <= dy_hi; ++dy) {
                                const int tap = dy * 3 + dx;
                                acc1[Rl - dy] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[tap][kc][pr == 0 ? 1 : 0], x[pr == 1 ? 1 : 0], acc1[Rl - dy], 0, 0, 0);
                                const int g = (((dx * X::NKC + kc) * 3) + pr) * nv + (dy - dy_lo);
                                int lo, hi;
                                share(g, G, HAND_N, lo, hi);
#pragma unroll
                                for (int t = lo; t < hi; ++t) {
                                    if (Rl == 0) {
                                        if (t == 0) {
#pragma unroll
                                            for (int pr = 1; pr < NAP; ++pr)
#pragma unroll
                                                for (int i = 0; i < 4; ++i) accP[0][i] += accP[pr][i];
                                        }
                                        hand_micro(accP[0], in_e, do_edge, hse0, t);
                                    } else if (Rl >= 3) {
                                        const int rl = Rl - 3;
                                        hand_micro(acc1[rl], rl == 0 ? iy_m0 >= 0 : true, true, hs + rl * X::MC * 16, t);
                                    }
                                }
                                SIDE_PIN();
                            }
                    }
            }
            ROMP_TRACE(11);
#pragma unroll
            for (int t = 0; t < HAND_N; ++t) hand_micro(acc1[X::MRW - 1], (unsigned)(iy_m0 + X::MRW - 1) < (unsigned)p.Ho, true, hs + (X::MRW - 1) * X::MC * 16, t);
            ROMP_TRACE(13);
        }
        // ---- 2. every wave is done with the input halo and m is complete
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ROMP_TRACE(12);

        // ================= 3. conv2 from m, m row after m row (local rows again).  Under m row Rl: a share of the next tile's halo DMA
        // and the finish of output row Rl - 3 (complete since m row Rl - 1); the wave's last output row waits for the next tile
#pragma unroll
        for (int r = 0; r < X::THW; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc2[r][i] = 0.f;
        constexpr int PF2 = 2, NU2 = X::MRW2 * 3 * X::NKC;       // unit (Rl * 3 + dx) * NKC + kc
        frag xg[PF2 + 1][2];
        auto read_m = [&](int u) __attribute__((always_inline)) {
            const int Rl = u / (3 * X::NKC), dx = (u / X::NKC) % 3, kc = u % X::NKC;
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
                xg[u % (PF2 + 1)][pc] = *reinterpret_cast<const frag*>(sBuf + ma + ((8 * kc + pc) * X::MPL + Rl * X::MC + dx) * 16);
        };
#pragma unroll
        for (int u = 0; u < PF2; ++u) read_m(u);
#pragma unroll
        for (int Rl = 0; Rl < X::MRW2; ++Rl) {
            const int dy_lo = Rl - (X::THW - 1) > 0 ? Rl - (X::THW - 1) : 0, dy_hi = Rl < 2 ? Rl : 2;        // output rows Rl - dy in [0, THW)
            const int nv = dy_hi - dy_lo + 1;
            const int G = 3 * X::NKC * nv * 3;
            // this row's share of the DMA pieces: all of them under the first half of the rows (the last ones need time to land
            // before the barrier at the end of the tile)
            constexpr int FR = (X::MRW2 + 1) / 2;
            const int f_lo = Rl < FR ? Rl * X::NI / FR : X::NI, f_hi = Rl < FR ? (Rl + 1) * X::NI / FR : X::NI;
            const int NS = (Rl >= 3 ? FIN_N : 0) + (f_hi - f_lo);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int kc = 0; kc < X::NKC; ++kc) {
                    const int u = (Rl * 3 + dx) * X::NKC + kc;
                    if (u + PF2 < NU2) read_m(u + PF2);
                    const frag (&x)[2] = xg[u % (PF2 + 1)];
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int dy = dy_lo; dy <= dy_hi; ++dy) {
                            const int tap = dy * 3 + dx;
                            if (!(DBG & 8))
                                acc2[Rl - dy] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[tap][kc][pr == 0 ? 1 : 0], x[pr == 1 ? 1 : 0], acc2[Rl - dy], 0, 0, 0);
                            const int g = (((dx * X::NKC + kc) * 3) + pr) * nv + (dy - dy_lo);
                            int lo, hi;
                            share(g, G, NS, lo, hi);
#pragma unroll
                            for (int t = lo; t < hi; ++t) {
                                if (t < f_hi - f_lo) fetch_piece(itn, has_next, next_interior, next_first, f_lo + t);
                                else fin_micro(it, true, Rl - 3, t - (f_hi - f_lo));
                            }
                            SIDE_PIN();
                        }
                }
        }
        ROMP_TRACE(17);
        if (DBG & 4) {
#pragma unroll
            for (int r = 0; r < X::THW; ++r) asm volatile("" :: "v"(acc2[r]));
        }
        // ---- 4. the next halo has landed, for every wave; m may be overwritten.  A COUNTED wait (round 5): the halo pieces go out under
        // the first FR m rows, each output row's store at the end of its row from row 3 on, so exactly MRW2 - max(3, FR - 1) stores are
        // younger than the last piece; a wave's memory operations retire in issue order: "at most that many outstanding" = every piece
        // has landed while the tile's last stores stay in flight across the barrier (vmcnt(0) waited one store round trip per tile
        // with nothing to compute).  Knock-out builds without the finish have no stores to count on: full drain
        constexpr int FRW = (X::MRW2 + 1) / 2;                  // (= FR of the conv2 loop above)
        constexpr int STORES_AFTER_DMA = X::MRW2 - (FRW - 1 > 3 ? FRW - 1 : 3);
        static_assert(STORES_AFTER_DMA == (C == 64 ? 6 : 3), "the counted wait below");
#ifdef ROMP_BBLOCK_DRAIN0                                      // (A/B builds: the full drain of rounds 3-4; python -m romp_amd.build with extra_flags)
        if (true) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
        if (DBG & 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
        else if (C == 64) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ROMP_TRACE(15);
        itp = it;
        it = itn;
        run = run_n; kr = kr_n;
    }
    if (!(DBG & 4)) {                                          // the last tile's last output row
        const int r = X::THW - 1;
        const uint2 rh = *reinterpret_cast<const uint2*>(sR + (r * 2 + 0) * 512), rl = *reinterpret_cast<const uint2*>(sR + (r * 2 + 1) * 512);
        float ev[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned wh = e < 2 ? rh.x : rh.y, wl = e < 2 ? rl.x : rl.y;
            const float v = fmaf(acc2[r][e], s2[e], b2[e]);
            ev[e] = (e & 1) ? add_pieces_relu<1>(v, wh, wl, H2_MAX) : add_pieces_relu<0>(v, wh, wl, H2_MAX);
        }
        unsigned eh[2] = {pack_hi(ev[0], ev[1]), pack_hi(ev[2], ev[3])};
        sat_pk = sat_track_pk(sat_pk, eh[0], eh[1]);
        unsigned el[2] = {h2_low_pair(eh[0], ev[0], ev[1]), h2_low_pair(eh[1], ev[2], ev[3])};
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t a = __builtin_amdgcn_permlane16_swap(eh[0], el[0], false, false);
        const u32x2_t b = __builtin_amdgcn_permlane16_swap(eh[1], el[1], false, false);
        const int oy = itp.ty * X::TH + tg * X::THW + r, ox = itp.tx * X::TW + px;
        float* o = p.out + (size_t)itp.b * p.out_bs + p.out_co + (unsigned)((oy * p.out_rs + ox * p.out_cs) + (16 * cg + 4 * q));
        *reinterpret_cast<uint4*>(o) = make_uint4(a[0], b[0], a[1], b[1]);
    }
    sat_report_pk(p.sat, sat_pk);
#undef SIDE_PIN
}

// `op` is the block's SECOND conv (its residual is the block input x, its output y); `op1` the first.  Their per-wave weight packs
// (plan.pack_h2_wave16) are in `weight_aux`.
template <int C>
static int launch_bblockr(const romp_op& op1, const romp_op& op, const float* x, float* y, int B, int* queue, hipStream_t st) {
    using X = RCfg<C>;
    ROMP_REQUIRE(op.ksize == 3 && op.stride == 1 && op.Cin == C && op.Cout == C && op.groups == 1 &&
                 op1.ksize == 3 && op1.stride == 1 && op1.Cin == C && op1.Cout == C && op1.groups == 1,
                 "bblock%d: two 3x3 stride-1 %d -> %d convs expected", C, C, C);
    ROMP_REQUIRE(op1.weight_aux && op1.scale_h2 && op.weight_aux && op.scale_h2 && (op1.flags & op.flags & ROMP_OPF_WAVE16) && op1.relu && op.relu,
                 "bblock%d: per-wave f16x2 weight packs (ROMP_OPF_WAVE16) and ReLUs expected", C);
    ROMP_REQUIRE(op1.in_fmt == ROMP_FMT_H2 && op.res_fmt == ROMP_FMT_H2 && op.out_fmt == ROMP_FMT_H2 && op1.act_shift == op.act_shift,
                 "bblock%d: H2 tensors expected", C);
    ROMP_REQUIRE(op.H % X::TH == 0 && op.W % X::TW == 0 && op1.H == op.H && op1.W == op.W, "bblock%d: %dx%d is not a multiple of the 8x16 tile", C, op.H, op.W);
    ROMP_REQUIRE(op1.in_cstride == op.res_cstride && op1.in_coff == op.res_coff && ((op1.in_cstride | op1.in_coff | op.out_cstride | op.out_coff) & 7) == 0,
                 "bblock%d: the residual must be the block input, octet aligned", C);
    static bool attr = false;
    static int num_cu = 256;
    using KernelFn = void (*)(ConvParams);
    static KernelFn fn = bblockr_kernel<C, 0>;
    static KernelFn fn_strip = bblockr_kernel<C, 0, true>;
    if (!attr) {                                               // (romp_net_create calls this path's set-up outside any stream capture)
#ifdef ROMP_BBLOCK_KNOCKOUTS
        const char* e = getenv("ROMP_CONV_DEBUG");
        switch ((e ? atoi(e) : 0) & 15) {                      // (the other bits belong to other kernels)
            case 0: break;
            case 1: fn = bblockr_kernel<C, 1>; break;
            case 2: fn = bblockr_kernel<C, 2>; break;
            case 4: fn = bblockr_kernel<C, 4>; break;
            case 7: fn = bblockr_kernel<C, 7>; break;
            case 8: fn = bblockr_kernel<C, 8>; break;
            default: ROMP_REQUIRE(false, "bblock%d: ROMP_CONV_DEBUG & 15 is one of 0 1 2 4 7 8 here", C);
        }
#endif
        ROMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, X::LDS_BYTES));
        ROMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn_strip), hipFuncAttributeMaxDynamicSharedMemorySize, X::LDS_BYTES));
        int dev = 0;
        hipDeviceProp_t prop;
        ROMP_HIP_CHECK(hipGetDevice(&dev));
        ROMP_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        attr = true;
    }
    if (x == nullptr && y == nullptr) return ROMP_OK;          // set-up only
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.in = x; p.res = x; p.out = y;
    p.w3 = reinterpret_cast<const uint4*>(op1.weight_aux);
    p.wh = reinterpret_cast<const uint4*>(op.weight_aux);
    p.scale = op1.scale_h2; p.w = op1.shift;
    p.scale_h = op.scale_h2; p.shift = op.shift;
    p.act_scale = ldexpf(1.f, op.act_shift);
    p.inv_act_scale = ldexpf(1.f, -op.act_shift);
    p.in_h2 = p.out_h2 = p.res_h2 = 1;
    p.queue = queue;
    p.trace = conv_trace_arm(st);
    p.sat = conv_sat_counter();
    {
        const unsigned long long bytes = ((unsigned long long)B * op.H * op.W * op1.in_cstride - op1.in_coff) * 4ull;
        ROMP_REQUIRE(bytes < 0x80000000ull, "bblock%d: input tensor of %llu bytes: beyond the 31-bit offsets of the halo fetch", C, bytes);
        p.in_bytes = (unsigned)bytes;
    }
    p.H = p.Ho = op.H; p.W = p.Wo = op.W;
    p.Cout = C; p.cin_valid = C; p.cin_pad = C; p.cout_pad = C;
    p.in_cs = op1.in_cstride; p.in_co = op1.in_coff;
    p.out_cs = op.out_cstride; p.out_co = op.out_coff;
    p.res_cs = op.res_cstride; p.res_co = op.res_coff;
    p.relu = 1;
    p.tiles_x = op.W / X::TW; p.tiles_y = op.H / X::TH; p.tiles_total = B * p.tiles_x * p.tiles_y;
    p.nslices = p.ns_total = 1;
    p.n_queues = (p.tiles_total % 8 == 0) ? 8 : 1;
    p.per_queue = p.tiles_total / p.n_queues;
    p.vec_io = 1;
    p.pad_h = p.pad_w = 1;
    p.out_rs = op.out_rstride > 0 ? op.out_rstride : p.Wo * op.out_cstride;
    p.out_bs = op.out_bstride > 0 ? op.out_bstride : p.Ho * p.Wo * op.out_cstride;
    const int cap = conv_wg_cap();
    const long grid_max = (long)num_cu * ((cap > 0 && cap < X::WG_PER_CU) ? cap : X::WG_PER_CU);
    // The strip form (RCfg): runs of L vertically consecutive tiles, L a divisor of the tile rows.  In units of one plain tile a
    // carrying tile costs ~0.85 and a run's first ~1.05 (the prologue is not hand-scheduled); a workgroup takes ceil(runs / grid) runs
    // against ceil(tiles / grid) plain tiles: the cheapest estimate wins, the plain kernel if none beats it (small batches: L = 1)
    int L = 0;
    {
        auto rounds = [&](long items) { const long g = items < grid_max ? items : grid_max; return (items + g - 1) / g; };
        double best = (double)rounds(p.tiles_total);
        for (int l = 2; l <= p.tiles_y; ++l) {
            if (p.tiles_y % l) continue;
            const double c = (double)rounds(p.tiles_total / l) * (1.05 + 0.85 * (l - 1));
            if (c < best - 1e-9) { best = c; L = l; }
        }
        // env ROMP_BBLOCK_RUN (tests, A/B runs; read at every launch -- launches are captured into the net's graph once): 0 = never the
        // strip form, n > 1 = runs of n tiles wherever n divides the tile rows (the plain kernel elsewhere)
        const char* e = getenv(C == 64 ? "ROMP_BBLOCK_RUN64" : "ROMP_BBLOCK_RUN32");       // (one channel count only: A/B runs)
        if (!e) e = getenv("ROMP_BBLOCK_RUN");
        const int run_env = e ? atoi(e) : -1;
        if (run_env == 0) L = 0;
        else if (run_env > 0) L = (run_env > 1 && p.tiles_y % run_env == 0) ? run_env : 0;
    }
    KernelFn f = fn;
    long items = p.tiles_total;
    if (L > 1 && fn == bblockr_kernel<C, 0>) {                 // (knock-out builds time the plain kernel)
        f = fn_strip;
        p.run_len = L;
        items = p.tiles_total / L;
        p.n_queues = (items % 8 == 0) ? 8 : 1;
        p.per_queue = (int)(items / p.n_queues);
    }
    long grid = grid_max;
    if (grid > items) grid = items;
    if (p.n_queues == 8) grid = grid >= 8 ? (grid / 8) * 8 : 8;
    hipLaunchKernelGGL(f, dim3((unsigned)grid), dim3(256), X::LDS_BYTES, st, p);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

}  // namespace romp
