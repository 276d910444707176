// conv_h2b.hip -- a whole 32-channel BasicBlock (simple_romp/romp/model.py:54-83) in ONE kernel on the f16x2 split:
//     y = relu(bn2(conv3x3(relu(bn1(conv3x3(x))))) + x),   32 -> 32 -> 32 channels, stride 1, H2 tensors in and out.
// Why (round 3, profiles/r03_h2r_notes.md): the 32-channel @128^2 class is the network's largest (64 launches per image batch,
// ~3.1 ms of a 14.9 ms forward) and its time does not respond to anything done INSIDE a conv kernel; as two launches a block
// moves 5 tensors over the fabric (x -> m; m, x -> y: 335 MB at B = 32) and pays launch / prologue / epilogue twice.  Fused, x is
// read once (haloed), y written once, and the intermediate m never leaves the CU.
//
// One 256-thread workgroup per CU -- ONE wave per SIMD, so each wave owns the SIMD's whole 512-entry register file -- persistent
// over 16x16-pixel output tiles.  Every wave keeps the split weights of BOTH convs in registers for the whole launch (2 x 144:
// conv1's pinned to AGPRs, which MFMA reads directly; no weight traffic after the prologue).  Per tile:
//   1. conv1 on the tile's 18x18 halo of m (11 blocks of 32 pixels: nine 2-row x 16-column blocks and two blocks holding the
//      edge columns; 3 block slots per wave) from the 20x20 input halo of x, which arrived by raw-buffer LDS-DMA (both 16-channel
//      chunks) under the previous tile's conv2.  Under slot 0's MFMAs: the finish of the PREVIOUS tile's second output block and
//      the parking of this tile's residual pieces (LDS halo -> per-lane LDS slots); under slot s: the hand-over of slot s - 1
//      (bn1 + ReLU, ZERO outside the image = conv2's padding, pre-split, into LDS in the rotated unit layout the fragment reads
//      use).  The last slot's hand-over is the one tail; barrier;
//   2. conv2 from m in LDS (two blocks per wave).  Under block 0: the DMA of the NEXT tile's halo; under block 1: the finish of
//      block 0 (bn2 + x + ReLU in the scaled domain, split, v_permlane32_swap, 16-byte stores); drain; barrier.
// With one wave per SIMD nothing hides a wave's own stalls and the wave issues in order: every kind of side work is cut into
// micro-steps of a few instructions, one behind each MFMA, pinned there (sched_barrier); table reads are issued an LDS round trip
// ahead of their use.  History and measurements: profiles/r03_fused_blocks_notes.md.  Batch plans run the 32-channel blocks on
// conv_h2c.h's row-pipelined kernel instead (launch_bblock32 hands over when the ops carry per-wave weight packs); this kernel
// serves single-image plans.
#include "conv_split.h"
#include "conv_fuse.h"
#include <string.h>

namespace romp {

struct BCfg {
    static constexpr int TH = 16, TW = 16;
    static constexpr int IR = TH + 4, IC = TW + 4;             // input halo 20 x 20
    static constexpr int MR = TH + 2, MC = TW + 2;             // intermediate halo 18 x 18
    static constexpr int RSU = 80;                             // 16-byte units per (input or m) row: 20 pixel slots x 4 units
    static constexpr int NI = 7;                               // DMA pieces per wave and 16-channel stage: 4 x 7 x 64 units >= 20 x 80
    static constexpr int STAGE_BYTES = 4 * NI * 1024;          // 28 KiB (1600 units used), one per input chunk
    static constexpr int MID_PLANE = MR * RSU * 16;            // one 16-channel chunk of m: 23 040 bytes
    static constexpr int OFF_M = 2 * STAGE_BYTES;
    static constexpr int OFF_R = OFF_M + 2 * MID_PLANE;        // the tile's residual pieces, parked per lane: [wave][block][group][piece][lane] x 8 bytes
    static constexpr int OFF_S = OFF_R + 4 * 2 * 4 * 2 * 64 * 8;  // [scale1 32 | shift1 32 | scale2 32 | shift2 32]
    static constexpr int LDS_BYTES = OFF_S + 128 * 4 + 16;
    static_assert(IR * RSU <= 4 * NI * 64, "stage pieces cover the input halo");
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
};

typedef __attribute__((address_space(3))) void lds_void_b;
typedef const __attribute__((address_space(1))) void glb_void_b;

// unit index of (column c, unit w of the 16-channel chunk) inside a row of RSU units (the rotated layout of conv_h2r.hip)
__device__ __forceinline__ int unit_of(int c, int w) { return (c >> 2) * 16 + (c & 3) + 4 * ((w + (c >> 2)) & 3); }

// ConvParams as used here: in = x (H2), res = x, out = y (H2); w3 = conv1's split weights, wh = conv2's; scale = conv1's
// f16x2 epilogue scale (32), w = conv1's shift (32, as floats), scale_h / shift = conv2's; the geometry fields as for a conv.
// DBG: timing knock-outs (env ROMP_CONV_DEBUG, wrong outputs): 1 no halo DMA, 2 no hand-over / residual parking, 4 no finish, 8 no MFMA,
// 64 no fragment reads
template <int DBG>
__global__ __launch_bounds__(256, 1) void bblock32_kernel(ConvParams p) {
    using X = BCfg;
    using frag = f16x8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sBuf = reinterpret_cast<char*>(smem);
    char* sM = sBuf + X::OFF_M;
    float* sS = reinterpret_cast<float*>(sBuf + X::OFF_S);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void_b*)sBuf;      // LDS byte address of the dynamic segment

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    int tr_n = 0;                                              // phase stamps (ROMP_CONV_TRACE=1, scripts/bblock_bench.py): 1 entry, 4 set-up done,
    constexpr int tr_wpw = 4;                                  // per tile 11 conv1 units, 13 hand-over tail, 12 barrier, 17 conv2 units, 15 barrier
    ROMP_TRACE(1);
    const int q = p.n_queues == 8 ? (blockIdx.x & 7) : 0;
    const int nwg_q = gridDim.x / p.n_queues;
    const int j0 = blockIdx.x / p.n_queues;
    if (j0 >= p.per_queue) return;
    const int n_mine = (p.per_queue - j0 + nwg_q - 1) / nwg_q;  // tiles of this workgroup: j0, j0 + nwg_q, ...

    // scale / shift of both convs, PRE-MULTIPLIED by 2^act_shift: m and y are produced directly in the scaled domain the H2 pieces
    // live in (ReLU commutes with the positive factor; the residual's pieces h1 + h2 are x * 2^act_shift already)
    if (tid < 128) {
        const float* src = tid < 32 ? p.scale : tid < 64 ? p.w : tid < 96 ? p.scale_h : p.shift;
        sS[tid] = src[tid & 31] * p.act_scale;
    }
    auto tile_of = [&](int k) __attribute__((always_inline)) { return decode_item(p, q, j0 + k * nwg_q, 32); };

    // DMA pieces of this wave for one 16-channel input stage (25 pieces of 64 units cover the 20 x 80 units exactly: piece 4 k + wv
    // is this wave's k-th): the 16-byte unit a lane fetches -- its (row, col) in the 20x20 halo and its byte offset from the halo origin
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    int d_rc[X::NI], d_off[X::NI];                             // row | col << 8;  ((row * W + col) * in_cs + 4 unit) * 4
#pragma unroll
    for (int k = 0; k < X::NI; ++k) {
        const int U = (k * 4 + wv) * 64 + lane;
        const int row = U / X::RSU, r = U % X::RSU;
        const int cg = r >> 4, r16 = r & 15;
        const int col = cg * 4 + (r16 & 3), w = ((r16 >> 2) - cg) & 3;
        d_rc[k] = row | (col << 8);
        d_off[k] = ((row * p.W + col) * p.in_cs + w * 4) * 4;
    }
    // the input tensor as a raw buffer: a lane whose pixel is outside the image asks for an offset beyond num_records and gets zeros
    i32x4_t rsrc;
    {
        const unsigned long long base = (unsigned long long)(p.in + p.in_co);
        rsrc[0] = (int)(unsigned)base;
        rsrc[1] = (int)(unsigned)(base >> 32) & 0xffff;
        rsrc[2] = (int)((unsigned)p.in_bytes);
        rsrc[3] = 0x00020000;
    }
    // piece kk of chunk ch of tile `it` -> stage buffer ch.  Inline asm, not the builtin: hipcc guards every LDS read that follows a DMA
    // it can see with a vmcnt wait (it cannot tell the buffers apart), which here would put the halo's whole memory latency in front
    // of conv2's first fragment read.  Ordered by hand: vmcnt(0) + barrier at the end of the tile.  An interior tile (no lane
    // outside the image) costs an add, the M0 write and the DMA; an edge tile adds the per-lane test.
    auto is_interior = [&](const Item& it) __attribute__((always_inline)) { return it.ty > 0 && it.ty < p.tiles_y - 1 && it.tx > 0 && it.tx < p.tiles_x - 1; };
    auto fetch_piece = [&](const Item& it, bool valid, bool interior, int ch, int kk) __attribute__((always_inline)) {
        if (DBG & 1) return;
        if (!valid || (kk == X::NI - 1 && wv != 0)) return;     // (uniform)
        const int iy0 = it.ty * X::TH - 2, ix0 = it.tx * X::TW - 2;
        const int origin = (((it.b * p.H + iy0) * p.W + ix0) * p.in_cs + ch * 16) * 4;    // may be "negative": the sum with d_off is not
        const unsigned dst = lds0 + (unsigned)(ch * X::STAGE_BYTES + (kk * 4 + wv) * 1024);
        int voff = d_off[kk] + origin;
        if (!interior) {
            int rc = d_rc[kk];
            asm volatile("" : "+v"(rc));                       // (opaque: keeps per-piece address parts out of the loop-invariant VGPRs)
            const int iy = iy0 + (rc & 255), ix = ix0 + (rc >> 8);
            const int ok = (int)((unsigned)iy < (unsigned)p.H) & (int)((unsigned)ix < (unsigned)p.W);
            voff = ok ? voff : (int)0x80000000;
        }
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsrc), "s"(dst) : "memory");
    };
    // conv1 block slots of this wave: blocks wv, wv + 4, wv + 8 of the 11 (slot 2 of wave 3 is idle).  Blocks 0-8: m rows 2b, 2b + 1,
    // m columns 1..16; block 9: m rows 0..15, columns {0, 17}; block 10: rows 16, 17, columns {0, 17} (4 lanes).
    // Slots 0 and 1 are always plain blocks 8 rows apart: slot 1's addresses are slot 0's plus a constant.
    int myA, mxA, myC, mxC;                                    // m pixel of this lane in slots 0 (1: + 8 rows) and 2
    bool actC;
    {
        myA = 2 * wv + li / 16; mxA = 1 + li % 16;
        const int b = wv + 8;
        if (b < 9) { myC = 2 * b + li / 16; mxC = 1 + li % 16; actC = true; }
        else if (b == 9) { myC = li >> 1; mxC = (li & 1) * 17; actC = true; }
        else if (b == 10) { myC = li < 4 ? 16 + (li >> 1) : 0; mxC = li < 4 ? (li & 1) * 17 : 0; actC = li < 4; }
        else { myC = 0; mxC = 0; actC = false; }
    }
    int xaA[3][2], xaC[3][2];                                  // conv1: fragment address of input pixel (my, mx + dx), unit 2 lh + pc; + dy rows
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            xaA[dx][pc] = (myA * X::RSU + unit_of(mxA + dx, lh * 2 + pc)) * 16;
            xaC[dx][pc] = (myC * X::RSU + unit_of(mxC + dx, lh * 2 + pc)) * 16;
        }
    int xa2[3][2];                                             // conv2: m pixel (4 wv + li / 16 + dy, li % 16 + dx); block j adds 2 rows
    int ra2[2][2];                                             // the residual x of output pixel (4 wv + li / 16, li % 16) in the input halo: [octet-in-chunk][piece]
    char* sR = sBuf + X::OFF_R + (wv * 1024 + lane) * 8;       // this lane's parking slots: + ((j * 4 + g4) * 2 + pc) * 512
    {
        const int prow = 4 * wv + li / 16, pcol = li % 16;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) xa2[dx][pc] = (prow * X::RSU + unit_of(pcol + dx, lh * 2 + pc)) * 16;
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) ra2[o][pc] = ((prow + 2) * X::RSU + unit_of(pcol + 2, o * 2 + pc)) * 16 + lh * 8;
    }

    Item it = tile_of(0);
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int kk = 0; kk < X::NI; ++kk) fetch_piece(it, true, false, ch, kk);
    // ---- the weights of both convs, all taps and both chunks, resident (asked for AFTER the first halo: both trips overlap): lane (li, lh) = channel li, k-half lh
    frag w1[2][9][2], w2[2][9][2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {                    // packed [tap][chunk 2][piece 2][k-half 2][32] units
                w1[ch][tap][pc] = __builtin_bit_cast(frag, p.w3[(((tap * 2 + ch) * 2 + pc) * 2 + lh) * 32 + li]);
                w2[ch][tap][pc] = __builtin_bit_cast(frag, p.wh[(((tap * 2 + ch) * 2 + pc) * 2 + lh) * 32 + li]);
            }
    // A "use" of every weight register in front of the tile loop: hipcc then waits for these loads HERE, once (left to the first
    // MFMA inside the loop its wait is a conservative vmcnt(0) on every iteration)
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            asm volatile("" : "+a"(w1[ch][tap][0]), "+a"(w1[ch][tap][1]), "+v"(w2[ch][tap][0]), "+v"(w2[ch][tap][1]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // tile 0's halo and the scale table are in; weights in registers
    ROMP_TRACE(4);

    // one H2 value pair from two scaled float values: high pieces / low pieces as packed fp16 (v_cvt_pk_f16_f32: round to nearest even)
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    auto split2 = [&](float a, float b, unsigned& hi, unsigned& lo) __attribute__((always_inline)) {
        const f32x2_t v = {h2_sat(a), h2_sat(b)};
        const f16x2_t h = __builtin_convertvector(v, f16x2_t);
        const f32x2_t r = {v[0] - (float)h[0], v[1] - (float)h[1]};
        const f16x2_t l = __builtin_convertvector(r, f16x2_t);
        hi = __builtin_bit_cast(unsigned, h);
        lo = __builtin_bit_cast(unsigned, l);
    };
    // the same in two micro-steps: (a, b) -> hi;  -> lo = fp16(a - hi.x), fp16(b - hi.y)
    auto split_a = [&](float a, float b, unsigned& hi) __attribute__((always_inline)) {
        const f32x2_t v = {a, b};
        hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
    };
    auto split_b = [&](float a, float b, unsigned hi, unsigned& lo) __attribute__((always_inline)) {
        lo = h2_low_pair(hi, a, b);
    };
    const float* sS1 = sS + lh * 4;                            // this lane's channels of a group: 8 g4 + 4 lh .. + 3

    // One wave per SIMD: nothing but this wave's OWN instructions can run in the shadow of its MFMAs, and the wave issues in
    // order -- side work placed after a unit's three MFMAs runs after them, not under them (phase trace, round 3: 4 400 of a
    // tile's 9 900 conv1 cycles).  So every kind of side work is cut into STEPS of <= ~8 VALU instructions (an MFMA is 32 cycles,
    // a VALU instruction 4) and one step follows each MFMA, pinned there with a scheduling barrier.
#define SIDE_PIN() __builtin_amdgcn_sched_barrier(0)
    f32x16 acc2[2];                                            // (block 1's outlives its tile: finished under the next tile's conv1)
    unsigned sat_pk = 0u;                                      // per-half maximum of the high pieces formed: 0x7BFF iff clamped (sat_track_pk)
    Item itp = it;
#pragma unroll 1
    for (int k = 0; k < n_mine; ++k) {
        const bool has_next = k + 1 < n_mine;
        const Item itn = has_next ? tile_of(k + 1) : it;
        const bool next_interior = is_interior(itn);
        f32x16 acc1[3];
#pragma unroll
        for (int sl = 0; sl < 3; ++sl)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[sl][r] = 0.f;
        // ---- the finish of a conv2 block: bn2 + x + ReLU in the scaled domain, split; the two lanes of an octet trade halves
        // (v_permlane32_swap) so that each stores one whole 16-byte unit.  Eight steps per 4-channel group.  Block 0 is finished
        // under block 1's MFMAs, block 1 under slot 0 of the NEXT tile's conv1 (`tl` is then the previous tile; `live`: there is one).
        f32x4 e_sc, e_sh;
        uint2 e_rh, e_rl;
        float ev[4];
        unsigned eh[2], el[2];
        int e_o = 0;
        constexpr int FIN_N = 4 + 4 * 15;
        auto fin_micro = [&](const Item& tl, bool live, int j, int t) __attribute__((always_inline)) {   // output pixel (4 wv + 2 j + li / 16, li % 16), channels 8 g4 + 4 lh .. + 3
            if (DBG & 4) return;
            auto prep = [&](int g4) __attribute__((always_inline)) {      // (an LDS round trip ahead of its first use: micro-steps 1-3 are empty)
                e_sc = *reinterpret_cast<const f32x4*>(sS1 + 64 + g4 * 8);
                e_sh = *reinterpret_cast<const f32x4*>(sS1 + 96 + g4 * 8);
                e_rh = *reinterpret_cast<const uint2*>(sR + ((j * 4 + g4) * 2 + 0) * 512);
                e_rl = *reinterpret_cast<const uint2*>(sR + ((j * 4 + g4) * 2 + 1) * 512);
            };
            if (t < 4) { if (t == 0) prep(0); return; }
            const int g4 = (t - 4) / 15, w = (t - 4) % 15;
            switch (w) {
            case 0: case 2: case 6: case 8: {                  // value e, first half: bn2
                const int e = w == 0 ? 0 : w == 2 ? 1 : w == 6 ? 2 : 3;
                ev[e] = fmaf(acc2[j][g4 * 4 + e], e_sc[e], e_sh[e]);
                break; }
            case 1: case 3: case 7: case 9: {                  // second half: + the residual's two pieces, ReLU, saturate
                const int e = w == 1 ? 0 : w == 3 ? 1 : w == 7 ? 2 : 3;
                const unsigned wh = e < 2 ? e_rh.x : e_rh.y, wl = e < 2 ? e_rl.x : e_rl.y;
                ev[e] = (e & 1) ? add_pieces_relu<1>(ev[e], wh, wl, H2_MAX) : add_pieces_relu<0>(ev[e], wh, wl, H2_MAX);
                if (w == 9 && g4 < 3) prep(g4 + 1);            // the tables are free: the next group's reads go out now
                break; }
            case 4: split_a(ev[0], ev[1], eh[0]); break;
            case 5: split_b(ev[0], ev[1], eh[0], el[0]); break;
            case 10:
                split_a(ev[2], ev[3], eh[1]);
                if (live) sat_pk = sat_track_pk(sat_pk, eh[0], eh[1]);      // (not live: the first tile's pass over a block that does not exist)
                break;
            case 11: split_b(ev[2], ev[3], eh[1], el[1]); break;
            case 12: {
                // lanes L (channels .. + 0..3) and L + 32 (.. + 4..7): after the swaps L holds the octet's 8 high pieces, L + 32 its 8 low
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                const u32x2_t s0 = __builtin_amdgcn_permlane32_swap(eh[0], el[0], false, false);
                const u32x2_t s1 = __builtin_amdgcn_permlane32_swap(eh[1], el[1], false, false);
                eh[0] = s0[0]; el[0] = s0[1]; eh[1] = s1[0]; el[1] = s1[1];
                break; }
            case 13: {
                const int oy = tl.ty * X::TH + 4 * wv + 2 * j + li / 16, ox = tl.tx * X::TW + li % 16;
                e_o = (oy * p.out_rs + ox * p.out_cs) + (g4 * 8 + lh * 4);
                break; }
            default: {
                float* o = p.out + (size_t)tl.b * p.out_bs + p.out_co + (unsigned)e_o;
                if (live) *reinterpret_cast<uint4*>(o) = make_uint4(eh[0], eh[1], el[0], el[1]);
                break; }
            }
        };
        // ---- side work of conv1 ------------------------------------------------------------------------------------------------
        // (a) under slot 0, after the previous tile's finish: the residual pieces of this wave's output pixels go from the input
        //     halo to a per-lane parking area (the halo buffers are refilled under conv2): copy i = (block, group, piece) is read
        //     at step 32 + i and written two steps later
        uint2 pk[4];
        constexpr int PARK_N = 32;
        auto park_micro = [&](int t) __attribute__((always_inline)) {       // r0 r1 r2 | w0 r3 | w1 r4 | ... | w12 r15 | w13 w14 w15
            if (DBG & 2) return;
            auto rd = [&](int i) __attribute__((always_inline)) {
                const int j = i / 8, g4 = (i % 8) / 2, pc = i % 2;
                pk[i % 4] = *reinterpret_cast<const uint2*>(sBuf + (g4 >> 1) * X::STAGE_BYTES + ra2[g4 & 1][pc] + 2 * j * (X::RSU * 16));
            };
            auto wr = [&](int i) __attribute__((always_inline)) { *reinterpret_cast<uint2*>(sR + i * 512) = pk[i % 4]; };
            if (t < 3) rd(t);
            else if (t < 29) { if ((t - 3) % 2 == 0) wr((t - 3) / 2); else rd((t - 3) / 2 + 3); }
            else wr(13 + (t - 29));
        };
        // (b) under slot s: the hand-over of slot s - 1 -- bn1 + ReLU (in the scaled domain), zero outside the image (conv2's
        //     padding), split, into LDS as m.  Micro-steps 0-3: the first group's table reads, is this lane's pixel inside the
        //     image, two empty; then ten per 4-channel group: values 0, 1; split (2); values 2, 3 (+ the next group's table reads);
        //     split (2); the two stores
        f32x4 h_sc, h_sh;
        float hv[4];
        unsigned hh[2], ll[2];
        bool h_in = false;
        constexpr int HAND_N = 4 + 4 * 10;
        auto hand_micro = [&](int sl, int t) __attribute__((always_inline)) {
            if (DBG & 2) return;
            const int my = sl == 2 ? myC : myA + 8 * sl, mx = sl == 2 ? mxC : mxA;
            const bool act = sl == 2 ? actC : true;
            auto prep = [&](int g4) __attribute__((always_inline)) {      // (an LDS round trip ahead of its first use)
                h_sc = *reinterpret_cast<const f32x4*>(sS1 + g4 * 8);
                h_sh = *reinterpret_cast<const f32x4*>(sS1 + 32 + g4 * 8);
            };
            if (t < 4) {
                if (t == 0) prep(0);
                if (t == 1) {
                    const int iy = it.ty * X::TH - 1 + my, ix = it.tx * X::TW - 1 + mx;
                    h_in = act && (unsigned)iy < (unsigned)p.Ho && (unsigned)ix < (unsigned)p.Wo;
                }
                return;
            }
            const int g4 = (t - 4) / 10, w = (t - 4) % 10;
            char* m = sM + (g4 >> 1) * X::MID_PLANE + lh * 8 + my * (X::RSU * 16);   // channels 8 g4 + 4 lh ..: chunk g4 >> 1, octet-in-chunk g4 & 1, half lh of the unit
            switch (w) {
            case 0: case 1: case 4: case 5: {
                const int e = w < 2 ? w : w - 2;
                const float a = h2_sat(fmaxf(fmaf(acc1[sl][g4 * 4 + e], h_sc[e], h_sh[e]), 0.f));
                hv[e] = h_in ? a : 0.f;
                if (w == 5 && g4 < 3) prep(g4 + 1);            // the tables are free: the next group's reads go out now
                break; }
            case 2: split_a(hv[0], hv[1], hh[0]); break;
            case 3: split_b(hv[0], hv[1], hh[0], ll[0]); break;
            case 6:
                split_a(hv[2], hv[3], hh[1]);
                sat_pk = sat_track_pk(sat_pk, hh[0], hh[1]);       // (pixels outside the image are zeros)
                break;
            case 7: split_b(hv[2], hv[3], hh[1], ll[1]); break;
            case 8: if (act) *reinterpret_cast<uint2*>(m + unit_of(mx, (g4 & 1) * 2 + 0) * 16) = make_uint2(hh[0], hh[1]); break;
            default: if (act) *reinterpret_cast<uint2*>(m + unit_of(mx, (g4 & 1) * 2 + 1) * 16) = make_uint2(ll[0], ll[1]); break;
            }
        };
        // Micro-steps [lo, hi) of a phase's N for step m of a block's 54: spread evenly, the step in front of a unit's fragment
        // reads (every third) taking half a share
        auto share = [](int m, int N, int& lo, int& hi) __attribute__((always_inline)) {
            const int c0 = (m / 3) * 5 + (m % 3) * 2, c1 = ((m + 1) / 3) * 5 + ((m + 1) % 3) * 2;
            lo = c0 * N / 90; hi = c1 * N / 90;
        };
        // ---- 1. conv1, slot after slot
        {
            constexpr int PF = 2, NU = 54;                     // unit u: slot u / 18, chunk (u % 18) / 9, tap u % 9
            frag xf[PF + 1][2];
            auto read_x = [&](int u) __attribute__((always_inline)) {
                const int sl = u / 18, ch = (u % 18) / 9, tap = u % 9;
#pragma unroll
                for (int pc = 0; pc < 2; ++pc)
                    xf[u % (PF + 1)][pc] = *reinterpret_cast<const frag*>(sBuf + ch * X::STAGE_BYTES + (sl == 2 ? xaC : xaA)[tap % 3][pc] +
                                                                          ((sl == 1 ? 8 : 0) + tap / 3) * (X::RSU * 16));
            };
            auto side = [&](int sl, int m) __attribute__((always_inline)) {
                int lo, hi;
                if (sl == 0) {                                 // the previous tile's second block
                    share(m, FIN_N, lo, hi);
#pragma unroll
                    for (int t = lo; t < hi; ++t) fin_micro(itp, k > 0, 1, t);
                } else {                                       // slot 1 also parks this tile's residual
                    const int N = HAND_N + (sl == 1 ? PARK_N : 0);
                    share(m, N, lo, hi);
#pragma unroll
                    for (int t = lo; t < hi; ++t) { if (t < HAND_N) hand_micro(sl - 1, t); else park_micro(t - HAND_N); }
                }
                SIDE_PIN();
            };
#pragma unroll
            for (int u = 0; u < PF; ++u) read_x(u);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int sl = u / 18, ch = (u % 18) / 9, tap = u % 9, v = u % 18;
                if (u + PF < NU && !((DBG & 64) && u > 2)) read_x(u + PF);
                const frag (&x)[2] = xf[u % (PF + 1)];
                f32x16& aa = acc1[sl];
                f32x16& ab = acc1[sl];
                if (!(DBG & 8)) aa = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[ch][tap][1], x[0], aa, 0, 0, 0);
                side(sl, v * 3 + 0);
                if (!(DBG & 8)) ab = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[ch][tap][0], x[1], ab, 0, 0, 0);
                side(sl, v * 3 + 1);
                if (!(DBG & 8)) aa = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[ch][tap][0], x[0], aa, 0, 0, 0);
                side(sl, v * 3 + 2);
            }
            ROMP_TRACE(11);
            if (DBG & 2) asm volatile("" :: "v"(acc1[0]), "v"(acc1[1]), "v"(acc1[2]));   // (knock-out builds: keep the MFMAs)
#pragma unroll
            for (int t = 0; t < HAND_N; ++t) hand_micro(2, t); // the last slot's: nothing left to hide it under
            ROMP_TRACE(13);
        }
        // ---- 2. every wave is done with the input halo and m is complete
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ROMP_TRACE(12);
        // ---- 3. conv2 from m, block after block.  Under block 0 the NEXT tile's halo is fetched, a DMA piece every fourth step.  Under block 1 block 0 is finished: bn2 + x + ReLU in the scaled domain, split; the two lanes of an
        // octet trade halves (v_permlane32_swap) so that each stores one whole 16-byte unit.  Eight steps per 4-channel group.
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
        auto fetch_step = [&](int m) __attribute__((always_inline)) {      // piece i at step 4 i + 1 of block 0
            if (m % 4 == 1 && m / 4 < 2 * X::NI) fetch_piece(itn, has_next, next_interior, (m / 4) / X::NI, (m / 4) % X::NI);
        };
        {
            constexpr int PF = 2, NU = 36;                     // unit u: block u / 18, chunk (u % 18) / 9, tap u % 9
            frag xf[PF + 1][2];
            auto read_x = [&](int u) __attribute__((always_inline)) {
                const int j = u / 18, ch = (u % 18) / 9, tap = u % 9;
#pragma unroll
                for (int pc = 0; pc < 2; ++pc)
                    xf[u % (PF + 1)][pc] = *reinterpret_cast<const frag*>(sM + ch * X::MID_PLANE + xa2[tap % 3][pc] + (2 * j + tap / 3) * (X::RSU * 16));
            };
            auto side = [&](int j, int m) __attribute__((always_inline)) {
                if (j == 0) fetch_step(m);
                else {
                    int lo, hi;
                    share(m, FIN_N, lo, hi);
#pragma unroll
                    for (int t = lo; t < hi; ++t) fin_micro(it, true, 0, t);
                }
                SIDE_PIN();
            };
#pragma unroll
            for (int u = 0; u < PF; ++u) read_x(u);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int j = u / 18, ch = (u % 18) / 9, tap = u % 9, v = u % 18;
                if (u + PF < NU && !((DBG & 64) && u > 2)) read_x(u + PF);
                const frag (&x)[2] = xf[u % (PF + 1)];
                f32x16& aa = acc2[j];
                f32x16& ab = acc2[j];
                if (!(DBG & 8)) aa = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[ch][tap][1], x[0], aa, 0, 0, 0);
                side(j, v * 3 + 0);
                if (!(DBG & 8)) ab = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[ch][tap][0], x[1], ab, 0, 0, 0);
                side(j, v * 3 + 1);
                if (!(DBG & 8)) aa = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[ch][tap][0], x[0], aa, 0, 0, 0);
                side(j, v * 3 + 2);
            }
            ROMP_TRACE(17);
            if (DBG & 4) asm volatile("" :: "v"(acc2[0]), "v"(acc2[1]));
        }
        // ---- 4. the next halo has landed (and this tile's stores are out), for every wave; m may be overwritten
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ROMP_TRACE(15);
        itp = it;
        it = itn;
    }
    if (!(DBG & 4)) {                                          // the last tile's second block
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 e_sc = *reinterpret_cast<const f32x4*>(sS1 + 64 + g4 * 8);
            const f32x4 e_sh = *reinterpret_cast<const f32x4*>(sS1 + 96 + g4 * 8);
            const f16x4 rh = __builtin_bit_cast(f16x4, *reinterpret_cast<const uint2*>(sR + ((4 + g4) * 2 + 0) * 512));
            const f16x4 rl = __builtin_bit_cast(f16x4, *reinterpret_cast<const uint2*>(sR + ((4 + g4) * 2 + 1) * 512));
            float ev[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) ev[e] = fmaxf(fmaf(acc2[1][g4 * 4 + e], e_sc[e], e_sh[e]) + ((float)rh[e] + (float)rl[e]), 0.f);
            unsigned eh[2], el[2];
            split2(ev[0], ev[1], eh[0], el[0]);
            split2(ev[2], ev[3], eh[1], el[1]);
            sat_pk = sat_track_pk(sat_pk, eh[0], eh[1]);
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
            const u32x2_t s0 = __builtin_amdgcn_permlane32_swap(eh[0], el[0], false, false);
            const u32x2_t s1 = __builtin_amdgcn_permlane32_swap(eh[1], el[1], false, false);
            const int oy = itp.ty * X::TH + 4 * wv + 2 + li / 16, ox = itp.tx * X::TW + li % 16;
            float* o = p.out + (size_t)itp.b * p.out_bs + p.out_co + ((unsigned)(oy * p.out_rs + ox * p.out_cs) + (unsigned)(g4 * 8 + lh * 4));
            *reinterpret_cast<uint4*>(o) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
    }
    sat_report_pk(p.sat, sat_pk);
#undef SIDE_PIN
}

// `op` is the block's SECOND conv (its residual is the block input x, its output y); `op1` the first (weights / scale / shift).
int launch_bblock32(const romp_op& op1, const romp_op& op, const float* x, float* y, int B, int* queue, hipStream_t st) {
    {   // two implementations: this file's (v1: 16x16 tiles, one wave per SIMD) and conv_h2c.h's row-pipelined one (two waves per SIMD)
        // The plan decides (plan.fuse_basic_blocks): the row-pipelined
        // kernel needs the per-wave weight packs, announced by ROMP_OPF_WAVE16 -- weight_aux alone may just as well be the bf16x3 pack
        // of conv_math='all' (a single-image 'all' plan ran the row kernel on bf16x3 bytes).
        if ((op1.flags & op.flags & ROMP_OPF_WAVE16) && op.H % 8 == 0) return launch_bblock32r(op1, op, x, y, B, queue, st);
    }
    ROMP_REQUIRE(op.ksize == 3 && op.stride == 1 && op.Cin == 32 && op.Cout == 32 && op.cin_pad == 32 && op.cout_pad == 32 && op.groups == 1 &&
                 op1.ksize == 3 && op1.stride == 1 && op1.Cin == 32 && op1.Cout == 32 && op1.cin_pad == 32 && op1.cout_pad == 32 && op1.groups == 1,
                 "bblock32: two 3x3 stride-1 32 -> 32 convs expected");
    ROMP_REQUIRE(op1.weight_h2 && op1.scale_h2 && op.weight_h2 && op.scale_h2 && op1.relu && op.relu, "bblock32: f16x2 weights and ReLUs expected");
    ROMP_REQUIRE(op1.in_fmt == ROMP_FMT_H2 && op.res_fmt == ROMP_FMT_H2 && op.out_fmt == ROMP_FMT_H2 && op1.act_shift == op.act_shift,
                 "bblock32: H2 tensors expected");
    ROMP_REQUIRE(op.H % 16 == 0 && op.W % 16 == 0 && op1.H == op.H && op1.W == op.W, "bblock32: %dx%d is not a multiple of the 16x16 tile", op.H, op.W);
    ROMP_REQUIRE(op1.in_cstride == op.res_cstride && op1.in_coff == op.res_coff && ((op1.in_cstride | op1.in_coff | op.out_cstride | op.out_coff) & 7) == 0,
                 "bblock32: the residual must be the block input, octet aligned");
    static bool attr = false;
    static float* zero = nullptr;                              // 256 bytes of zeros: what out-of-image lanes of the halo DMA fetch
    static int num_cu = 256;
    using KernelFn = void (*)(ConvParams);
    static KernelFn fn = bblock32_kernel<0>;                    // (always counts the values it clamps: conv_common.h sat_track_pk)
    if (!attr) {                                               // (romp_net_create calls this path's setup outside any stream capture: bblock_init)
#ifdef ROMP_BBLOCK_KNOCKOUTS                                   // the timing knock-outs: developer builds only (see conv_h2c.h)
        const char* e = getenv("ROMP_CONV_DEBUG");
        switch ((e ? atoi(e) : 0) & 127) {
            case 0: break;
            case 1: fn = bblock32_kernel<1>; break;
            case 2: fn = bblock32_kernel<2>; break;
            case 4: fn = bblock32_kernel<4>; break;
            case 7: fn = bblock32_kernel<7>; break;
            case 8: fn = bblock32_kernel<8>; break;
            case 15: fn = bblock32_kernel<15>; break;
            case 71: fn = bblock32_kernel<71>; break;
            default: ROMP_REQUIRE(false, "bblock32: ROMP_CONV_DEBUG & 127 is one of 0 1 2 4 7 8 15 71 here");
        }
#endif
        ROMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, BCfg::LDS_BYTES));
        ROMP_HIP_CHECK(hipMalloc((void**)&zero, 256));
        ROMP_HIP_CHECK(hipMemset(zero, 0, 256));
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
    p.w3 = reinterpret_cast<const uint4*>(op1.weight_h2);
    p.wh = reinterpret_cast<const uint4*>(op.weight_h2);
    p.scale = op1.scale_h2; p.w = op1.shift;                   // conv1's epilogue scale / shift (see the kernel's header)
    p.scale_h = op.scale_h2; p.shift = op.shift;
    p.zero = zero;
    p.act_scale = ldexpf(1.f, op.act_shift);
    p.inv_act_scale = ldexpf(1.f, -op.act_shift);
    p.in_h2 = p.out_h2 = p.res_h2 = 1;
    p.queue = queue;
    p.trace = conv_trace_arm(st);
    p.sat = conv_sat_counter();
    {
        const unsigned long long bytes = ((unsigned long long)B * op.H * op.W * op1.in_cstride - op1.in_coff) * 4ull;
        ROMP_REQUIRE(bytes < 0x80000000ull, "bblock32: input tensor of %llu bytes: beyond the 31-bit offsets of the halo fetch", bytes);
        p.in_bytes = (unsigned)bytes;
    }
    p.H = p.Ho = op.H; p.W = p.Wo = op.W;
    p.Cout = 32; p.cin_valid = 32; p.cin_pad = 32; p.cout_pad = 32;
    p.in_cs = op1.in_cstride; p.in_co = op1.in_coff; p.in_gs = 0;
    p.out_cs = op.out_cstride; p.out_co = op.out_coff; p.out_gs = 0;
    p.res_cs = op.res_cstride; p.res_co = op.res_coff; p.res_gs = 0;
    p.relu = 1;
    p.tiles_x = op.W / 16; p.tiles_y = op.H / 16; p.tiles_total = B * p.tiles_x * p.tiles_y;
    p.nslices = p.ns_total = 1;
    p.n_queues = (p.tiles_total % 8 == 0) ? 8 : 1;
    p.per_queue = p.tiles_total / p.n_queues;
    p.vec_io = 1;
    p.pad_h = p.pad_w = 1;
    p.out_rs = op.out_rstride > 0 ? op.out_rstride : p.Wo * op.out_cstride;
    p.out_bs = op.out_bstride > 0 ? op.out_bstride : p.Ho * p.Wo * op.out_cstride;
    long grid = num_cu;                                        // one workgroup per CU
    if (grid > p.tiles_total) grid = p.tiles_total;
    if (p.n_queues == 8) grid = grid >= 8 ? (grid / 8) * 8 : 8;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(256), BCfg::LDS_BYTES, st, p);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

}  // namespace romp
