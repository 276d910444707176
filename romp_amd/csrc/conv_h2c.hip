// conv_h2c.hip -- a whole 64-channel BasicBlock (simple_romp/romp/model.py:54-83) in ONE kernel on the f16x2 split:
//     y = relu(bn2(conv3x3(relu(bn1(conv3x3(x))))) + x),   64 -> 64 -> 64 channels, stride 1, H2 tensors in and out.
// The 64-channel @64^2 class is the network's largest after the 32-channel blocks were fused (66 launches, 2.75 ms of a 12.2 ms
// forward at B = 32, 0.29 of the matrix roof as separate convs).  Same idea as conv_h2b.hip -- x read once (haloed), y written
// once, the intermediate m never leaves the CU, both convs' weights register-resident -- with the geometry the larger weights
// force:
//   * one 256-thread workgroup per CU, ONE wave per SIMD (512 registers); wave w owns OUTPUT CHANNELS 16 w .. 16 w + 15 of both
//     convs for every pixel of the tile: its share of the split weights is 2 x 144 registers (a wave that owned pixels instead
//     would need all 2 x 576).  M = 16 channels means v_mfma_f32_16x16x32_f16: A = 16 channels x 32 input channels (weights,
//     registers), B = 32 input channels x 16 pixels (an LDS fragment), D = 4 consecutive channels of one pixel per lane;
//   * tiles of 8 x 16 output pixels: 12 x 20 input halo (60 KB in LDS), 10 x 18 halo of m (48 KB);
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
// ConvParams as used here: in = x (H2), res = x, out = y (H2); w3 = conv1's weights REPACKED per wave ([wave 4][tap 9][k-chunk 2]
// [piece 2][lane 64] 16-byte units, plan.pack_h2_wave16), wh = conv2's; scale / w = conv1's f16x2 scale and shift (64 floats
// each), scale_h / shift = conv2's; the geometry fields as for a conv.
#include "conv_split.h"
#include "conv_fuse.h"
#include <string.h>

namespace romp {

struct CCfg {
    static constexpr int TH = 8, TW = 16;
    static constexpr int IR = TH + 4, IC = TW + 4;             // input halo 12 x 20
    static constexpr int MR = TH + 2, MC = TW + 2;             // intermediate halo 10 x 18
    static constexpr int XPL = IR * IC;                        // units per input plane: 240 (= 0 mod 16)
    static constexpr int MPL = 192;                            // units per m plane: 180 used, padded to 0 mod 16
    static constexpr int NPL = 16;                             // planes: 8 octets x {high, low}; plane = 2 * octet + piece
    static constexpr int NI = NPL * XPL / 64 / 4;              // DMA pieces (64 units) per wave: 15
    static constexpr int OFF_M = NPL * XPL * 16;               // 61 440
    static constexpr int OFF_R = OFF_M + NPL * MPL * 16;       // 110 592: residual parking, [wave][row 8][piece 2][lane] x 8 bytes
    static constexpr int LDS_BYTES = OFF_R + 4 * 16 * 64 * 8 + 64;
    static_assert(XPL % 16 == 0 && MPL % 16 == 0 && MPL >= MR * MC, "planes a multiple of 16 units apart");
    static_assert(NPL * XPL == 4 * NI * 64, "the DMA pieces cover the input planes exactly");
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
};

typedef float f32x4c __attribute__((ext_vector_type(4)));

// DBG: timing knock-outs (env ROMP_CONV_DEBUG, wrong outputs): 1 no halo DMA, 2 no hand-over / parking, 4 no finish, 8 no MFMA
template <int DBG>
__global__ __launch_bounds__(256, 1) void bblock64_kernel(ConvParams p) {
    using X = CCfg;
    using frag = f16x8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sBuf = reinterpret_cast<char*>(smem);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void_f*)sBuf;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, q = lane >> 4;                   // B / D operand: pixel px of the block; A: channel px; k-quarter (D: channel quad) q
    int tr_n = 0;                                              // phase stamps (ROMP_CONV_TRACE=1): 1 entry, 4 set-up done, per tile 11 conv1 MFMAs,
    constexpr int tr_wpw = 4;                                  // 13 last hand-over, 12 barrier, 17 conv2 MFMAs, 15 barrier
    ROMP_TRACE(1);
    const int qx = p.n_queues == 8 ? (blockIdx.x & 7) : 0;
    const int nwg_q = gridDim.x / p.n_queues;
    const int j0 = blockIdx.x / p.n_queues;
    if (j0 >= p.per_queue) return;
    const int n_mine = (p.per_queue - j0 + nwg_q - 1) / nwg_q;
    auto tile_of = [&](int k) __attribute__((always_inline)) { return decode_item(p, qx, j0 + k * nwg_q, 32); };

    // scale / shift of this lane's 4 channels (16 wv + 4 q ..), PRE-MULTIPLIED by 2^act_shift: m and y are produced in the scaled
    // domain the H2 pieces live in (ReLU commutes with the positive factor; the residual's pieces are x * 2^act_shift already)
    f32x4c s1, b1, s2, b2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 16 * wv + 4 * q + i;
        s1[i] = p.scale[c] * p.act_scale;   b1[i] = p.w[c] * p.act_scale;
        s2[i] = p.scale_h[c] * p.act_scale; b2[i] = p.shift[c] * p.act_scale;
    }

    // ---- halo DMA: piece 4 k + wv is this wave's k-th; a lane's unit U = 64 piece + lane -> plane U / 240, pixel U % 240
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    int d_rc[X::NI], d_off[X::NI];                             // row | col << 8;  byte offset from the halo origin
#pragma unroll
    for (int k = 0; k < X::NI; ++k) {
        const int U = (k * 4 + wv) * 64 + lane;
        const int plane = U / X::XPL, r = U % X::XPL;
        const int row = r / X::IC, col = r % X::IC;
        d_rc[k] = row | (col << 8);
        d_off[k] = ((row * p.W + col) * p.in_cs + (plane >> 1) * 8 + (plane & 1) * 4) * 4;
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
    auto fetch_piece = [&](const Item& it, bool valid, bool interior, int kk) __attribute__((always_inline)) {
        if (DBG & 1) return;
        if (!valid) return;                                    // (uniform)
        const int iy0 = it.ty * X::TH - 2, ix0 = it.tx * X::TW - 2;
        const int origin = ((it.b * p.H + iy0) * p.W + ix0) * p.in_cs * 4;     // may be "negative": the sum with d_off is not
        const unsigned dst = lds0 + (unsigned)((kk * 4 + wv) * 1024);
        int voff = d_off[kk] + origin;
        if (!interior) {
            int rc = d_rc[kk];
            asm volatile("" : "+v"(rc));
            const int iy = iy0 + (rc & 255), ix = ix0 + (rc >> 8);
            const int ok = (int)((unsigned)iy < (unsigned)p.H) & (int)((unsigned)ix < (unsigned)p.W);
            voff = ok ? voff : (int)0x80000000;
        }
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsrc), "s"(dst) : "memory");
    };

    // ---- LDS addresses (bytes).  Input plane unit (plane, row, col) = plane * 240 + row * 20 + col; m: OFF_M + plane * 192 + row * 18 + col
    // fragment of a row block: lane (px, q) reads octet 4 kc + q, piece pc of the pixel dx columns right of its own: base + immediates
    // (conv1's blocks are m columns 1..16 = input columns 1 + px + dx; conv2's are output columns px = m columns px + dx)
    const int xa = (2 * q * X::XPL + 1 + px) * 16;             // + ((8 kc + pc) * 240 + R * 20 + dx) * 16
    const int ma = X::OFF_M + (2 * q * X::MPL + px) * 16;      // + ((8 kc + pc) * 192 + R * 18 + dx) * 16
    // the two edge blocks of conv1: m pixels (row, col in {0, 17}); E0 rows 0..7 (lane px -> row px / 2, col 17 (px & 1)), E1 rows 8, 9 (px < 4)
    const int e_row = px >> 1, e_col = (px & 1) * 17;
    const bool e1_act = px < 4;
    const int xe0 = (2 * q * X::XPL + e_row * X::IC + e_col) * 16;                         // + ((8 kc + pc) * 240 + dy * 20 + dx) * 16
    const int xe1 = (2 * q * X::XPL + (e1_act ? 8 + e_row : 8) * X::IC + (e1_act ? e_col : 0)) * 16;
    // hand-over stores: lane (px, q) holds channels 16 wv + 4 q .. + 3 = half (q & 1) of octet 2 wv + q / 2
    const int mo = 2 * wv + (q >> 1);
    const int hs = X::OFF_M + (2 * mo * X::MPL + 1 + px) * 16 + (q & 1) * 8;               // row block: + (pc * 192 + r * 18) * 16
    const int hse0 = X::OFF_M + (2 * mo * X::MPL + e_row * X::MC + e_col) * 16 + (q & 1) * 8;   // + pc * 192 * 16  (E1: + 8 * 18 * 16)
    // the residual x of output pixel (r, px) in the input halo: pixel (r + 2, px + 2), same octet half
    const int ra = (2 * mo * X::XPL + 2 * X::IC + 2 + px) * 16 + (q & 1) * 8;              // + (pc * 240 + r * 20) * 16
    char* sR = sBuf + X::OFF_R + (wv * 1024 + lane) * 8;       // this lane's parking slots: + (r * 2 + pc) * 512

    Item it = tile_of(0);
#pragma unroll
    for (int kk = 0; kk < X::NI; ++kk) fetch_piece(it, true, false, kk);
    // ---- this wave's weights (asked for AFTER the first halo: both trips overlap): 16 output channels x 64 input channels x 9 taps x 2 pieces of each conv
    frag w1[9][2][2], w2[9][2][2];                             // [tap][k-chunk of 32 input channels][piece]
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                w1[tap][kc][pc] = __builtin_bit_cast(frag, p.w3[(((wv * 9 + tap) * 2 + kc) * 2 + pc) * 64 + lane]);
                w2[tap][kc][pc] = __builtin_bit_cast(frag, p.wh[(((wv * 9 + tap) * 2 + kc) * 2 + pc) * 64 + lane]);
            }
    // a "use" of every weight register in front of the tile loop: hipcc waits for these loads HERE, once; the halo DMAs (invisible to
    // it) are covered by the explicit wait.  conv1's weights are pinned to AGPRs (MFMA reads them there), conv2's stay in VGPRs.
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
            asm volatile("" : "+a"(w1[tap][kc][0]), "+a"(w1[tap][kc][1]), "+v"(w2[tap][kc][0]), "+v"(w2[tap][kc][1]));
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

    f32x4c acc2[X::TH];                                        // (the last row's outlives its tile: finished under the next tile's first MFMAs)
    Item itp = it;
#pragma unroll 1
    for (int k = 0; k < n_mine; ++k) {
        const bool has_next = k + 1 < n_mine;
        const Item itn = has_next ? tile_of(k + 1) : it;
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
            case 5: eh[0] = pack_hi(ev[0], ev[1]); eh[1] = pack_hi(ev[2], ev[3]); break;
            case 6: el[0] = h2_low_pair(eh[0], ev[0], ev[1]); el[1] = h2_low_pair(eh[1], ev[2], ev[3]); break;
            case 7: {
                // lanes (px, q even) and (px, q odd) hold channels .. + 0..3 and .. + 4..7 of an octet: after the swaps the even one
                // holds the octet's 8 high pieces, the odd one its 8 low pieces
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                const u32x2_t a = __builtin_amdgcn_permlane16_swap(eh[0], el[0], false, false);
                const u32x2_t b = __builtin_amdgcn_permlane16_swap(eh[1], el[1], false, false);
                eh[0] = a[0]; el[0] = a[1]; eh[1] = b[0]; el[1] = b[1];
                const int oy = tl.ty * X::TH + r, ox = tl.tx * X::TW + px;
                e_o = (oy * p.out_rs + ox * p.out_cs) + (16 * wv + 4 * q);
                break; }
            default: {
                float* o = p.out + (size_t)tl.b * p.out_bs + p.out_co + (unsigned)e_o;
                if (live) *reinterpret_cast<uint4*>(o) = make_uint4(eh[0], eh[1], el[0], el[1]);
                break; }
            }
        };
        // ---- residual parking: 16 copies (row r, piece pc) from the input halo, read at step i, written three steps later
        uint2 pk[4];
        constexpr int PARK_N = 19;
        auto park_micro = [&](int t) __attribute__((always_inline)) {
            if (DBG & 2) return;
            if (t < 16) pk[t % 4] = *reinterpret_cast<const uint2*>(sBuf + ra + ((t & 1) * X::XPL + (t >> 1) * X::IC) * 16);
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
            case 2: hh[0] = pack_hi(hv[0], hv[1]); hh[1] = pack_hi(hv[2], hv[3]); break;
            case 3: hl[0] = h2_low_pair(hh[0], hv[0], hv[1]); hl[1] = h2_low_pair(hh[1], hv[2], hv[3]); break;
            case 4: if (act) *reinterpret_cast<uint2*>(sBuf + addr) = make_uint2(hh[0], hh[1]); break;
            default: if (act) *reinterpret_cast<uint2*>(sBuf + addr + X::MPL * 16) = make_uint2(hl[0], hl[1]); break;
            }
        };
        const int iy_m0 = it.ty * X::TH - 1;                    // image row of m row 0
        const int ix_e = it.tx * X::TW - 1 + e_col;             // image column of this lane's edge-block pixel

        // ================= 1. conv1
        f32x4c accE[2], acc1[X::MR];
#pragma unroll
        for (int i = 0; i < 4; ++i) accE[0][i] = accE[1][i] = 0.f;
#pragma unroll
        for (int r = 0; r < X::MR; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc1[r][i] = 0.f;
        {
            // (a) the two edge blocks: no row reuse (54 MFMAs each).  Under them: the previous tile's last output row, then this
            // tile's residual parking
            // fragment reads run PF units (a unit = the MFMAs fed by one fragment pair) ahead of their MFMAs: unit u < 36 is edge block
            // u % 2, tap u / 4, k-chunk (u / 2) % 2; unit 36 + ((R * 3 + dx) * 2 + kc) the row-block fragment of input row R
            constexpr int PF = 3, NU = 36 + X::IR * 6, RING = PF + 2;   // (+ 2: the edge blocks consume two units at a time)
            frag xf[RING][2];
            auto read_x = [&](int u) __attribute__((always_inline)) {
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    if (u < 36) {
                        const int eb = u % 2, tap = u / 4, kc = (u / 2) % 2;
                        xf[u % RING][pc] = *reinterpret_cast<const frag*>(sBuf + (eb ? xe1 : xe0) + ((8 * kc + pc) * X::XPL + (tap / 3) * X::IC + tap % 3) * 16);
                    } else {
                        const int v = u - 36, R = v / 6, dx = (v % 6) / 2, kc = v % 2;
                        xf[u % RING][pc] = *reinterpret_cast<const frag*>(sBuf + xa + ((8 * kc + pc) * X::XPL + R * X::IC + dx) * 16);
                    }
                }
            };
#pragma unroll
            for (int u = 0; u < PF; ++u) read_x(u);
            constexpr int GE = 108, NE = FIN_N + PARK_N;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const int u = (tap * 2 + kc) * 2;          // units u (block E0) and u + 1 (E1): their MFMAs take turns (no two in a row on one accumulator)
                    if (u + PF < NU) read_x(u + PF);
                    if (u + 1 + PF < NU) read_x(u + 1 + PF);
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int eb = 0; eb < 2; ++eb) {
                            const frag (&x)[2] = xf[(u + eb) % RING];
                            if (!(DBG & 8))
                                accE[eb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[tap][kc][pr == 0 ? 1 : 0], x[pr == 1 ? 1 : 0], accE[eb], 0, 0, 0);
                            const int g = ((tap * 2 + kc) * 3 + pr) * 2 + eb;
                            int lo, hi;
                            share(g, GE, NE, lo, hi);
#pragma unroll
                            for (int t = lo; t < hi; ++t) { if (t < FIN_N) fin_micro(itp, k > 0, X::TH - 1, t); else park_micro(t - FIN_N); }
                            SIDE_PIN();
                        }
                }
            // (b) the ten row blocks, input row after input row.  Under input row R: the hand-over of the edge blocks (R = 0, 1), then of
            // m row R - 3 (complete since input row R - 1)
#pragma unroll
            for (int R = 0; R < X::IR; ++R) {
                const int dy_lo = R - (X::MR - 1) > 0 ? R - (X::MR - 1) : 0, dy_hi = R < 2 ? R : 2;      // m rows R - dy in [0, 10)
                const int nv = dy_hi - dy_lo + 1;
                const int G = 6 * nv * 3;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) {
                        const int u = 36 + (R * 3 + dx) * 2 + kc;
                        if (u + PF < NU) read_x(u + PF);
                        const frag (&x)[2] = xf[u % RING];
#pragma unroll
                        for (int pr = 0; pr < 3; ++pr)          // (products outside, rows inside: consecutive MFMAs on different accumulators)
#pragma unroll
                            for (int dy = dy_lo; dy <= dy_hi; ++dy) {
                                const int tap = dy * 3 + dx;
                                if (!(DBG & 8))
                                    acc1[R - dy] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[tap][kc][pr == 0 ? 1 : 0], x[pr == 1 ? 1 : 0], acc1[R - dy], 0, 0, 0);
                                const int g = (((dx * 2 + kc) * 3) + pr) * nv + (dy - dy_lo);
                                int lo, hi;
                                share(g, G, HAND_N, lo, hi);
#pragma unroll
                                for (int t = lo; t < hi; ++t) {
                                    if (R < 2) {
                                        const int iy = iy_m0 + (R ? 8 : 0) + e_row;
                                        const bool in = (R == 0 || e1_act) && (unsigned)iy < (unsigned)p.Ho && (unsigned)ix_e < (unsigned)p.Wo;
                                        hand_micro(accE[R], in, R == 0 || e1_act, hse0 + (R ? 8 * X::MC * 16 : 0), t);
                                    } else if (R >= 3) {
                                        const int r = R - 3;
                                        // (m rows 1..8 are output rows -1..8 of the tile + 1: always inside the image; row 0 may lie above it)
                                        hand_micro(acc1[r], r == 0 ? iy_m0 >= 0 : true, true, hs + r * X::MC * 16, t);
                                    }
                                }
                                SIDE_PIN();
                            }
                    }
            }
            ROMP_TRACE(11);
            if (DBG & 2) {                                     // (knock-out builds: keep every MFMA)
#pragma unroll
                for (int r = 0; r < X::MR; ++r) asm volatile("" :: "v"(acc1[r]));
                asm volatile("" :: "v"(accE[0]), "v"(accE[1]));
            }
#pragma unroll
            for (int t = 0; t < HAND_N; ++t) hand_micro(acc1[X::MR - 1], (unsigned)(iy_m0 + X::MR - 1) < (unsigned)p.Ho, true, hs + (X::MR - 1) * X::MC * 16, t);
            ROMP_TRACE(13);
        }
        // ---- 2. every wave is done with the input halo and m is complete
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ROMP_TRACE(12);

        // ================= 3. conv2 from m, m row after m row.  Under m row R: a share of the next tile's halo DMA and the finish of
        // output row R - 3 (complete since m row R - 1); output row 7 waits for the next tile
#pragma unroll
        for (int r = 0; r < X::TH; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc2[r][i] = 0.f;
        constexpr int PF2 = 2, NU2 = X::MR * 6;                  // unit (R * 3 + dx) * 2 + kc
        frag xg[PF2 + 1][2];
        auto read_m = [&](int u) __attribute__((always_inline)) {
            const int R = u / 6, dx = (u % 6) / 2, kc = u % 2;
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
                xg[u % (PF2 + 1)][pc] = *reinterpret_cast<const frag*>(sBuf + ma + ((8 * kc + pc) * X::MPL + R * X::MC + dx) * 16);
        };
#pragma unroll
        for (int u = 0; u < PF2; ++u) read_m(u);
#pragma unroll
        for (int R = 0; R < X::MR; ++R) {
            const int dy_lo = R - (X::TH - 1) > 0 ? R - (X::TH - 1) : 0, dy_hi = R < 2 ? R : 2;          // output rows R - dy in [0, 8)
            const int nv = dy_hi - dy_lo + 1;
            const int G = 6 * nv * 3;
            // DMA pieces of this m row: 15 over the 10 rows (two under the long rows, one under the short ones)
            const int f_lo = R * X::NI / X::MR, f_hi = (R + 1) * X::NI / X::MR;
            const int NS = (R >= 3 ? FIN_N : 0) + (f_hi - f_lo);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const int u = (R * 3 + dx) * 2 + kc;
                    if (u + PF2 < NU2) read_m(u + PF2);
                    const frag (&x)[2] = xg[u % (PF2 + 1)];
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int dy = dy_lo; dy <= dy_hi; ++dy) {
                            const int tap = dy * 3 + dx;
                            if (!(DBG & 8))
                                acc2[R - dy] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[tap][kc][pr == 0 ? 1 : 0], x[pr == 1 ? 1 : 0], acc2[R - dy], 0, 0, 0);
                            const int g = (((dx * 2 + kc) * 3) + pr) * nv + (dy - dy_lo);
                            int lo, hi;
                            share(g, G, NS, lo, hi);
#pragma unroll
                            for (int t = lo; t < hi; ++t) {
                                // interleave: DMA pieces first in the row's list, then the finish steps
                                if (t < f_hi - f_lo) fetch_piece(itn, has_next, next_interior, f_lo + t);
                                else fin_micro(it, true, R - 3, t - (f_hi - f_lo));
                            }
                            SIDE_PIN();
                        }
                }
        }
        ROMP_TRACE(17);
        if (DBG & 4) {
#pragma unroll
            for (int r = 0; r < X::TH; ++r) asm volatile("" :: "v"(acc2[r]));
        }
        // ---- 4. the next halo has landed, for every wave; m may be overwritten
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ROMP_TRACE(15);
        itp = it;
        it = itn;
    }
    if (!(DBG & 4)) {                                          // the last tile's last output row
        const int r = X::TH - 1;
        const uint2 rh = *reinterpret_cast<const uint2*>(sR + (r * 2 + 0) * 512), rl = *reinterpret_cast<const uint2*>(sR + (r * 2 + 1) * 512);
        float ev[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned wh = e < 2 ? rh.x : rh.y, wl = e < 2 ? rl.x : rl.y;
            const float v = fmaf(acc2[r][e], s2[e], b2[e]);
            ev[e] = (e & 1) ? add_pieces_relu<1>(v, wh, wl, H2_MAX) : add_pieces_relu<0>(v, wh, wl, H2_MAX);
        }
        unsigned eh[2] = {pack_hi(ev[0], ev[1]), pack_hi(ev[2], ev[3])};
        unsigned el[2] = {h2_low_pair(eh[0], ev[0], ev[1]), h2_low_pair(eh[1], ev[2], ev[3])};
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t a = __builtin_amdgcn_permlane16_swap(eh[0], el[0], false, false);
        const u32x2_t b = __builtin_amdgcn_permlane16_swap(eh[1], el[1], false, false);
        const int oy = itp.ty * X::TH + r, ox = itp.tx * X::TW + px;
        float* o = p.out + (size_t)itp.b * p.out_bs + p.out_co + (unsigned)((oy * p.out_rs + ox * p.out_cs) + (16 * wv + 4 * q));
        *reinterpret_cast<uint4*>(o) = make_uint4(a[0], b[0], a[1], b[1]);
    }
#undef SIDE_PIN
}

// `op` is the block's SECOND conv (its residual is the block input x, its output y); `op1` the first.  Their per-wave weight packs
// (plan.pack_h2_wave16) are in `weight_aux`.
int launch_bblock64(const romp_op& op1, const romp_op& op, const float* x, float* y, int B, int* queue, hipStream_t st) {
    ROMP_REQUIRE(op.ksize == 3 && op.stride == 1 && op.Cin == 64 && op.Cout == 64 && op.groups == 1 &&
                 op1.ksize == 3 && op1.stride == 1 && op1.Cin == 64 && op1.Cout == 64 && op1.groups == 1,
                 "bblock64: two 3x3 stride-1 64 -> 64 convs expected");
    ROMP_REQUIRE(op1.weight_aux && op1.scale_h2 && op.weight_aux && op.scale_h2 && op1.relu && op.relu, "bblock64: per-wave f16x2 weight packs and ReLUs expected");
    ROMP_REQUIRE(op1.in_fmt == ROMP_FMT_H2 && op.res_fmt == ROMP_FMT_H2 && op.out_fmt == ROMP_FMT_H2 && op1.act_shift == op.act_shift,
                 "bblock64: H2 tensors expected");
    ROMP_REQUIRE(op.H % 8 == 0 && op.W % 16 == 0 && op1.H == op.H && op1.W == op.W, "bblock64: %dx%d is not a multiple of the 8x16 tile", op.H, op.W);
    ROMP_REQUIRE(op1.in_cstride == op.res_cstride && op1.in_coff == op.res_coff && ((op1.in_cstride | op1.in_coff | op.out_cstride | op.out_coff) & 7) == 0,
                 "bblock64: the residual must be the block input, octet aligned");
    static bool attr = false;
    static int num_cu = 256;
    using KernelFn = void (*)(ConvParams);
    static KernelFn fn = bblock64_kernel<0>;
    if (!attr) {                                               // (romp_net_create calls this path's set-up outside any stream capture)
        const char* e = getenv("ROMP_CONV_DEBUG");
        switch (e ? atoi(e) : 0) {
            case 0: break;
            case 1: fn = bblock64_kernel<1>; break;
            case 2: fn = bblock64_kernel<2>; break;
            case 4: fn = bblock64_kernel<4>; break;
            case 7: fn = bblock64_kernel<7>; break;
            case 8: fn = bblock64_kernel<8>; break;
            default: ROMP_REQUIRE(false, "bblock64: ROMP_CONV_DEBUG is one of 0 1 2 4 7 8 here");
        }
        ROMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, CCfg::LDS_BYTES));
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
    {
        const unsigned long long bytes = ((unsigned long long)B * op.H * op.W * op1.in_cstride - op1.in_coff) * 4ull;
        ROMP_REQUIRE(bytes < 0x80000000ull, "bblock64: input tensor of %llu bytes: beyond the 31-bit offsets of the halo fetch", bytes);
        p.in_bytes = (unsigned)bytes;
    }
    p.H = p.Ho = op.H; p.W = p.Wo = op.W;
    p.Cout = 64; p.cin_valid = 64; p.cin_pad = 64; p.cout_pad = 64;
    p.in_cs = op1.in_cstride; p.in_co = op1.in_coff;
    p.out_cs = op.out_cstride; p.out_co = op.out_coff;
    p.res_cs = op.res_cstride; p.res_co = op.res_coff;
    p.relu = 1;
    p.tiles_x = op.W / CCfg::TW; p.tiles_y = op.H / CCfg::TH; p.tiles_total = B * p.tiles_x * p.tiles_y;
    p.nslices = p.ns_total = 1;
    p.n_queues = (p.tiles_total % 8 == 0) ? 8 : 1;
    p.per_queue = p.tiles_total / p.n_queues;
    p.tile_contig = 1;
    p.vec_io = 1;
    p.pad_h = p.pad_w = 1;
    p.out_rs = op.out_rstride > 0 ? op.out_rstride : p.Wo * op.out_cstride;
    p.out_bs = op.out_bstride > 0 ? op.out_bstride : p.Ho * p.Wo * op.out_cstride;
    long grid = num_cu;                                        // one workgroup per CU
    if (grid > p.tiles_total) grid = p.tiles_total;
    if (p.n_queues == 8) grid = grid >= 8 ? (grid / 8) * 8 : 8;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(256), CCfg::LDS_BYTES, st, p);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

}  // namespace romp
