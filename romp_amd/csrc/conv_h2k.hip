// conv_h2k.hip -- 3x3 (stride 1 | 2) and 1x1 convolution on the f16x2 split with the INPUT CHANNELS split across the waves of a workgroup ("h2k",
// round 4): the single-image form of conv_h2r.hip.
//
// At batch 1 the deep layers have few pixels (32^2 x 128, 16^2 x 256 channels): a 64-pixel x 128-channel workgroup tile is 16 work items
// with an 8- or 16-stage serial channel loop.  Rounds 2-3 bought parallelism by lowering such a layer as a grouped conv over G
// input-channel slices writing float32 partial tensors plus a `ksum` launch that adds them and applies the epilogue: 171 extra
// launches of ~3 us on a launch-bound chain ("split-K reduced inside the workgroup").  Here the four waves of a
// workgroup ARE the K-split:
//   * workgroup tile = P x 32 pixels x 32 output channels; wave w walks the 16-channel chunks w, w + 4, w + 8, .. -- a "super-stage" is
//     64 input channels, a 128-channel layer is two of them instead of eight stages;
//   * a wave's pixel chunk (LDS-DMA, the rotated layout of conv_h2r.hip) and its weights (72 registers, re-loaded tap by tap for its
//     next chunk) are PRIVATE to the wave: no barrier inside the channel loop, each wave paces itself with its own vmcnt;
//   * at the end of an item the four partial accumulators meet in LDS (one barrier), wave w sums channel octet w of every pixel and
//     runs the direct H2 epilogue (conv_common.h) for it: BN (+ residual) (+ ReLU), split, one 16-byte store per lane.
// Work items = B x (H x W / (32 P)) x Cout / 32: 128 for a 128-channel 32^2 layer at B = 1, 256 for 64 channels at 64^2.
// H2 tensors in and out (the residual, if any, H2 too); cin a multiple of 64.
#include "conv_split.h"

namespace romp {

template <int KS, int S, int P, int TW>                        // KS = 3: 3x3, 1: 1x1 (the 1x1 up-convs of a single-image plan); S: stride 1 | 2
struct KCfg {
    static constexpr int NWV = 4, TAPS = KS * KS;
    using C = ConvCfg<KS, S, P, 1, TW, 16, 1>;                 // TH = P * (32 / TW) rows, NW = 32 channels
    static constexpr int CG = (C::HC + 3) / 4;                 // 4-pixel column groups per haloed row
    static constexpr int RSU = CG * 16;                        // 16-byte units per haloed row
    static constexpr int NIW = (C::HR * RSU + 63) / 64;        // DMA pieces (wave-instructions of 1 KiB) of ONE wave's chunk
    static constexpr int SUB_BYTES = NIW * 1024;               // a wave's chunk buffer
    static constexpr int STAGE_BYTES = NWV * SUB_BYTES;        // a super-stage: the four waves' chunks
    static constexpr int OFF_R = 2 * STAGE_BYTES;              // reduction tiles: [block][wave][channel quad g4][lane] float4
    static constexpr int RED_BYTES = P * NWV * 4 * 64 * 16;
    static constexpr int LDS_BYTES = OFF_R + RED_BYTES + 16;
    static_assert((KS == 1 || KS == 3) && (S == 1 || S == 2), "1x1 or 3x3, stride 1 or 2");
    static_assert((C::HR - 1) * RSU * 16 + RSU * 16 < 65536, "fragment read offsets are ds_read immediates");
};

typedef __attribute__((address_space(3))) void lds_void_k;
typedef const __attribute__((address_space(1))) void glb_void_k;

struct KStage {                 // wave-uniform description of one of this wave's chunks
    const float* in;            // image + group + chunk base of the pixel tensor
    const uint4* wg;            // group + chunk + channel slice of the split weights
    int iy0, ix0, c0;
};

template <int KS, int S, int P, int TW>
__global__ __launch_bounds__(256, 2) void conv_h2k_kernel(ConvParams p) {
    conv_args_now(p);
    if (p.dbg & 32) return;
    using X = KCfg<KS, S, P, TW>;
    constexpr int TAPS = X::TAPS;
    using C = typename X::C;
    using frag = f16x8;
    typedef unsigned u32x2_k __attribute__((ext_vector_type(2)));
    constexpr int NWV = X::NWV;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sBuf = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int q = p.n_queues == 8 ? (blockIdx.x & 7) : 0;
    const int n_super = p.cin_pad >> 6;                        // super-stages: 64 input channels each
    const int cin16 = p.cin_pad >> 4;
    const int nwg_q = gridDim.x / p.n_queues;
    const int j_cur0 = blockIdx.x / p.n_queues;
    if (j_cur0 >= p.per_queue) return;

    // ---- per-lane DMA descriptors of the wave's chunk (the same for every stage): conv_h2r.hip's rotated unit layout
    int d_rc[X::NIW];                                          // row | col << 8 | inside-the-tile << 16 | unit w << 17
#pragma unroll
    for (int k = 0; k < X::NIW; ++k) {
        const int U = k * 64 + lane;
        const int row = U / X::RSU, r = U % X::RSU;
        const int cg = r >> 4, r16 = r & 15;
        const int col = cg * 4 + (r16 & 3), w = ((r16 >> 2) - cg) & 3;
        d_rc[k] = row | (col << 8) | ((row < C::HR && col < C::HC) ? 1 << 16 : 0) | (w << 17);
    }
    auto make_desc = [&](const Item& it, int s) {              // this wave's chunk of super-stage s
        KStage d;
        d.c0 = (s * NWV + wave) * 16;
        d.in = p.in + (size_t)it.b * p.H * p.W * p.in_cs + p.in_co + it.g * p.in_gs + d.c0;
        d.wg = p.wh + (size_t)it.g * (TAPS * cin16 * 4 * p.cout_pad) + (d.c0 >> 4) * 4 * p.cout_pad + it.n0;
        d.iy0 = it.ty * C::TH * S - p.pad_h;
        d.ix0 = it.tx * TW * S - p.pad_w;
        return d;
    };
    auto issue_piece = [&](int k, const KStage& d, int buf) {
        int rc = d_rc[k];
        asm volatile("" : "+v"(rc));
        const int row = rc & 255, col = (rc >> 8) & 255, w = (rc >> 17) & 3;
        const int iy = d.iy0 + row, ix = d.ix0 + col;
        const int ok = ((rc >> 16) & 1) & (int)((unsigned)iy < (unsigned)p.H) & (int)((unsigned)ix < (unsigned)p.W) &
                       (int)(d.c0 + (w >> 1) * 8 < p.cin_valid);
        const unsigned long long a_in = (unsigned long long)(d.in + ((iy * p.W + ix) * p.in_cs + w * 4));
        const unsigned long long a = ok ? a_in : (unsigned long long)p.zero;
        __builtin_amdgcn_global_load_lds((glb_void_k*)a, (lds_void_k*)(sBuf + buf * X::STAGE_BYTES + wave * X::SUB_BYTES + k * 1024), 16, 0, 0);
    };
    frag wreg[TAPS][2];
    const unsigned w_lane = (unsigned)(lh * p.cout_pad + li);
    const unsigned w_tap = (unsigned)(cin16 * 4 * p.cout_pad), w_pc = (unsigned)(2 * p.cout_pad);
    auto load_w = [&](const uint4*& wp, int tap) {
        wreg[tap][0] = __builtin_bit_cast(frag, wp[0]);
        wreg[tap][1] = __builtin_bit_cast(frag, wp[w_pc]);
        wp += w_tap;
    };
    int xa[KS][2];                                             // fragment addresses of block 0: input pixel (S row, S col + dx), unit 2 lh + piece
    {                                                          // (stride 2 reads every other pixel of the rotated layout: two-way bank conflicts,
        const int prow = li / TW, pcol = li % TW;              //  which a single image's latency-bound launches do not notice)
#pragma unroll
        for (int dx = 0; dx < KS; ++dx)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const int col = pcol * S + dx, w = lh * 2 + pc;
                xa[dx][pc] = (prow * S * X::RSU + (col >> 2) * 16 + (col & 3) + 4 * ((w + (col >> 2)) & 3)) * 16;
            }
    }

    Item cur = decode_item(p, q, j_cur0, 32);
    {
        const KStage d0 = make_desc(cur, 0);
#pragma unroll
        for (int k = 0; k < X::NIW; ++k) issue_piece(k, d0, 0);
        const uint4* wp0 = d0.wg + w_lane;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) load_w(wp0, tap);
    }
    f32x16 acc[P];
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    int j_next = j_cur0 + nwg_q;
    Item nxt = cur;
    bool have_next = j_next < p.per_queue;
    if (have_next) nxt = decode_item(p, q, j_next, 32);
    int s = 0, buf = 0;

#pragma unroll 1
    while (true) {
        const bool last = s + 1 == n_super;
        // this wave's next chunk (of this item, or the first of the next; a workgroup's final stage re-fetches itself: harmless)
        const KStage nd = make_desc(last ? (have_next ? nxt : cur) : cur, last ? (have_next ? 0 : s) : s + 1);
        // the epilogue's operands of this item, asked for before the last chunk's MFMAs: scale | shift of the wave's channel quad pair
        // (channels n0 + 8 wave + 4 lh .. + 3) and the H2 residual unit (half-wave lh) of octet `wave` of each block's pixel
        float4 e_sc, e_sh;
        uint4 e_ru[P];
        if (last) {
            const int c = cur.g * p.cout_pad + cur.n0 + 8 * wave + 4 * lh;
            e_sc = *reinterpret_cast<const float4*>(p.scale_h + c);
            e_sh = *reinterpret_cast<const float4*>(p.shift + c);
            if (p.res) {
                const float* res = p.res + (size_t)cur.b * p.Ho * p.Wo * p.res_cs + p.res_co + cur.g * p.res_gs + cur.n0 + 8 * wave + 4 * lh;
#pragma unroll
                for (int m = 0; m < P; ++m) {
                    const int oy = cur.ty * C::TH + m * C::RPB + li / TW, ox = cur.tx * TW + li % TW;
                    const unsigned pix = oy < p.Ho ? (unsigned)(oy * p.Wo + ox) : 0u;
                    e_ru[m] = *reinterpret_cast<const uint4*>(res + pix * (unsigned)p.res_cs);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's chunk (and its weights) have landed: a private wait, no barrier
        // (the waits above also cover e_sc / e_ru: they are not needed before the epilogue, but a counted wait would have to know how
        // many of them there are; the loads are L2 hits issued a stage's MFMAs ahead of the next wait in all but one-stage layers)
        {
            const char* sA = sBuf + buf * X::STAGE_BYTES + wave * X::SUB_BYTES;
            const uint4* wp = nd.wg + w_lane;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int dy = tap / KS, dx = tap % KS;
                frag x[P][2];
#pragma unroll
                for (int j = 0; j < P; ++j)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc)
                        x[j][pc] = *reinterpret_cast<const frag*>(sA + xa[dx][pc] + (j * C::RPB * S + dy) * (X::RSU * 16));
#pragma unroll
                for (int j = 0; j < P; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[tap][1], x[j][0], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < P; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[tap][0], x[j][1], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < P; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[tap][0], x[j][0], acc[j], 0, 0, 0);
                load_w(wp, tap);                               // the tap's registers take the next chunk's weights
                if (tap < X::NIW) issue_piece(tap, nd, buf ^ 1);
            }
#pragma unroll
            for (int k = TAPS; k < X::NIW; ++k) issue_piece(k, nd, buf ^ 1);      // (1x1: more pieces than taps)
        }
        buf ^= 1;
        if (!last) { ++s; continue; }
        // ---- the four partial sums meet: every wave parks its accumulators, wave w sums channel quads {w} of every pixel
        float* sR = reinterpret_cast<float*>(sBuf + X::OFF_R);
#pragma unroll
        for (int m = 0; m < P; ++m)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<float4*>(sR + (((m * NWV + wave) * 4 + g4) * 64 + lane) * 4) =
                    make_float4(acc[m][g4 * 4 + 0], acc[m][g4 * 4 + 1], acc[m][g4 * 4 + 2], acc[m][g4 * 4 + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float v[P][4];
#pragma unroll
        for (int m = 0; m < P; ++m) {
            float4 t = *reinterpret_cast<const float4*>(sR + (((m * NWV + 0) * 4 + wave) * 64 + lane) * 4);
#pragma unroll
            for (int ww = 1; ww < NWV; ++ww) {                  // (wave order: deterministic)
                const float4 u = *reinterpret_cast<const float4*>(sR + (((m * NWV + ww) * 4 + wave) * 64 + lane) * 4);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            v[m][0] = t.x; v[m][1] = t.y; v[m][2] = t.z; v[m][3] = t.w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // the tiles may be overwritten by the next item
        // ---- epilogue of channel octet `wave` (conv_epilogue_h2direct for one g4): scaled domain, residual by half-wave swap
        {
            const float lo_b = (p.relu && cur.n0 >= p.relu_from) ? 0.f : -H2_MAX;
            const float sc[4] = {e_sc.x * p.act_scale, e_sc.y * p.act_scale, e_sc.z * p.act_scale, e_sc.w * p.act_scale};
            const float sh[4] = {e_sh.x * p.act_scale, e_sh.y * p.act_scale, e_sh.z * p.act_scale, e_sh.w * p.act_scale};
            float* out = p.out + (size_t)cur.b * p.out_bs + p.out_co + cur.g * p.out_gs + cur.n0 + 8 * wave + 4 * lh;
            float sat_mx = 0.f;
#pragma unroll
            for (int m = 0; m < P; ++m) {
                const int oy = cur.ty * C::TH + m * C::RPB + li / TW, ox = cur.tx * TW + li % TW;
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = fmaf(v[m][e], sc[e], sh[e]);
                if (p.res) {
                    const u32x2_k s0 = __builtin_amdgcn_permlane32_swap(e_ru[m].x, e_ru[m].z, false, false);
                    const u32x2_k s1 = __builtin_amdgcn_permlane32_swap(e_ru[m].y, e_ru[m].w, false, false);
                    y[0] = h2_add_pieces_clamp(y[0], s0[0], s0[1], 0, lo_b, H2_MAX);
                    y[1] = h2_add_pieces_clamp(y[1], s0[0], s0[1], 1, lo_b, H2_MAX);
                    y[2] = h2_add_pieces_clamp(y[2], s1[0], s1[1], 0, lo_b, H2_MAX);
                    y[3] = h2_add_pieces_clamp(y[3], s1[0], s1[1], 1, lo_b, H2_MAX);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = __builtin_amdgcn_fmed3f(y[e], lo_b, H2_MAX);
                }
                sat_track(sat_mx, y[0], y[1]);
                sat_track(sat_mx, y[2], y[3]);
                const unsigned h0 = h2_high_pair(y[0], y[1]), h1 = h2_high_pair(y[2], y[3]);
                const unsigned l0 = h2_low_pair(h0, y[0], y[1]), l1 = h2_low_pair(h1, y[2], y[3]);
                const u32x2_k a = __builtin_amdgcn_permlane32_swap(h0, l0, false, false);
                const u32x2_k b = __builtin_amdgcn_permlane32_swap(h1, l1, false, false);
                if (oy < p.Ho) *reinterpret_cast<uint4*>(out + (unsigned)(oy * p.out_rs + ox * p.out_cs)) = make_uint4(a[0], b[0], a[1], b[1]);
            }
            sat_report(p.sat, sat_mx);
        }
#pragma unroll
        for (int j = 0; j < P; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        if (!have_next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the self re-fetch must not outlive the workgroup's LDS
            break;
        }
        cur = nxt;
        s = 0;
        j_next += nwg_q;
        have_next = j_next < p.per_queue;
        if (have_next) nxt = decode_item(p, q, j_next, 32);
    }
}

// math 10: the input channels split across the workgroup's waves; `ck` = 64 (a super-stage)
#define ROMP_CONV_VARIANT_H2K(KS, S, P, TW)                                                            \
    { KS, S, P, 1, TW, 64, conv_h2k_kernel<KS, S, P, TW>, KCfg<KS, S, P, TW>::LDS_BYTES, KCfg<KS, S, P, TW>::C::TH, 0, 0, 10, 256 }

static ConvVariant kVariantsH2k[] = { ROMP_CONV_VARIANT_H2K(3, 1, 1, 16), ROMP_CONV_VARIANT_H2K(3, 1, 2, 16), ROMP_CONV_VARIANT_H2K(1, 1, 1, 16),
                                      ROMP_CONV_VARIANT_H2K(3, 2, 1, 16) };
ConvVariant* conv_variants_h2k(int* n) { *n = (int)(sizeof(kVariantsH2k) / sizeof(kVariantsH2k[0])); return kVariantsH2k; }

}  // namespace romp
