// conv_split.h -- the split-precision (bf16x3 / f16x2) conv kernels; instantiated by conv_bx3.hip, conv_h2.hip, conv_h2d.hip.
#pragma once
#include "conv_common.h"

namespace romp {

// ------------------------------------------------------------------------------------------------
// Split-precision variants: the same persistent implicit GEMM on the 16-bit matrix pipe at float32
// accuracy.  Every f32 operand is split exactly into NP low-precision pieces and the product is
// rebuilt from the piece products that matter, accumulated in f32 by the MFMA.  Two families:
//
//   NP = 3, bf16 ("bf16x3"):  x = x1 + x2 + x3 (3 x 8 significand bits); six products of weight
//       >= 2^-16 (x1w1, x1w2, x2w1, x1w3, x2w2, x3w1; the dropped ones are below the f32 rounding of
//       the product).  6 x 32 cycles per 32x32x16 block instead of 8 x 64 cycles on the f32 pipe.
//   NP = 2, fp16 ("f16x2"):   x = h1 + h2 (2 x 11 significand bits, residual <= 2^-22 |x|); three
//       products h1w1 + h1w2 + h2w1 (the dropped h2w2 is 2^-22 relative).  3 x 32 cycles per block:
//       twice the bf16x3 rate.  fp16's narrow exponent range is handled with exact power-of-two
//       scales: the weights are pre-scaled per group on the host (plan.py:pack_conv_weight_h2), the
//       activations by 2^act_shift while they are split, and the epilogue scale carries the inverse.
//       Whole-network error against the f32 reference (scripts/precision_emul.py, CPU emulation of
//       exactly this arithmetic): 3.3e-6 on the maps, the same as f32 itself (2.3e-6 vs f64) --
//       bf16x3 measures 2.6e-6, plain fp16 operands 3.2e-3.
//
// Activations stay f32 in HBM; they are split while being staged into LDS, the weights are pre-split
// on the host.  LDS pixel layout: [row][pixel][piece][CK] 16-bit values, pixel and row strides padded
// so that the 16-lane groups of the ds_read_b128 fragment reads touch 16 distinct 16-byte bank slots
// (best_pads below replays the bank model of MI355X_MICROARCH.md at compile time).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NP> struct Piece;
template <> struct Piece<3> {
    using frag = bf16x8;
    static __device__ __forceinline__ void split(float x, unsigned short (&h)[3]) {
        const __bf16 b1 = (__bf16)x;
        const float r1 = x - (float)b1;                        // exact
        const __bf16 b2 = (__bf16)r1;
        const float r2 = r1 - (float)b2;                       // exact
        const __bf16 b3 = (__bf16)r2;
        h[0] = __builtin_bit_cast(unsigned short, b1);
        h[1] = __builtin_bit_cast(unsigned short, b2);
        h[2] = __builtin_bit_cast(unsigned short, b3);
    }
    static __device__ __forceinline__ f32x16 mma(const frag (&w)[3], const frag (&x)[3], f32x16 acc) {   // smallest terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], x[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], acc, 0, 0, 0);
        return acc;
    }
};
template <> struct Piece<2> {
    using frag = f16x8;
    static __device__ __forceinline__ void split(float x, unsigned short (&h)[2]) {
        x = h2_sat(x);                                         // saturate instead of inf / NaN pieces (conv_common.h)
        const _Float16 h1 = (_Float16)x;                       // round to nearest even
        const float r1 = x - (float)h1;                        // exact
        const _Float16 h2 = (_Float16)r1;
        h[0] = __builtin_bit_cast(unsigned short, h1);
        h[1] = __builtin_bit_cast(unsigned short, h2);
    }
    static __device__ __forceinline__ f32x16 mma(const frag (&w)[2], const frag (&x)[2], f32x16 acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1], x[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], x[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], x[0], acc, 0, 0, 0);
        return acc;
    }
};

// LDS cycles of one half-wave ds_read_b128 of the pixel fragments (2 = conflict-free): lanes of a 16-lane
// service group that fall on the same 16-byte bank slot with different addresses serialise.
constexpr int lds_read_cycles(int psb, int rowb, int tw, int s) {
    const int groups[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                               {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    int tot = 0;
    for (int g = 0; g < 2; ++g) {
        int addr[16] = {}, worst = 0;
        for (int i = 0; i < 16; ++i) {
            const int li = groups[g][i];
            addr[i] = (li / tw) * s * rowb + (li % tw) * s * psb;
        }
        for (int i = 0; i < 16; ++i) {
            int n = 0;                                   // distinct addresses on addr[i]'s slot, counted at its first lane
            bool first = true;
            for (int j = 0; j < i; ++j)
                if ((addr[j] / 16) % 16 == (addr[i] / 16) % 16) first = false;
            if (!first) continue;
            for (int j = i; j < 16; ++j) {
                if ((addr[j] / 16) % 16 != (addr[i] / 16) % 16) continue;
                bool dup = false;
                for (int k = i; k < j; ++k)
                    if (addr[k] == addr[j]) dup = true;
                if (!dup) ++n;
            }
            if (n > worst) worst = n;
        }
        tot += worst;
    }
    return tot;
}
struct LdsPads { int pixel, row; };
constexpr LdsPads best_pads(int base, int hc, int tw, int s) {
    LdsPads best{16, 0};
    int bc = 1 << 30, bsz = 1 << 30;
    for (int p = 0; p <= 64; p += 16)
        for (int r = 0; r <= 256; r += 16) {
            const int c = lds_read_cycles(base + p, hc * (base + p) + r, tw, s);
            const int sz = p * hc + r;
            if (c < bc || (c == bc && sz < bsz)) { bc = c; bsz = sz; best = LdsPads{p, r}; }
        }
    return best;
}

template <int NP, int KS, int S, int MT, int NT, int TW, int CK>
struct SplitCfg {
    using C = ConvCfg<KS, S, MT, NT, TW, CK>;
    static constexpr int K16 = CK / 16;
    static constexpr LdsPads PADS = best_pads(NP * CK * 2, C::HC, TW, S);
    static constexpr int PSB = NP * CK * 2 + PADS.pixel;   // LDS bytes per pixel: NP pieces x CK 16-bit values (+pad)
    static constexpr int ROWB = C::HC * PSB + PADS.row;    // LDS bytes per haloed row
    static constexpr int A_BYTES = C::HR * ROWB;
    // register-staged weight slab (whole chunk): [tap][piece][k16][kg][NW] 16-byte units
    static constexpr int B_UNITS = C::TAPS * NP * K16 * 2 * C::NW;
    static constexpr int NB = (B_UNITS + 255) / 256;
    static constexpr int LDS_MAIN = A_BYTES + B_UNITS * 16 + 4 * C::NW * 4 + 16;
    // The register-staged kernels run the epilogue's transposes THROUGH the pixel / weight regions (dead between an item's last
    // MFMA block and the next stage's LDS write) when those are big enough: no LDS of its own, one more workgroup per CU for
    // the small tiles.
    static constexpr bool EPI_ALIAS = A_BYTES + B_UNITS * 16 >= EPI_BYTES;
    static constexpr int LDS_BYTES = LDS_MAIN + (EPI_ALIAS ? 0 : EPI_BYTES);
    // LDS-DMA weight rows: [dx][piece][k16][kg][NW] units of one tap row, double-buffered
    static constexpr int SUB_UNITS = C::KW * NP * K16 * 2 * C::NW;
    static constexpr int NBD = (SUB_UNITS + 255) / 256;
    static constexpr int LDS_MAIN_DMA = A_BYTES + 2 * SUB_UNITS * 16 + 4 * C::NW * 4 + 32;
    // DMA kernels: the epilogue's four staging tiles go into the pixel region (as many as fit) and the tap-row buffer the last
    // sub-stage has just consumed (the other one is receiving the next item's first row)
    static constexpr int EPI_IN_A = A_BYTES / EPI_WAVE < 4 ? A_BYTES / EPI_WAVE : 4;
    static constexpr bool EPI_ALIAS_DMA = EPI_IN_A + SUB_UNITS * 16 / EPI_WAVE >= 4;
    static constexpr int LDS_BYTES_DMA = LDS_MAIN_DMA + (EPI_ALIAS_DMA ? 0 : EPI_BYTES);
};

// LDS layout of a pixel's chunk.  bf16x3: [piece][CK channels].  f16x2: the H2 order, [octet][piece][8 channels], so that a
// tensor stored in the H2 format (conv_common.h) is staged by plain 16-byte copies.
template <int NP, int CK>
__device__ __forceinline__ constexpr int frag_off(int k16, int pc) {         // + frag_lane<NP>(lh): this lane's k-half
    return NP == 2 ? k16 * 64 + pc * 16 : pc * (CK * 2) + k16 * 32;
}
template <int NP>
__device__ __forceinline__ constexpr int frag_lane(int lh) { return NP == 2 ? lh * 32 : lh * 16; }

// staging registers of one float4 of float32 activations (channels 4*qq .. 4*qq+3 of the chunk) -> NP pieces in LDS
template <int NP, int CK>
__device__ __forceinline__ void write_pieces(char* pix, int qq, float4 av, float act_scale, float& sat_mx) {
    unsigned short h[4][NP];
    if (NP == 2) {
        av.x *= act_scale; av.y *= act_scale; av.z *= act_scale; av.w *= act_scale;
        sat_track(sat_mx, av.x, av.y);                         // Piece<2>::split clamps at +-65504: keep that observable (conv_common.h)
        sat_track(sat_mx, av.z, av.w);
    }
    Piece<NP>::split(av.x, h[0]);
    Piece<NP>::split(av.y, h[1]);
    Piece<NP>::split(av.z, h[2]);
    Piece<NP>::split(av.w, h[3]);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) {
        uint2 u;
        u.x = (unsigned)h[0][pc] | ((unsigned)h[1][pc] << 16);
        u.y = (unsigned)h[2][pc] | ((unsigned)h[3][pc] << 16);
        char* dst = NP == 2 ? pix + (qq >> 1) * 32 + pc * 16 + (qq & 1) * 8 : pix + qq * 8 + pc * (CK * 2);
        *reinterpret_cast<uint2*>(dst) = u;
    }
}

template <int NP, int KS, int S, int MT, int NT, int TW, int CK>
__device__ __forceinline__ void mma_stage_split(const char* sA, const char* sB, const int (&xoff)[MT], int woff,
                                                f32x16 (&acc)[MT][NT]) {
    using C = ConvCfg<KS, S, MT, NT, TW, CK>;
    using X = SplitCfg<NP, KS, S, MT, NT, TW, CK>;
    using frag = typename Piece<NP>::frag;
    constexpr int STEPS = C::TAPS * X::K16;
    constexpr bool DB = MT * NT <= 2;                  // register double-buffer of the fragments only for small tiles
    frag xf[DB ? 2 : 1][MT][NP], wf[DB ? 2 : 1][NT][NP];
    auto load = [&](int step, int buf) {
        const int tap = step / X::K16, k16 = step % X::K16;
        const int dy = tap / C::KW, dx = tap % C::KW;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc)
                xf[buf][m][pc] = *reinterpret_cast<const frag*>(sA + xoff[m] + dy * X::ROWB + dx * X::PSB + frag_off<NP, CK>(k16, pc));
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc)
                wf[buf][n][pc] = *reinterpret_cast<const frag*>(sB + woff + ((((tap * NP + pc) * X::K16 + k16) * 2) * C::NW + n * 32) * 16);
    };
    if (DB) load(0, 0);
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        const int cb = DB ? (step & 1) : 0;
        if (DB) {
            if (step + 1 < STEPS) load(step + 1, cb ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            load(step, 0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = Piece<NP>::mma(wf[cb][n], xf[cb][m], acc[m][n]);
        if (DB) __builtin_amdgcn_sched_barrier(0);
    }
}

// First generation: pixels AND the weight slab of a channel chunk staged through registers.
template <int NP, int KS, int S, int MT, int NT, int TW, int CK>
__device__ __forceinline__ void conv_split_body(const ConvParams& p) {
    using C = ConvCfg<KS, S, MT, NT, TW, CK>;
    using X = SplitCfg<NP, KS, S, MT, NT, TW, CK>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sA = reinterpret_cast<char*>(smem);          // haloed pixels, NP pieces per channel
    char* sB = sA + X::A_BYTES;                        // weight slab (pre-split)
    float* sS = reinterpret_cast<float*>(sB + X::B_UNITS * 16);
    int* sQ = reinterpret_cast<int*>(sS + 4 * C::NW);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    char* sE = sA + (X::EPI_ALIAS ? 0 : X::LDS_MAIN) + wave * EPI_WAVE;     // this wave's epilogue staging tile
    const int li = lane & 31, lh = lane >> 5;
    const int q = p.n_queues == 8 ? (blockIdx.x & 7) : 0;
    const int n_chunks = p.cin_pad / CK;
    const uint4* wsplit = NP == 2 ? p.wh : p.w3;
    const float* ep_scale = NP == 2 ? p.scale_h : p.scale;

    const int nwg_q = gridDim.x / p.n_queues;
    const int j_cur0 = blockIdx.x / p.n_queues;
    if (j_cur0 >= p.per_queue) return;
    if (tid == 0) sQ[1] = atomicAdd(p.queue + q * QUEUE_STRIDE, 1) + nwg_q;
    int j_cur = j_cur0;

    float4 ra[C::NA];
    unsigned ra_ok = 0;
    uint4 rb[X::NB];
    float rs = 0.f;
    float sat_in = 0.f;                                // max |x * 2^act_shift| of the float32 pixels split while staged (f16x2, float32 input)

    auto issue_loads = [&](const Item& it, int c0) {
        const float* in = p.in + (size_t)it.b * p.H * p.W * p.in_cs + p.in_co + it.g * p.in_gs;
        const uint4* wg = wsplit + (size_t)it.g * (C::TAPS * (p.cin_pad >> 4) * (2 * NP) * p.cout_pad);
        const int iy0 = it.ty * C::TH * S - p.pad_h, ix0 = it.tx * TW * S - p.pad_w;
#pragma unroll
        for (int k = 0; k < C::NA; ++k) {
            // BRANCH-FREE: an out-of-image / out-of-tile lane loads the tensor's first float4 instead and is zeroed when
            // the stage is written to LDS.  With `if (ok) v = load` hipcc waits (vmcnt(0)) inside every conditional block,
            // i.e. the tile arrives as NA serialized HBM round trips BEFORE the MFMA loop instead of underneath it.
            const int idx = tid + k * 256;
            const int idc = idx < C::A_VEC ? idx : 0;
            const int qq = idc % C::QC, pix = idc / C::QC;
            const int hx = pix % C::HC, hy = pix / C::HC;
            const int iy = iy0 + hy, ix = ix0 + hx, c = c0 + qq * 4;
            const bool ok = idx < C::A_VEC && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W &&
                            ((NP == 2 && p.in_h2) ? c0 + (qq >> 1) * 8 : c) < p.cin_valid;     // H2: unit qq = piece (qq & 1) of octet qq >> 1
            ra[k] = ldg4(in + (ok ? (unsigned)((iy * p.W + ix) * p.in_cs + c) : 0u));
            ra_ok = k == 0 ? (ok ? 1u : 0u) : (ra_ok | ((ok ? 1u : 0u) << k));
        }
#pragma unroll
        for (int k = 0; k < X::NB; ++k) {
            const int idx = tid + k * 256;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (idx < X::B_UNITS) {
                // LDS unit index: ((((tap*NP + pc)*K16 + k16)*2 + kg)*NW + j
                int r = idx;
                const int j = r % C::NW; r /= C::NW;
                const int kg = r & 1; r >>= 1;
                const int k16 = r % X::K16; r /= X::K16;
                const int pc = r % NP;
                const int tap = r / NP;
                // global: [tap][cin_pad/16][piece][kg][cout_pad] units
                v = wg[(unsigned)(((((tap * (p.cin_pad >> 4) + (c0 >> 4) + k16) * NP + pc) * 2 + kg) * p.cout_pad) + it.n0 + j)];
            }
            rb[k] = v;
        }
        if (c0 == 0 && tid < 2 * C::NW) {
            const float* src = tid < C::NW ? ep_scale : p.shift;
            rs = src[it.g * p.cout_pad + it.n0 + (tid & (C::NW - 1))];
        }
    };
    auto write_lds = [&](bool first_chunk, int slot) {
#pragma unroll
        for (int k = 0; k < C::NA; ++k) {
            const int idx = tid + k * 256;
            if (idx < C::A_VEC) {
                const int qq = idx % C::QC, pix = idx / C::QC;
                const float4 av = ((ra_ok >> k) & 1u) ? ra[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                char* pixp = sA + (pix / C::HC) * X::ROWB + (pix % C::HC) * X::PSB;
                if (NP == 2 && p.in_h2) *reinterpret_cast<float4*>(pixp + qq * 16) = av;       // already split: a plain copy
                else write_pieces<NP, CK>(pixp, qq, av, p.act_scale, sat_in);
            }
        }
#pragma unroll
        for (int k = 0; k < X::NB; ++k) {
            const int idx = tid + k * 256;
            if (idx < X::B_UNITS) *reinterpret_cast<uint4*>(sB + idx * 16) = rb[k];
        }
        if (first_chunk && tid < 2 * C::NW) sS[slot * 2 * C::NW + tid] = rs;
    };

    int xoff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mb = wave * MT + m;
        const int row = mb * C::RPB + li / TW, col = li % TW;
        xoff[m] = (row * S) * X::ROWB + (col * S) * X::PSB + frag_lane<NP>(lh);
    }
    const int woff = (lh * C::NW + li) * 16;

    int tr_n = 0;
    constexpr int tr_wpw = 4;
    ROMP_TRACE(1);                                     // kernel entry
    Item cur = decode_item(p, q, j_cur, C::NW);
    issue_loads(cur, 0);
    ROMP_TRACE(2);                                     // first loads issued
    write_lds(true, 0);
    ROMP_TRACE(3);                                     // first stage written to LDS (the loads have landed)
    __syncthreads();                                   // stage 0 in LDS; also publishes sQ[1]
    ROMP_TRACE(4);
    int j_next = sQ[1];
    int slot = 0, ch = 0;
    Item nxt = cur;
    bool have_next = j_next < p.per_queue;
    if (have_next) nxt = decode_item(p, q, j_next, C::NW);
    int j_after = 0x7fffffff;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

#pragma unroll 1
    while (true) {
        const bool last = ch + 1 == n_chunks;
        const bool pf = !last || have_next;
        Item tgt = last ? nxt : cur;
        const int c0 = last ? 0 : (ch + 1) * CK;
        if (ch == 0 && tid == 0) j_after = atomicAdd(p.queue + q * QUEUE_STRIDE, 1) + nwg_q;
        if (pf && !(p.dbg & 1)) issue_loads(tgt, c0);
        ROMP_TRACE(10);                                // stage start: next loads issued
        if (p.dbg & 256) __builtin_amdgcn_s_setprio(2);        // experiment: the wave in its MFMA block wins issue arbitration
        if (!(p.dbg & 8)) mma_stage_split<NP, KS, S, MT, NT, TW, CK>(sA, sB, xoff, woff, acc);
        if (p.dbg & 256) __builtin_amdgcn_s_setprio(0);
        ROMP_TRACE(11);                                // MFMA block done
        if (ch == 0 && tid == 0) sQ[0] = j_after;
        __syncthreads();
        ROMP_TRACE(12);                                // barrier: all waves done reading
        if (X::EPI_ALIAS && last) {                    // epilogue first: it stages through the (now dead) pixel / weight regions
            if (!(p.dbg & 4)) conv_epilogue<KS, S, MT, NT, TW, CK>(p, cur, acc, sS + slot * 2 * C::NW, sE, wave, li, lh);
            ROMP_TRACE(14);
            if (!have_next) break;
            __syncthreads();                           // every wave is done with its staging tile
        }
        if (pf && !(p.dbg & 2)) write_lds(last, slot ^ 1);
        ROMP_TRACE(13);                                // next stage written to LDS
        if (last) {
            if (!X::EPI_ALIAS) {
                if (!(p.dbg & 4)) conv_epilogue<KS, S, MT, NT, TW, CK>(p, cur, acc, sS + slot * 2 * C::NW, sE, wave, li, lh);
                ROMP_TRACE(14);                        // epilogue issued
            }
            if (NP == 2 && !p.in_h2) { sat_report(p.sat, sat_in); sat_in = 0.f; }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        }
        if (last && !have_next) break;
        __syncthreads();
        ROMP_TRACE(15);                                // barrier: next stage visible
        if (last) {
            cur = nxt;
            slot ^= 1;
            ch = 0;
            j_next = sQ[0];
            have_next = j_next < p.per_queue;
            if (have_next) nxt = decode_item(p, q, j_next, C::NW);
        } else {
            ++ch;
        }
    }
    if (NP == 2 && !p.in_h2) sat_report(p.sat, sat_in);        // (the paths that leave the loop before the per-item report)
}

// ------------------------------------------------------------------------------------------------
// Second generation ("bxd" / "h2d"): same arithmetic, different data movement.
// Measured on conv_bx3 (64->64 @64x64, B=32): LDS traffic (fragment reads + staging writes) ran at
// ~95 % of the LDS peak at the MFMA rate the kernel was aiming for, and the re-fetch of the pre-split
// weight slab by every 128-pixel workgroup tile drew ~10 TB/s from L2.  Here:
//   * the weight slab is staged one TAP ROW (KW taps) at a time, by LDS-DMA (global_load_lds_dwordx4:
//     no staging VGPRs, no ds_write pass), double-buffered: the DMA of sub-stage n+1 runs under the
//     MFMAs of sub-stage n and is retired (vmcnt(0)) before the barrier that ends sub-stage n;
//   * the 56 VGPRs the weight staging used are gone, so a wave can own a 2x2 block tile (64 pixels x
//     64 channels) at TWO workgroups per CU without spilling: 12 fragment reads per 24 MFMAs instead
//     of 9 per 12, and each weight byte fetched from L2 feeds twice the pixels;
//   * activations are still staged through registers (they must be split into pieces on the way).
// LDS per workgroup (bf16x3, MT=NT=2, TW=16, CK=16): 36.3 KB pixels + 2 x 18.4 KB weight rows = 74 KB.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int NP, int KS, int S, int MT, int NT, int TW, int CK>
__device__ __forceinline__ void conv_splitd_body(const ConvParams& p) {
    using C = ConvCfg<KS, S, MT, NT, TW, CK>;
    using X = SplitCfg<NP, KS, S, MT, NT, TW, CK>;
    using frag = typename Piece<NP>::frag;
    static_assert(X::SUB_UNITS % 64 == 0, "a wave's LDS-DMA writes 64 consecutive units");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sA = reinterpret_cast<char*>(smem);
    char* sB = sA + X::A_BYTES;                                   // two tap-row buffers
    float* sS = reinterpret_cast<float*>(sB + 2 * X::SUB_UNITS * 16);
    int* sQ = reinterpret_cast<int*>(sS + 4 * C::NW);             // [0..1] first two items, [2..3] item-ahead mailbox

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    char* sE = sA + X::LDS_MAIN_DMA + wave * EPI_WAVE;            // this wave's epilogue staging tile (unless aliased, see EPI_ALIAS_DMA)
    const int li = lane & 31, lh = lane >> 5;
    const int q = p.n_queues == 8 ? (blockIdx.x & 7) : 0;
    const int n_chunks = p.cin_pad / CK;
    const int cin16 = p.cin_pad >> 4;
    const uint4* wsplit = NP == 2 ? p.wh : p.w3;
    const float* ep_scale = NP == 2 ? p.scale_h : p.scale;

    const int nwg_q = gridDim.x / p.n_queues;
    const int j_cur0 = blockIdx.x / p.n_queues;
    if (j_cur0 >= p.per_queue) return;
    if (tid == 0) sQ[1] = atomicAdd(p.queue + q * QUEUE_STRIDE, 1) + nwg_q;
    int j_cur = j_cur0;

    float4 ra[C::NA];
    unsigned ra_ok = 0;
    float rs = 0.f;
    float sat_in = 0.f;                                // (see conv_split_body)

    // LDS-DMA of the weight units of tap row `row`, channel chunk c0, into buffer `buf`
    auto issue_B = [&](const Item& it, int c0, int row, int buf) {
        const uint4* wg = wsplit + (size_t)it.g * (C::TAPS * cin16 * (2 * NP) * p.cout_pad);
        char* dst = sB + buf * (X::SUB_UNITS * 16);
#pragma unroll
        for (int k = 0; k < X::NBD; ++k) {
            if (k * 256 + wave * 64 < X::SUB_UNITS) {             // wave-uniform
                int r = k * 256 + tid;
                const int j = r % C::NW; r /= C::NW;
                const int kg = r & 1; r >>= 1;
                const int k16 = r % X::K16; r /= X::K16;
                const int pc = r % NP;
                const int dx = r / NP;
                const int tap = row * C::KW + dx;
                const uint4* src = wg + (unsigned)(((((tap * cin16 + (c0 >> 4) + k16) * NP + pc) * 2 + kg) * p.cout_pad) + it.n0 + j);
                __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(dst + (k * 256 + wave * 64) * 16), 16, 0, 0);
            }
        }
    };
    auto issue_A = [&](const Item& it, int c0) {
        const float* in = p.in + (size_t)it.b * p.H * p.W * p.in_cs + p.in_co + it.g * p.in_gs;
        const int iy0 = it.ty * C::TH * S - p.pad_h, ix0 = it.tx * TW * S - p.pad_w;
#pragma unroll
        for (int k = 0; k < C::NA; ++k) {
            // BRANCH-FREE (see conv_split_body)
            const int idx = tid + k * 256;
            const int idc = idx < C::A_VEC ? idx : 0;
            const int qq = idc % C::QC, pix = idc / C::QC;
            const int hx = pix % C::HC, hy = pix / C::HC;
            const int iy = iy0 + hy, ix = ix0 + hx, c = c0 + qq * 4;
            const bool ok = idx < C::A_VEC && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W &&
                            ((NP == 2 && p.in_h2) ? c0 + (qq >> 1) * 8 : c) < p.cin_valid;     // H2: unit qq = piece (qq & 1) of octet qq >> 1
            ra[k] = ldg4(in + (ok ? (unsigned)((iy * p.W + ix) * p.in_cs + c) : 0u));
            ra_ok = k == 0 ? (ok ? 1u : 0u) : (ra_ok | ((ok ? 1u : 0u) << k));
        }
        if (c0 == 0 && tid < 2 * C::NW) {
            const float* src = tid < C::NW ? ep_scale : p.shift;
            rs = src[it.g * p.cout_pad + it.n0 + (tid & (C::NW - 1))];
        }
    };
    auto write_A = [&](bool first_chunk, int slot) {
#pragma unroll
        for (int k = 0; k < C::NA; ++k) {
            const int idx = tid + k * 256;
            if (idx < C::A_VEC) {
                const int qq = idx % C::QC, pix = idx / C::QC;
                const float4 av = ((ra_ok >> k) & 1u) ? ra[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                char* pixp = sA + (pix / C::HC) * X::ROWB + (pix % C::HC) * X::PSB;
                if (NP == 2 && p.in_h2) *reinterpret_cast<float4*>(pixp + qq * 16) = av;       // already split: a plain copy
                else write_pieces<NP, CK>(pixp, qq, av, p.act_scale, sat_in);
            }
        }
        if (first_chunk && tid < 2 * C::NW) sS[slot * 2 * C::NW + tid] = rs;
    };

    int xoff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mb = wave * MT + m;
        const int row = mb * C::RPB + li / TW, col = li % TW;
        xoff[m] = (row * S) * X::ROWB + (col * S) * X::PSB + frag_lane<NP>(lh);
    }
    const int woff = (lh * C::NW + li) * 16;

    int tr_n = 0;
    constexpr int tr_wpw = 4;
    ROMP_TRACE(1);
    Item cur = decode_item(p, q, j_cur, C::NW);
    issue_B(cur, 0, 0, 0);
    issue_A(cur, 0);
    ROMP_TRACE(2);
    write_A(true, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ROMP_TRACE(3);
    __syncthreads();                                   // stage 0 in LDS; also publishes sQ[1]
    ROMP_TRACE(4);
    int j_next = sQ[1];
    int slot = 0, ch = 0, row = 0, bbuf = 0, par = 0;
    Item nxt = cur;
    bool have_next = j_next < p.per_queue;
    if (have_next) nxt = decode_item(p, q, j_next, C::NW);
    int j_after = 0x7fffffff;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

#pragma unroll 1
    while (true) {
        const bool last_row = row + 1 == C::KH;
        const bool last_ch = ch + 1 == n_chunks;
        const bool last = last_row && last_ch;                    // last sub-stage of the item
        // what the NEXT sub-stage needs
        const bool pfB = !last || have_next;
        const Item tgtB = last ? nxt : cur;
        const int c0B = last_row ? (last_ch ? 0 : (ch + 1) * CK) : ch * CK;
        const int rowB = last_row ? 0 : row + 1;
        if (ch == 0 && row == 0 && tid == 0) j_after = atomicAdd(p.queue + q * QUEUE_STRIDE, 1) + nwg_q;
        if (pfB && !(p.dbg & (1 | 128))) issue_B(tgtB, c0B, rowB, bbuf ^ 1);
        const bool pfA = last_row && pfB;                          // next chunk's pixels: loaded under the last tap row
        if (pfA && !(p.dbg & (1 | 64))) issue_A(tgtB, c0B);
        __builtin_amdgcn_sched_barrier(0);               // keep every DMA / load issue ABOVE the MFMA block (hipcc sank 3 of the 5 DMAs below it)
        ROMP_TRACE(10);
        if (p.dbg & 256) __builtin_amdgcn_s_setprio(2);
        if (!(p.dbg & 8)) {
            const char* sBc = sB + bbuf * (X::SUB_UNITS * 16);
#pragma unroll
            for (int dx = 0; dx < C::KW; ++dx)
#pragma unroll
                for (int k16 = 0; k16 < X::K16; ++k16) {
                    frag xf[MT][NP], wf[NT][NP];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int pc = 0; pc < NP; ++pc)
                            xf[m][pc] = *reinterpret_cast<const frag*>(sA + xoff[m] + row * X::ROWB + dx * X::PSB + frag_off<NP, CK>(k16, pc));
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int pc = 0; pc < NP; ++pc)
                            wf[n][pc] = *reinterpret_cast<const frag*>(sBc + woff + ((((dx * NP + pc) * X::K16 + k16) * 2) * C::NW + n * 32) * 16);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n) acc[m][n] = Piece<NP>::mma(wf[n], xf[m], acc[m][n]);
                }
        }
        if (p.dbg & 256) __builtin_amdgcn_s_setprio(0);
        ROMP_TRACE(11);
        if (ch == 0 && row == 0 && tid == 0) sQ[2 + par] = j_after;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's LDS-DMA has landed (and ra is in)
        ROMP_TRACE(16);                                            // DMA / loads landed
        __syncthreads();                                           // all waves: done reading bbuf / sA, DMA visible
        ROMP_TRACE(12);
        bbuf ^= 1;
        if (last_row) {
            if (X::EPI_ALIAS_DMA && last_ch) {                     // epilogue first: it stages through the pixel region and the consumed row buffer
                char* se = wave < X::EPI_IN_A ? sA + wave * EPI_WAVE
                                              : sB + (bbuf ^ 1) * (X::SUB_UNITS * 16) + (wave - X::EPI_IN_A) * EPI_WAVE;
                if (!(p.dbg & 4)) conv_epilogue<KS, S, MT, NT, TW, CK>(p, cur, acc, sS + slot * 2 * C::NW, se, wave, li, lh);
                ROMP_TRACE(14);
                if (!have_next) break;
                __syncthreads();                                   // every wave is done with its staging tile
            }
            if (pfA && !(p.dbg & 2)) write_A(last_ch, slot ^ 1);
            ROMP_TRACE(13);
            if (last_ch) {
                if (!X::EPI_ALIAS_DMA) {
                    if (!(p.dbg & 4)) conv_epilogue<KS, S, MT, NT, TW, CK>(p, cur, acc, sS + slot * 2 * C::NW, sE, wave, li, lh);
                    ROMP_TRACE(14);
                }
                if (NP == 2 && !p.in_h2) { sat_report(p.sat, sat_in); sat_in = 0.f; }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
                if (!have_next) break;
            }
            __syncthreads();                                       // next chunk's pixels visible
            ROMP_TRACE(15);
            row = 0;
            if (last_ch) {
                cur = nxt;
                slot ^= 1;
                ch = 0;
                j_next = sQ[2 + par];
                par ^= 1;
                have_next = j_next < p.per_queue;
                if (have_next) nxt = decode_item(p, q, j_next, C::NW);
            } else {
                ++ch;
            }
        } else {
            ++row;
        }
    }
    if (NP == 2 && !p.in_h2) sat_report(p.sat, sat_in);
}

}  // namespace romp
