// ATTIC (round 3): measured negative result, no longer part of libromp_hip.so (VERDICT r2 #8e).  To revive: copy next to
// romp_amd/csrc/conv_split.h, add to build.py SOURCES and to collect_variants() in conv_mfma.hip (math codes 5 / 6 / 7).
// conv_h2p.hip -- 3x3 stride-1 convolution on the f16x2 split (conv_split.h), third generation: a software pipeline fed
// ENTIRELY by LDS-DMA.
//
// What the phase traces of the first two generations showed (scripts/conv_trace.py, B=32): the MFMA block is 17-38 % of a
// wave's time; the rest is staging -- global loads issued, waited for, split / copied through VGPRs into LDS, two barriers
// per channel chunk -- and an epilogue, all serialized with the matrix work because every wave of the chip is in the same
// phase at the same time.  With the activations stored pre-split (H2 format, conv_common.h) nothing has to pass through
// registers any more:
//   * a stage = one 16-channel chunk of one work item: the haloed pixel tile (A) and the chunk's 9-tap weight slab (B) are
//     fetched by global_load_lds_dwordx4 straight into one of two stage buffers.  The DMA of stage s+1 is issued in PIECES
//     BETWEEN THE TAPS of stage s's MFMA block (one piece = one wave-instruction = 1 KiB; its address arithmetic is ~10 VALU
//     on per-lane descriptors computed once per kernel), so it costs no phase of its own and has the whole block to land;
//   * ONE barrier per stage (raw s_barrier after s_waitcnt vmcnt(0): the wave's pieces of the next stage have landed, and
//     every wave is done reading the buffer the stage after next will overwrite);
//   * no staging VGPRs, no split VALU, no ds_write pass: between two barriers a wave's instruction stream is ds_read + MFMA
//     + a handful of DMA issues;
//   * 512-thread workgroups, one per CU: two waves per SIMD share the matrix pipe, so one wave's LDS-read latency / DMA
//     issue / epilogue is the other's MFMA time; a workgroup tile is 8 x MT x 32 pixels x NT x 32 channels;
//   * WRES variants (layers whose whole split weight set is <= 36 KiB, i.e. 32 -> 32 channels): the weights are fetched once
//     per workgroup and stay in LDS, only pixels stream (these layers sit at the HBM ridge: the weight slab was 40 % of the
//     bytes a CU pulled per work item);
//   * LDS pixel layout: a haloed row is a run of 4-pixel column groups of 16 units (16 B each: [octet][piece] of the
//     chunk); inside a group the unit of (column c, unit w) sits at (c & 3) + 4 * ((w + (c >> 2)) & 3): the DMA needs a
//     dense destination (lane i writes unit i), and with this rotation the 16-lane groups of the fragment ds_read_b128
//     still hit 16 distinct bank slots for every tap;
//   * residual loads of an item are issued before its last MFMA block; the epilogue (conv_common.h) transposes through
//     per-wave staging tiles of its own.
// Work items are uniform in cost and a workgroup takes 2-4 of them: a static stride over the XCD's items replaces the atomic
// queue (whose returning atomic put an s_waitcnt vmcnt(0) into the stage loop of the earlier generations).
#include "conv_split.h"

namespace romp {

template <int MT, int NT, int TW, bool WRES>
struct PipeCfg {
    static constexpr int NWV = 8, CK = 16;
    using C = ConvCfg<3, 1, MT, NT, TW, CK, NWV>;
    static constexpr int CG = (C::HC + 3) / 4;                  // 4-pixel column groups per haloed row
    static constexpr int RSU = CG * 16;                         // 16-byte units per haloed row
    static constexpr int NA_I = (C::HR * RSU + 63) / 64;        // DMA wave-instructions (1 KiB each) of the pixel tile
    static constexpr int NB_I = 9 * 2 * 2 * C::NW / 64;         // ... of one chunk's weight slab [tap][piece][k-half][NW]
    static constexpr int RES_CHUNKS = 2;                        // WRES: Cin = 32
    static constexpr int STAGE_I = WRES ? NA_I : NA_I + NB_I;
    static constexpr int NI = (STAGE_I + NWV - 1) / NWV;        // pieces per wave (piece k of wave w = instruction k * 8 + w)
    static constexpr int STAGE_BYTES = STAGE_I * 1024;
    static constexpr int WRES_BYTES = WRES ? RES_CHUNKS * NB_I * 1024 : 0;
    static constexpr int NRES = (RES_CHUNKS * NB_I + NWV - 1) / NWV;
    static constexpr int SS_DMA_BYTES = NWV * 256;              // every wave DMAs one dword per lane per stage: scale | shift | filler
    static constexpr int OFF_W = 2 * STAGE_BYTES;               // resident weights
    static constexpr int OFF_E = OFF_W + WRES_BYTES;            // epilogue staging tiles ...
    // ... unless they do not fit beside the two stage buffers (512-pixel x 64-channel tiles): then the epilogue stages through
    // the stage buffer its item has just finished with (one extra barrier per item)
    static constexpr bool EALIAS = OFF_E + NWV * EPI_WAVE + 2 * SS_DMA_BYTES + 16 > 160 * 1024;
    static constexpr int OFF_S = OFF_E + (EALIAS ? 0 : NWV * EPI_WAVE);        // two scale | shift slots
    static constexpr int LDS_BYTES = OFF_S + 2 * SS_DMA_BYTES + 16;
    static_assert(!EALIAS || STAGE_BYTES >= NWV * EPI_WAVE, "aliased epilogue staging must fit a stage buffer");
    static_assert((9 * 2 * 2 * C::NW) % 64 == 0, "weight slab must be whole DMA instructions");
    static_assert(2 * C::NW <= NWV * 64, "scale | shift fit the per-stage dword DMA");
    static_assert(NI <= 12, "one DMA piece per tap, the rest after the block");
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ void wait_all_and_barrier() {    // raw barrier: this wave's DMA landed and LDS traffic retired, then everybody's
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
// The same, but the N youngest VMEM operations (the epilogue's output stores, issued after the DMA pieces) may stay in flight:
// memory operations retire in order, so everything older -- the DMA -- has landed.
template <int N>
__device__ __forceinline__ void wait_but_and_barrier() {
    static_assert(N == 4 || N == 8 || N == 16, "store count of an epilogue");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

struct StageDesc {              // wave-uniform description of one stage's sources
    const float* in;            // image + group + chunk base of the pixel tensor
    const uint4* wg;            // group + chunk + channel-slice base of the split weights
    const float* sc; const float* sh;   // scale / shift of the item's channel slice
    int iy0, ix0, pix0;         // tile origin (may be negative: zero padding) and its float offset
    int c0;
};

template <int MT, int NT, int TW, bool WRES>
__global__ __launch_bounds__(512, 2) void conv_h2p_kernel(ConvParams p) {
    if (p.dbg & 32) return;                            // ablation: launch cost only
    using X = PipeCfg<MT, NT, TW, WRES>;
    using C = typename X::C;
    using frag = f16x8;
    constexpr int CK = X::CK, NWV = X::NWV;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sBuf = reinterpret_cast<char*>(smem);                       // two stage buffers: [pixel tile | weight slab]
    float* sS = reinterpret_cast<float*>(sBuf + X::OFF_S);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int q = p.n_queues == 8 ? (blockIdx.x & 7) : 0;
    const int n_chunks = p.cin_pad / CK;
    const int cin16 = p.cin_pad >> 4;
    char* sE = sBuf + X::OFF_E + wave * EPI_WAVE;

    const int nwg_q = gridDim.x / p.n_queues;
    const int j_cur0 = blockIdx.x / p.n_queues;
    if (j_cur0 >= p.per_queue) return;

    // ---- per-lane DMA descriptors of this wave's pieces (the same for every stage): source offset relative to the stage base,
    // and for pixel pieces the (row, col, octet) of the unit for the zero-padding test
    int d_off[X::NI], d_rc[X::NI];
#pragma unroll
    for (int k = 0; k < X::NI; ++k) {
        const int i = k * NWV + wave;
        if (i < X::NA_I) {
            const int U = i * 64 + lane;
            const int row = U / X::RSU, r = U % X::RSU;
            const int cg = r >> 4, r16 = r & 15;
            const int col = cg * 4 + (r16 & 3), w = ((r16 >> 2) - cg) & 3;      // unit w of the chunk = piece (w & 1) of octet (w >> 1)
            d_off[k] = (row * p.W + col) * p.in_cs + w * 4;
            d_rc[k] = row | (col << 8) | ((row < C::HR && col < C::HC) ? 1 << 16 : 0) | ((w >> 1) << 17);
        } else {
            int r = (i - X::NA_I) * 64 + lane;                                   // [tap][piece][k-half][NW]
            const int j = r % C::NW; r /= C::NW;
            const int kg = r & 1; r >>= 1;
            const int pc = r & 1;
            const int tap = r >> 1;
            d_off[k] = (tap * cin16 * 4 + pc * 2 + kg) * p.cout_pad + j;
            d_rc[k] = 0;
        }
    }

    auto make_desc = [&](const Item& it, int c0) {
        StageDesc d;
        d.in = p.in + (size_t)it.b * p.H * p.W * p.in_cs + p.in_co + it.g * p.in_gs + c0;
        d.wg = p.wh + (size_t)it.g * (9 * cin16 * 4 * p.cout_pad) + (c0 >> 4) * 4 * p.cout_pad + it.n0;
        d.sc = p.scale_h + it.g * p.cout_pad + it.n0;
        d.sh = p.shift + it.g * p.cout_pad + it.n0;
        d.iy0 = it.ty * C::TH - p.pad_h;
        d.ix0 = it.tx * TW - p.pad_w;
        d.pix0 = (d.iy0 * p.W + d.ix0) * p.in_cs;
        d.c0 = c0;
        return d;
    };
    // piece k of a stage -> stage buffer `buf`
    auto issue_piece = [&](int k, const StageDesc& d, int buf) {
        const int i = k * NWV + wave;                                  // wave-uniform
        if (i >= X::STAGE_I) return;
        const void* src;
        if (i < X::NA_I) {
            const int row = d_rc[k] & 255, col = (d_rc[k] >> 8) & 255;
            const int iy = d.iy0 + row, ix = d.ix0 + col;
            const bool ok = ((d_rc[k] >> 16) & 1) && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W &&
                            d.c0 + ((d_rc[k] >> 17) & 1) * 8 < p.cin_valid;
            src = ok ? (const void*)(d.in + (d.pix0 + d_off[k])) : (const void*)p.zero;
        } else {
            src = d.wg + (unsigned)d_off[k];
        }
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(sBuf + buf * X::STAGE_BYTES + i * 1024), 16, 0, 0);
    };
    // scale | shift of the stage's item: 2*NW floats, one dword per lane, waves 0 .. 2*NW/64 - 1 (others: nothing)
    auto issue_ss = [&](const StageDesc& d, int slot) {
        if (wave * 64 >= 2 * C::NW) return;
        const int e = wave * 64 + lane;
        const float* src = e < C::NW ? d.sc + e : d.sh + (e - C::NW);
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(reinterpret_cast<char*>(sS) + slot * X::SS_DMA_BYTES + wave * 256), 4, 0, 0);
    };

    // ---- fragment addresses: pixel (row, col + dx), unit w = 2 * lh + piece
    int xaddr[MT][3][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mb = wave * MT + m;
        const int prow = mb * C::RPB + li / TW, pcol = li % TW;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const int col = pcol + dx, w = lh * 2 + pc;
                xaddr[m][dx][pc] = (prow * X::RSU + (col >> 2) * 16 + (col & 3) + 4 * ((w + (col >> 2)) & 3)) * 16;
            }
    }
    const int woff = (lh * C::NW + li) * 16;

    int tr_n = 0;
    constexpr int tr_wpw = NWV;
    ROMP_TRACE(1);
    Item cur = decode_item(p, q, j_cur0, C::NW);
    {
        const StageDesc d0 = make_desc(cur, 0);
#pragma unroll
        for (int k = 0; k < X::NI; ++k) issue_piece(k, d0, 0);
        issue_ss(d0, 0);
        if (WRES) {                                                     // the layer's whole weight set, once
#pragma unroll
            for (int k = 0; k < X::NRES; ++k) {
                const int i = k * NWV + wave;
                if (i < X::RES_CHUNKS * X::NB_I) {
                    int r = (i % X::NB_I) * 64 + lane;
                    const int j = r % C::NW; r /= C::NW;
                    const int kg = r & 1; r >>= 1;
                    const int pc = r & 1;
                    const int tap = r >> 1;
                    const uint4* src = d0.wg + (unsigned)((((tap * cin16 + i / X::NB_I) * 2 + pc) * 2 + kg) * p.cout_pad + j);
                    __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(sBuf + X::OFF_W + i * 1024), 16, 0, 0);
                }
            }
        }
    }
    ROMP_TRACE(2);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    int j_next = j_cur0 + nwg_q;
    Item nxt = cur;
    bool have_next = j_next < p.per_queue;
    if (have_next) nxt = decode_item(p, q, j_next, C::NW);
    int ch = 0, buf = 0, slot = 0;
    EpiRes<MT, NT> pre;
    const bool use_pre = MT * NT < 4 && p.res && p.vec_io;             // (a 2x2 tile's residual would be 64 VGPRs: loaded in the epilogue)
    const bool full_tiles = p.Ho % C::TH == 0 && p.vec_io;              // every output store of an epilogue is issued: its count is known
    wait_all_and_barrier();
    ROMP_TRACE(4);

#pragma unroll 1
    while (true) {
        const bool last = ch + 1 == n_chunks;
        const bool has_nx = (!last || have_next) && !(p.dbg & 1);       // is there a next stage to fetch?
        const StageDesc nd = make_desc(last ? nxt : cur, last ? 0 : (ch + 1) * CK);
        const int nslot = last ? slot ^ 1 : slot;
        if (ch == 0 && use_pre) conv_epilogue_prefetch<3, 1, MT, NT, TW, CK, NWV>(p, cur, wave, lane, pre);   // an item ahead of its use
        ROMP_TRACE(10);
        {
            const char* sA = sBuf + buf * X::STAGE_BYTES;
            const char* sB = WRES ? sBuf + X::OFF_W + ch * (X::NB_I * 1024) : sA + X::NA_I * 1024;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap % 3;
                if (!(p.dbg & 8)) {
                    frag xf[MT][2], wf[NT][2];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int pc = 0; pc < 2; ++pc)
                            xf[m][pc] = *reinterpret_cast<const frag*>(sA + xaddr[m][dx][pc] + dy * (X::RSU * 16));
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int pc = 0; pc < 2; ++pc)
                            wf[n][pc] = *reinterpret_cast<const frag*>(sB + woff + (((tap * 2 + pc) * 2) * C::NW + n * 32) * 16);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n) acc[m][n] = Piece<2>::mma(wf[n], xf[m], acc[m][n]);
                }
                if (tap < X::NI && has_nx) issue_piece(tap, nd, buf ^ 1);   // the next stage's DMA, one piece per tap
                __builtin_amdgcn_sched_barrier(0);
            }
            if (has_nx) {
#pragma unroll
                for (int k = 9; k < X::NI; ++k) issue_piece(k, nd, buf ^ 1);
                issue_ss(nd, nslot);
            }
        }
        ROMP_TRACE(11);
        if (last) {
            char* stage_e = sE;
            if (X::EALIAS) {                                           // every wave is done reading this stage buffer: stage through it
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                stage_e = sBuf + buf * X::STAGE_BYTES + wave * EPI_WAVE;
            }
            if (!(p.dbg & 4))
                conv_epilogue<3, 1, MT, NT, TW, CK, NWV>(p, cur, acc, sS + slot * (X::SS_DMA_BYTES / 4), stage_e, wave, li, lh, pre, use_pre);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
            ROMP_TRACE(14);
            if (!have_next) break;
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
        // the next stage has landed; everybody is done with the buffer the stage after it overwrites.  Right after an epilogue the
        // item's output stores stay in flight (they drain under the next item's first MFMA blocks).
        if (last && full_tiles) wait_but_and_barrier<MT * NT * 4>(); else wait_all_and_barrier();
        ROMP_TRACE(12);
    }
}

// ------------------------------------------------------------------------------------------------
// Producer / consumer form of the same pipeline ("h2q").  In conv_h2p_kernel a wave's DMA pieces sit in its own in-order
// instruction stream: each global_load_lds costs it 150-250 issue cycles (measured: a 54-MFMA block of 3 456 pipe cycles took
// 5 200 with its eight pieces inside), during which it issues no MFMA.  Here the workgroup is 4 CONSUMER waves (one per SIMD,
// MT x NT blocks each, nothing but ds_read + MFMA + their item's epilogue) and 4 LOADER waves (one per SIMD, the second wave
// slot: they issue every DMA piece of the next stage, wait for them, and meet the consumers at the stage barrier).  Same
// two stage buffers, same one barrier per stage; the consumers' epilogue overlaps the loaders' next fetch for free.
template <int MT, int NT, int TW>
struct QCfg {
    static constexpr int NCW = 4, NLW = 4, CK = 16;             // consumer / loader waves
    using C = ConvCfg<3, 1, MT, NT, TW, CK, NCW>;
    static constexpr int CG = (C::HC + 3) / 4;
    static constexpr int RSU = CG * 16;
    static constexpr int NA_I = (C::HR * RSU + 63) / 64;
    static constexpr int NB_I = 9 * 2 * 2 * C::NW / 64;
    static constexpr int STAGE_I = NA_I + NB_I;
    static constexpr int NI = (STAGE_I + NLW - 1) / NLW;        // pieces per loader wave
    static constexpr int STAGE_BYTES = STAGE_I * 1024;
    static constexpr int SS_DMA_BYTES = 512;                    // scale | shift: 2 x 64 dwords, loader waves 0 and 1
    static constexpr int OFF_E = 2 * STAGE_BYTES;
    static constexpr int OFF_S = OFF_E + NCW * EPI_WAVE;
    static constexpr int LDS_BYTES = OFF_S + 2 * SS_DMA_BYTES + 16;
    static_assert(C::NW <= 64, "scale | shift DMA: one loader wave each");
};

template <int MT, int NT, int TW>
__global__ __launch_bounds__(512, 2) void conv_h2q_kernel(ConvParams p) {
    if (p.dbg & 32) return;
    using X = QCfg<MT, NT, TW>;
    using C = typename X::C;
    using frag = f16x8;
    constexpr int CK = X::CK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sBuf = reinterpret_cast<char*>(smem);
    float* sS = reinterpret_cast<float*>(sBuf + X::OFF_S);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= X::NCW;
    const int lw = wave - X::NCW;                                      // loader index
    const int li = lane & 31, lh = lane >> 5;
    const int q = p.n_queues == 8 ? (blockIdx.x & 7) : 0;
    const int n_chunks = p.cin_pad / CK;
    const int cin16 = p.cin_pad >> 4;
    const int nwg_q = gridDim.x / p.n_queues;
    const int j_cur0 = blockIdx.x / p.n_queues;
    if (j_cur0 >= p.per_queue) return;

    auto make_desc = [&](const Item& it, int c0) {
        StageDesc d;
        d.in = p.in + (size_t)it.b * p.H * p.W * p.in_cs + p.in_co + it.g * p.in_gs + c0;
        d.wg = p.wh + (size_t)it.g * (9 * cin16 * 4 * p.cout_pad) + (c0 >> 4) * 4 * p.cout_pad + it.n0;
        d.sc = p.scale_h + it.g * p.cout_pad + it.n0;
        d.sh = p.shift + it.g * p.cout_pad + it.n0;
        d.iy0 = it.ty * C::TH - p.pad_h;
        d.ix0 = it.tx * TW - p.pad_w;
        d.pix0 = (d.iy0 * p.W + d.ix0) * p.in_cs;
        d.c0 = c0;
        return d;
    };
    int tr_n = 0;
    constexpr int tr_wpw = 8;
    ROMP_TRACE(1);
    Item cur = decode_item(p, q, j_cur0, C::NW);
    int j_next = j_cur0 + nwg_q;
    Item nxt = cur;
    bool have_next = j_next < p.per_queue;
    if (have_next) nxt = decode_item(p, q, j_next, C::NW);
    int ch = 0, buf = 0, slot = 0;

    if (loader) {
        // ================================================================ loader waves
        int d_off[X::NI], d_rc[X::NI];
#pragma unroll
        for (int k = 0; k < X::NI; ++k) {
            const int i = k * X::NLW + lw;
            if (i < X::NA_I) {
                const int U = i * 64 + lane;
                const int row = U / X::RSU, r = U % X::RSU;
                const int cg = r >> 4, r16 = r & 15;
                const int col = cg * 4 + (r16 & 3), w = ((r16 >> 2) - cg) & 3;
                d_off[k] = (row * p.W + col) * p.in_cs + w * 4;
                d_rc[k] = row | (col << 8) | ((row < C::HR && col < C::HC) ? 1 << 16 : 0) | ((w >> 1) << 17);
            } else {
                int r = (i - X::NA_I) * 64 + lane;
                const int j = r % C::NW; r /= C::NW;
                const int kg = r & 1; r >>= 1;
                const int pc = r & 1;
                const int tap = r >> 1;
                d_off[k] = (tap * cin16 * 4 + pc * 2 + kg) * p.cout_pad + j;
                d_rc[k] = 0;
            }
        }
        auto issue_stage = [&](const StageDesc& d, int b, int sl) {
#pragma unroll
            for (int k = 0; k < X::NI; ++k) {
                const int i = k * X::NLW + lw;
                if (i < X::STAGE_I) {
                    const void* src;
                    if (i < X::NA_I) {
                        const int row = d_rc[k] & 255, col = (d_rc[k] >> 8) & 255;
                        const int iy = d.iy0 + row, ix = d.ix0 + col;
                        const bool ok = ((d_rc[k] >> 16) & 1) && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W &&
                                        d.c0 + ((d_rc[k] >> 17) & 1) * 8 < p.cin_valid;
                        src = ok ? (const void*)(d.in + (d.pix0 + d_off[k])) : (const void*)p.zero;
                    } else {
                        src = d.wg + (unsigned)d_off[k];
                    }
                    __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(sBuf + b * X::STAGE_BYTES + i * 1024), 16, 0, 0);
                }
            }
            if (lw < 2 && lane < C::NW) {                                 // scale[NW] (loader 0) | shift[NW] (loader 1) of the item's slice
                const float* src = (lw == 0 ? d.sc : d.sh) + lane;
                __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(reinterpret_cast<char*>(sS) + sl * X::SS_DMA_BYTES + lw * (C::NW * 4)), 4, 0, 0);
            }
        };
        if (!(p.dbg & 1)) issue_stage(make_desc(cur, 0), 0, 0);
        wait_all_and_barrier();                                          // stage 0 landed
#pragma unroll 1
        while (true) {
            const bool last = ch + 1 == n_chunks;
            const bool has_nx = !last || have_next;
            if (has_nx && !(p.dbg & 1)) issue_stage(make_desc(last ? nxt : cur, last ? 0 : (ch + 1) * CK), buf ^ 1, last ? slot ^ 1 : slot);
            ROMP_TRACE(13);
            if (last) {
                if (!have_next) break;
                cur = nxt; slot ^= 1; ch = 0;
                j_next += nwg_q;
                have_next = j_next < p.per_queue;
                if (have_next) nxt = decode_item(p, q, j_next, C::NW);
            } else {
                ++ch;
            }
            buf ^= 1;
            wait_all_and_barrier();                                      // the stage just issued has landed; consumers are done with the other buffer
            ROMP_TRACE(12);
        }
        return;
    }

    // ==================================================================== consumer waves
    char* sE = sBuf + X::OFF_E + wave * EPI_WAVE;
    int xaddr[MT][3][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mb = wave * MT + m;
        const int prow = mb * C::RPB + li / TW, pcol = li % TW;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const int col = pcol + dx, w = lh * 2 + pc;
                xaddr[m][dx][pc] = (prow * X::RSU + (col >> 2) * 16 + (col & 3) + 4 * ((w + (col >> 2)) & 3)) * 16;
            }
    }
    const int woff = X::NA_I * 1024 + (lh * C::NW + li) * 16;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    EpiRes<MT, NT> pre;
    const bool use_pre = MT * NT < 4 && p.res && p.vec_io;               // (64 VGPRs for a 2x2 tile: those load their residual in the epilogue)
    wait_all_and_barrier();                                              // stage 0 landed
    ROMP_TRACE(4);
#pragma unroll 1
    while (true) {
        const bool last = ch + 1 == n_chunks;
        if (ch == 0 && use_pre) conv_epilogue_prefetch<3, 1, MT, NT, TW, CK, X::NCW>(p, cur, wave, lane, pre);
        ROMP_TRACE(10);
        if (!(p.dbg & 8)) {
            const char* sA = sBuf + buf * X::STAGE_BYTES;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap % 3;
                frag xf[MT][2], wf[NT][2];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc)
                        xf[m][pc] = *reinterpret_cast<const frag*>(sA + xaddr[m][dx][pc] + dy * (X::RSU * 16));
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc)
                        wf[n][pc] = *reinterpret_cast<const frag*>(sA + woff + (((tap * 2 + pc) * 2) * C::NW + n * 32) * 16);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = Piece<2>::mma(wf[n], xf[m], acc[m][n]);
            }
        }
        ROMP_TRACE(11);
        if (last) {
            if (!(p.dbg & 4))
                conv_epilogue<3, 1, MT, NT, TW, CK, X::NCW>(p, cur, acc, sS + slot * (X::SS_DMA_BYTES / 4), sE, wave, li, lh, pre, use_pre);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
            ROMP_TRACE(14);
            if (!have_next) break;
            cur = nxt; slot ^= 1; ch = 0;
            j_next += nwg_q;
            have_next = j_next < p.per_queue;
            if (have_next) nxt = decode_item(p, q, j_next, C::NW);
        } else {
            ++ch;
        }
        buf ^= 1;
        wait_all_and_barrier();
        ROMP_TRACE(12);
    }
}

#define ROMP_CONV_VARIANT_H2Q(MT, NT, TW)                                                                \
    { 3, 1, MT, NT, TW, 16, conv_h2q_kernel<MT, NT, TW>, QCfg<MT, NT, TW>::LDS_BYTES,                    \
      QCfg<MT, NT, TW>::C::TH, 0, 0, 7, 512 }

#define ROMP_CONV_VARIANT_H2P(MT, NT, TW, WRES)                                                          \
    { 3, 1, MT, NT, TW, 16, conv_h2p_kernel<MT, NT, TW, WRES>, PipeCfg<MT, NT, TW, WRES>::LDS_BYTES,     \
      PipeCfg<MT, NT, TW, WRES>::C::TH, 0, 0, WRES ? 6 : 5, 512 }

static ConvVariant kVariantsH2p[] = {
    ROMP_CONV_VARIANT_H2P(1, 2, 16, false), ROMP_CONV_VARIANT_H2P(1, 2, 32, false), ROMP_CONV_VARIANT_H2P(2, 1, 32, false),
    ROMP_CONV_VARIANT_H2P(2, 1, 16, false), ROMP_CONV_VARIANT_H2P(1, 1, 16, false), ROMP_CONV_VARIANT_H2P(1, 1, 32, false),
    ROMP_CONV_VARIANT_H2P(2, 2, 32, false), ROMP_CONV_VARIANT_H2P(2, 2, 16, false),
    ROMP_CONV_VARIANT_H2P(2, 1, 32, true), ROMP_CONV_VARIANT_H2P(2, 1, 16, true), ROMP_CONV_VARIANT_H2P(1, 1, 16, true),
    ROMP_CONV_VARIANT_H2P(1, 1, 32, true),
    ROMP_CONV_VARIANT_H2Q(2, 2, 16), ROMP_CONV_VARIANT_H2Q(2, 2, 32), ROMP_CONV_VARIANT_H2Q(4, 1, 32), ROMP_CONV_VARIANT_H2Q(2, 1, 32),
    ROMP_CONV_VARIANT_H2Q(2, 1, 16), ROMP_CONV_VARIANT_H2Q(1, 2, 16), ROMP_CONV_VARIANT_H2Q(4, 2, 32),
};
ConvVariant* conv_variants_h2p(int* n) { *n = (int)(sizeof(kVariantsH2p) / sizeof(kVariantsH2p[0])); return kVariantsH2p; }

}  // namespace romp
