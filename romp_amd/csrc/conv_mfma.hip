// conv_mfma.hip -- NHWC float32 convolution (1x1 / 3x3, stride 1 / 2) as an implicit GEMM on the
// gfx950 f32-input matrix cores, with the inference BatchNorm (folded to scale/shift), the
// residual add and the ReLU fused into the epilogue.
//
// Replaces, per layer, the conv2d + batch_norm + add + relu op sequence the reference launches
// (BasicBlock.forward model.py:67-83, Bottleneck.forward :103-123, transition / fuse / head convs).
//
// GEMM view:  M = Cout, N = output pixels (B*Ho*Wo), K = taps*Cin.
//   v_mfma_f32_32x32x2_f32: A[i=lane&31][k=lane>>5], B[k=lane>>5][j=lane&31],
//   D[row][col]: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
//   A rows are output channels (weights), B columns are pixels: a lane ends up with 4 groups of 4
//   CONSECUTIVE channels of ONE pixel, so the epilogue loads scale/shift/residual and stores the
//   result as float4 (NHWC keeps channels contiguous).
//   The arithmetic is exact f32 (one rounding per product, f32 accumulate) -- the parity mode the
//   1e-4 gate needs; gfx950 has no TF32-like shortcut.
//
// Workgroup = 4 waves (one per SIMD), PERSISTENT: it pulls work items (pixel tile x channel slice
// x group) from a per-XCD queue (one returning atomicAdd per item, issued a whole item ahead), so
// all CUs finish within one item of each other regardless of how the item count divides the chip,
// and the channel slices of one pixel tile run back-to-back on the same XCD (shared L2).
// A work item = TH x TW output pixels x NT*32 output channels; each wave owns MT pixel blocks
// (32 pixels each) x NT channel blocks.  Per input-channel chunk (CK channels) the haloed input
// tile and the weight slab are staged in LDS; the NEXT stage's global loads (next chunk, or the
// next item's first chunk) are issued before the MFMA loop of the current stage and written to
// LDS after it (issue-early / write-late), so HBM/L2 latency hides under the MFMAs and there is no
// exposed prologue between items.  LDS pixel stride is CK+4 floats: the ds_read_b128 fragment
// reads of 16 consecutive pixels hit 16 distinct 16-byte bank slots (conflict-free at stride 1,
// 2-way at stride 2).
#include "common.h"
#include <stdlib.h>

namespace romp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float* in; const float* w; const float* scale; const float* shift; const float* res;
    float* out;
    const uint4* w3;          // bf16x3-split weights (conv_bx3_kernel), or nullptr
    int* queue;               // 8 per-XCD work counters, QUEUE_STRIDE ints apart, zeroed before the launch
    int H, W, Ho, Wo;
    int Cout;                 // valid output channels per group (store mask)
    int cin_valid;            // channels physically present in the input (loader mask)
    int cin_pad, cout_pad;    // packed weight dims
    int in_cs, in_co, in_gs;
    int out_cs, out_co, out_gs;
    int res_cs, res_co, res_gs;
    int relu;
    int tiles_x, tiles_y, tiles_total;
    int nslices, ns_total;    // channel slices per group; slices*groups
    int n_queues, per_queue;  // 8 (XCD-aware) or 1
    int vec_io;               // epilogue may use float4 loads/stores
    int w_gs;                 // floats per group in the packed weight
    int pad_h, pad_w;         // rows / columns of zero padding before the first tap
    int out_rs, out_bs;       // output row stride / image stride in floats (dense: Wo*out_cs, Ho*Wo*out_cs)
    int dbg;                  // ablation switches, env ROMP_CONV_DEBUG (timing experiments only: outputs are wrong).
                              // bits: 1 skip global loads / DMA, 2 skip LDS staging writes, 4 skip epilogue, 8 skip MFMA loop,
                              // 16 skip a stage barrier (f32 kernel), 32 return at once (launch cost), 64 / 128 (bxd) skip
                              // only the pixel loads / only the weight DMA.  scripts/conv_ablate.py and DESIGN.md §4 use them.
};

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int KS, int S, int MT, int NT, int TW, int CK>
struct ConvCfg {
    // KS = 1: 1x1, 2: 2x2 (one output parity of a ConvTranspose2d k4 s2), 3: 3x3, 13: 1x3 (Conv1d k=3 along W; rows of
    // the "image" are independent sequences).  The zero padding before the first tap is a run-time parameter.
    static constexpr int KH = (KS == 13) ? 1 : KS;
    static constexpr int KW = (KS == 13) ? 3 : KS;
    static constexpr int TAPS = KH * KW;
    static constexpr int RPB = 32 / TW;              // tile rows per 32-pixel block
    static constexpr int TH = 4 * MT * RPB;          // output tile rows
    static constexpr int HR = (TH - 1) * S + KH;     // haloed input rows
    static constexpr int HC = (TW - 1) * S + KW;
    static constexpr int PS = CK + 4;                // LDS floats per pixel (padded)
    static constexpr int NW = NT * 32;               // output channels per work item
    static constexpr int QC = CK / 4;                // float4 per pixel per chunk
    static constexpr int A_VEC = HR * HC * QC;
    static constexpr int B_VEC = TAPS * QC * NW;
    static constexpr int NA = (A_VEC + 255) / 256;
    static constexpr int NB = (B_VEC + 255) / 256;
    static constexpr int LDS_BYTES = (HR * HC * PS + TAPS * CK * NW + 4 * NW) * 4 + 16;
};

struct Item { int b, ty, tx, n0, g; };

__device__ __forceinline__ Item decode_item(const ConvParams& p, int q, int j, int NW) {
    const int s = j % p.ns_total, tl = j / p.ns_total;
    int t = tl * p.n_queues + q;
    Item it;
    it.g = s / p.nslices;
    it.n0 = (s % p.nslices) * NW;
    it.tx = t % p.tiles_x; t /= p.tiles_x;
    it.ty = t % p.tiles_y;
    it.b = t / p.tiles_y;
    return it;
}

// Epilogue of one work item: y = acc*scale + shift (+ residual) (ReLU), NHWC float4 stores.
// Lane owns pixel li of pixel-block m and channels n0 + n*32 + 8*g4 + 4*lh + {0..3}.
// All residual loads of the item are issued up front in ONE batch under ONE uniform branch (a
// branch per float4 serialises MT*NT*4 dependent global round trips -- that alone held the
// 3x3 kernels at ~100 TFLOP/s), ReLU is branch-free (max with 0 or -inf).
template <int KS, int S, int MT, int NT, int TW, int CK>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, const Item& cur, f32x16 (&acc)[MT][NT],
                                              const float* sSc, int wave, int li, int lh) {
    using C = ConvCfg<KS, S, MT, NT, TW, CK>;
    float* out = p.out + (size_t)cur.b * p.out_bs + p.out_co + cur.g * p.out_gs;
    const float* res = p.res ? p.res + (size_t)cur.b * p.Ho * p.Wo * p.res_cs + p.res_co + cur.g * p.res_gs : nullptr;
    const float floor_v = p.relu ? 0.f : -__builtin_inff();
    unsigned pixo[MT], outo[MT];                     // residual pixel index; output offset (row / pixel strides may be sparse)
    bool rowok[MT];                                  // partial tiles along H (e.g. Conv1d over B < TH sequences)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mb = wave * MT + m;
        const int oy = cur.ty * C::TH + mb * C::RPB + li / TW, ox = cur.tx * TW + li % TW;
        rowok[m] = oy < p.Ho;
        pixo[m] = rowok[m] ? (unsigned)(oy * p.Wo + ox) : 0u;
        outo[m] = rowok[m] ? (unsigned)(oy * p.out_rs + ox * p.out_cs) : 0u;
    }
    if (p.vec_io) {
        float4 r[MT][NT][4];
        if (res) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        r[m][n][g4] = ldg4(res + (pixo[m] * (unsigned)p.res_cs + (unsigned)(cur.n0 + n * 32 + g4 * 8 + lh * 4)));   // masked rows read pixel 0 (valid memory)
        } else {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) r[m][n][g4] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int cl = n * 32 + g4 * 8 + lh * 4;
                const float4 sc = *reinterpret_cast<const float4*>(sSc + cl);
                const float4 sh = *reinterpret_cast<const float4*>(sSc + C::NW + cl);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    float4 v;
                    v.x = fmaxf(fmaf(acc[m][n][g4 * 4 + 0], sc.x, sh.x) + r[m][n][g4].x, floor_v);
                    v.y = fmaxf(fmaf(acc[m][n][g4 * 4 + 1], sc.y, sh.y) + r[m][n][g4].y, floor_v);
                    v.z = fmaxf(fmaf(acc[m][n][g4 * 4 + 2], sc.z, sh.z) + r[m][n][g4].z, floor_v);
                    v.w = fmaxf(fmaf(acc[m][n][g4 * 4 + 3], sc.w, sh.w) + r[m][n][g4].w, floor_v);
                    if (rowok[m]) *reinterpret_cast<float4*>(out + (outo[m] + (unsigned)(cur.n0 + cl))) = v;
                }
            }
    } else {
        // scalar path: output convs of the head (Cout = 142 / 1 / 3 into unaligned NHWC slots)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int cl = n * 32 + g4 * 8 + lh * 4;
                    const int co = cur.n0 + cl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (co + e < p.Cout && rowok[m]) {
                            float t = fmaf(acc[m][n][g4 * 4 + e], sSc[cl + e], sSc[C::NW + cl + e]);
                            if (res) t += res[pixo[m] * (unsigned)p.res_cs + (unsigned)(co + e)];
                            out[outo[m] + (unsigned)(co + e)] = fmaxf(t, floor_v);
                        }
                    }
                }
    }
}

// One stage of the implicit GEMM: all taps x channel octets of the staged chunk.  The LDS fragment
// reads of step k+1 are issued BEFORE the MFMAs of step k (register double buffer, pinned with
// sched_barrier): an f32 MFMA group keeps the pipe busy for >= 512 cycles, so the ds_read latency
// is hidden instead of draining the matrix pipe at every step (hipcc otherwise sinks each read
// next to its use: `ds_read; s_waitcnt lgkmcnt(0); v_mfma`).
template <int KS, int S, int MT, int NT, int TW, int CK>
__device__ __forceinline__ void mma_stage(const float* sA, const float* sB, const int (&xoff)[MT], int woff,
                                          f32x16 (&acc)[MT][NT]) {
    using C = ConvCfg<KS, S, MT, NT, TW, CK>;
    constexpr int STEPS = C::TAPS * (CK / 8);
    float4 xf[2][MT], wf[2][NT];
    auto load = [&](int step, int buf) {
        const int tap = step / (CK / 8), q8 = step % (CK / 8);
        const int dy = tap / C::KW, dx = tap % C::KW;
#pragma unroll
        for (int m = 0; m < MT; ++m)
            xf[buf][m] = *reinterpret_cast<const float4*>(sA + xoff[m] + (dy * C::HC + dx) * C::PS + q8 * 8);
#pragma unroll
        for (int n = 0; n < NT; ++n)
            wf[buf][n] = *reinterpret_cast<const float4*>(sB + woff + ((tap * C::QC + q8 * 2) * C::NW + n * 32) * 4);
    };
    load(0, 0);
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        const int cb = step & 1;
        if (step + 1 < STEPS) load(step + 1, cb ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[cb][n].x, xf[cb][m].x, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[cb][n].y, xf[cb][m].y, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[cb][n].z, xf[cb][m].z, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[cb][n].w, xf[cb][m].w, acc[m][n], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int KS, int S, int MT, int NT, int TW, int CK>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvParams p) {
    if (p.dbg & 32) return;                            // ablation: launch cost only
    using C = ConvCfg<KS, S, MT, NT, TW, CK>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                                  // haloed pixels
    float* sB = smem + C::HR * C::HC * C::PS;          // weight slab
    float* sS = sB + C::TAPS * CK * C::NW;             // 2 slots x {scale[NW], shift[NW]} (current / next item)
    int* sQ = reinterpret_cast<int*>(sS + 4 * C::NW);  // work-queue mailbox

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int q = p.n_queues == 8 ? (blockIdx.x & 7) : 0;
    const int n_chunks = p.cin_pad / CK;

    // ---- work queue: the first item is static (this workgroup's rank within its queue -- no atomic round
    // trip before the first loads), every later one is counter + workgroups-per-queue, fetched a whole item
    // ahead.  The 8 per-XCD counters sit QUEUE_STRIDE ints apart (one cache line each).
    const int nwg_q = gridDim.x / p.n_queues;
    const int j_cur0 = blockIdx.x / p.n_queues;
    if (j_cur0 >= p.per_queue) return;
    if (tid == 0) sQ[1] = atomicAdd(p.queue + q * QUEUE_STRIDE, 1) + nwg_q;
    int j_cur = j_cur0;

    float4 ra[C::NA], rb[C::NB];
    unsigned ra_ok = 0;                                // bit k: ra[k] is a real (in-image, in-tile) load
    float rs = 0.f;                                    // one scale-or-shift value (threads < 2*NW)

    auto issue_loads = [&](const Item& it, int c0) {
        const float* in = p.in + (size_t)it.b * p.H * p.W * p.in_cs + p.in_co + it.g * p.in_gs;
        const float* wg = p.w + (size_t)it.g * p.w_gs;
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
            const bool ok = idx < C::A_VEC && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && c < p.cin_valid;
            ra[k] = ldg4(in + (ok ? (unsigned)((iy * p.W + ix) * p.in_cs + c) : 0u));
            ra_ok = k == 0 ? (ok ? 1u : 0u) : (ra_ok | ((ok ? 1u : 0u) << k));
        }
#pragma unroll
        for (int k = 0; k < C::NB; ++k) {
            const int idx = tid + k * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < C::B_VEC) {
                const int j = idx % C::NW, tq = idx / C::NW;
                const int qq = tq % C::QC, tap = tq / C::QC;
                v = ldg4(wg + (unsigned)(((tap * (p.cin_pad >> 2) + (c0 >> 2) + qq) * p.cout_pad + it.n0 + j) * 4));
            }
            rb[k] = v;
        }
        if (c0 == 0 && tid < 2 * C::NW) {
            const float* src = tid < C::NW ? p.scale : p.shift;
            rs = src[it.g * p.cout_pad + it.n0 + (tid & (C::NW - 1))];
        }
    };
    auto write_lds = [&](bool first_chunk, int slot) {
#pragma unroll
        for (int k = 0; k < C::NA; ++k) {
            const int idx = tid + k * 256;
            if (idx < C::A_VEC) {
                const int qq = idx % C::QC, pix = idx / C::QC;
                *reinterpret_cast<float4*>(sA + pix * C::PS + qq * 4) = ((ra_ok >> k) & 1u) ? ra[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int k = 0; k < C::NB; ++k) {
            const int idx = tid + k * 256;
            if (idx < C::B_VEC) *reinterpret_cast<float4*>(sB + idx * 4) = rb[k];
        }
        if (first_chunk && tid < 2 * C::NW) sS[slot * 2 * C::NW + tid] = rs;
    };

    // per-wave fragment base addresses (pixel fragments) and weight fragment base
    int xoff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mb = wave * MT + m;
        const int row = mb * C::RPB + li / TW, col = li % TW;
        xoff[m] = ((row * S) * C::HC + col * S) * C::PS + lh * 4;
    }
    const int woff = (lh * C::NW + li) * 4;

    Item cur = decode_item(p, q, j_cur, C::NW);
    issue_loads(cur, 0);
    write_lds(true, 0);
    __syncthreads();                                   // stage 0 in LDS; also publishes sQ[1]
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

    // Flat stage loop (stage = one channel chunk of one item).  Invariant at the top: the stage's
    // pixels/weights are in LDS and visible.  Per stage: issue the NEXT stage's global loads (next
    // chunk, or chunk 0 of the next item) -> MFMA loop -> barrier -> staging registers to LDS ->
    // (item finished: epilogue; the staging registers are dead by then) -> barrier.
    // One load site and one LDS-write site keep the compiler from hoisting per-item address math.
#pragma unroll 1
    while (true) {
        const bool last = ch + 1 == n_chunks;
        const bool pf = !last || have_next;          // is there a next stage to prefetch?
        Item tgt = last ? nxt : cur;
        const int c0 = last ? 0 : (ch + 1) * CK;
        if (ch == 0 && tid == 0) j_after = atomicAdd(p.queue + q * QUEUE_STRIDE, 1) + nwg_q;   // item after next
        if (pf && !(p.dbg & 1)) issue_loads(tgt, c0);
        if (!(p.dbg & 8)) mma_stage<KS, S, MT, NT, TW, CK>(sA, sB, xoff, woff, acc);
        if (ch == 0 && tid == 0) sQ[0] = j_after;
        if (!(p.dbg & 16)) __syncthreads();   // every wave finished reading this stage
        if (pf && !(p.dbg & 2)) write_lds(last, slot ^ 1);
        if (last) {
            if (!(p.dbg & 4)) conv_epilogue<KS, S, MT, NT, TW, CK>(p, cur, acc, sS + slot * 2 * C::NW, wave, li, lh);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        }
        if (last && !have_next) break;
        __syncthreads();                      // next stage (and sQ[0]) visible
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
}

// ------------------------------------------------------------------------------------------------
// Ping-pong variant: 8 waves per workgroup = two groups of 4 waves, each group an independent copy
// of the pipeline above (own work items, own LDS region), forced to ALTERNATE: while group A runs
// its MFMA phase, group B runs its memory phase (staging registers -> LDS, epilogue stores, residual
// loads), then they swap.  With 4-wave workgroups scheduled independently the two waves sharing a
// SIMD drift into the same phase and the matrix pipe idles ~35-40 % of the time (rocprofv3:
// SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE = 0.60); here every SIMD always has exactly one wave in
// its MFMA phase.  One workgroup-wide barrier per phase.  2 waves/SIMD => up to 256 VGPRs per wave.
template <int KS, int S, int MT, int NT, int TW, int CK>
__global__ __launch_bounds__(512) void conv_pp_kernel(ConvParams p) {
    if (p.dbg & 32) return;                            // ablation: launch cost only
    using C = ConvCfg<KS, S, MT, NT, TW, CK>;
    constexpr int GROUP_FLOATS = C::LDS_BYTES / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int grp = threadIdx.x >> 8;                  // wave group 0 / 1
    const int tid = threadIdx.x & 255;                 // thread within the group
    float* sA = smem + grp * GROUP_FLOATS;             // haloed pixels
    float* sB = sA + C::HR * C::HC * C::PS;            // weight slab
    float* sS = sB + C::TAPS * CK * C::NW;             // 2 slots x {scale[NW], shift[NW]}
    int* sQ = reinterpret_cast<int*>(sS + 4 * C::NW);  // per-group mailbox: [0],[1] items, [2] done flag
    int* sQ_other = reinterpret_cast<int*>(smem + (grp ^ 1) * GROUP_FLOATS + C::HR * C::HC * C::PS + C::TAPS * CK * C::NW + 4 * C::NW);

    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int q = p.n_queues == 8 ? (blockIdx.x & 7) : 0;
    const int n_chunks = p.cin_pad / CK;

    if (tid == 0) {
        sQ[0] = atomicAdd(p.queue + q * QUEUE_STRIDE, 1);
        sQ[1] = atomicAdd(p.queue + q * QUEUE_STRIDE, 1);
        sQ[2] = 0;
    }
    __syncthreads();
    int j_cur = sQ[0], j_next = sQ[1];
    bool done = j_cur >= p.per_queue;
    if (done && sQ_other[0] >= p.per_queue) return;    // uniform over the workgroup
    __syncthreads();                                   // everyone has read the mailboxes
    if (done && tid == 0) sQ[2] = 1;

    float4 ra[C::NA], rb[C::NB];
    unsigned ra_ok = 0;                                // bit k: ra[k] is a real (in-image, in-tile) load
    float rs = 0.f;

    auto issue_loads = [&](const Item& it, int c0) {
        const float* in = p.in + (size_t)it.b * p.H * p.W * p.in_cs + p.in_co + it.g * p.in_gs;
        const float* wg = p.w + (size_t)it.g * p.w_gs;
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
            const bool ok = idx < C::A_VEC && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && c < p.cin_valid;
            ra[k] = ldg4(in + (ok ? (unsigned)((iy * p.W + ix) * p.in_cs + c) : 0u));
            ra_ok = k == 0 ? (ok ? 1u : 0u) : (ra_ok | ((ok ? 1u : 0u) << k));
        }
#pragma unroll
        for (int k = 0; k < C::NB; ++k) {
            const int idx = tid + k * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < C::B_VEC) {
                const int j = idx % C::NW, tq = idx / C::NW;
                const int qq = tq % C::QC, tap = tq / C::QC;
                v = ldg4(wg + (unsigned)(((tap * (p.cin_pad >> 2) + (c0 >> 2) + qq) * p.cout_pad + it.n0 + j) * 4));
            }
            rb[k] = v;
        }
        if (c0 == 0 && tid < 2 * C::NW) {
            const float* src = tid < C::NW ? p.scale : p.shift;
            rs = src[it.g * p.cout_pad + it.n0 + (tid & (C::NW - 1))];
        }
    };
    auto write_lds = [&](bool first_chunk, int slot) {
#pragma unroll
        for (int k = 0; k < C::NA; ++k) {
            const int idx = tid + k * 256;
            if (idx < C::A_VEC) {
                const int qq = idx % C::QC, pix = idx / C::QC;
                *reinterpret_cast<float4*>(sA + pix * C::PS + qq * 4) = ((ra_ok >> k) & 1u) ? ra[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int k = 0; k < C::NB; ++k) {
            const int idx = tid + k * 256;
            if (idx < C::B_VEC) *reinterpret_cast<float4*>(sB + idx * 4) = rb[k];
        }
        if (first_chunk && tid < 2 * C::NW) sS[slot * 2 * C::NW + tid] = rs;
    };

    int xoff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mb = wave * MT + m;
        const int row = mb * C::RPB + li / TW, col = li % TW;
        xoff[m] = ((row * S) * C::HC + col * S) * C::PS + lh * 4;
    }
    const int woff = (lh * C::NW + li) * 4;

    Item cur, nxt;
    cur.b = cur.ty = cur.tx = cur.n0 = cur.g = 0;
    nxt = cur;
    bool have_next = false;
    if (!done) {
        cur = decode_item(p, q, j_cur, C::NW);
        issue_loads(cur, 0);
        write_lds(true, 0);
        have_next = j_next < p.per_queue;
        nxt = have_next ? decode_item(p, q, j_next, C::NW) : cur;
    }
    int slot = 0, ch = 0, j_after = 0x7fffffff;
    bool last = false, pf = false;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    __syncthreads();

#pragma unroll 1
    for (int phase = 0;; ++phase) {
        const int step = phase - grp;                  // group g: compute on even steps, memory on odd
        if (!done && step >= 0) {
            if ((step & 1) == 0) {
                // ---------------- MFMA phase of stage (cur, ch)
                last = ch + 1 == n_chunks;
                pf = !last || have_next;
                Item tgt = last ? nxt : cur;
                const int c0 = last ? 0 : (ch + 1) * CK;
                if (ch == 0 && tid == 0) j_after = atomicAdd(p.queue + q * QUEUE_STRIDE, 1);
                if (pf) issue_loads(tgt, c0);
                mma_stage<KS, S, MT, NT, TW, CK>(sA, sB, xoff, woff, acc);
                if (ch == 0 && tid == 0) sQ[0] = j_after;
            } else {
                // ---------------- memory phase: staging registers -> LDS, epilogue of a finished item
                if (pf) write_lds(last, slot ^ 1);
                if (last) {
                    conv_epilogue<KS, S, MT, NT, TW, CK>(p, cur, acc, sS + slot * 2 * C::NW, wave, li, lh);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
                    if (!have_next) {
                        done = true;
                        if (tid == 0) sQ[2] = 1;
                    } else {
                        cur = nxt;
                        slot ^= 1;
                        ch = 0;
                        j_next = sQ[0];               // written in this item's first MFMA phase (>= 1 barrier ago)
                        have_next = j_next < p.per_queue;
                        if (have_next) nxt = decode_item(p, q, j_next, C::NW);
                    }
                } else {
                    ++ch;
                }
            }
        }
        __syncthreads();
        if (sQ[2] && sQ_other[2]) break;               // both groups finished (uniform)
    }
}

// ------------------------------------------------------------------------------------------------
// bf16x3 variant: the same persistent implicit GEMM, but every f32 operand is split into three bf16
// pieces (x = x1 + x2 + x3 exactly: 3 x 8 significand bits) and the product is formed on the bf16
// matrix pipe from the six piece products with weight >= 2^-16 -- x1w1, x1w2, x2w1, x1w3, x2w2, x3w1
// (the dropped ones are below the f32 rounding of the product) -- accumulated in f32 by the MFMA.
// Result: f32-accurate convolution (same parity gates as the f32-MFMA kernel) at 6 x 32 cycles per
// 32x32x16 block instead of 8 x 64 cycles: 2.67x the f32 matrix rate.  Activations stay f32 in HBM;
// they are split while being staged into LDS, the weights are pre-split on the host
// (plan.py:pack_conv_weight_bx3).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KS, int S, int MT, int NT, int TW, int CK>
struct BxCfg {
    using C = ConvCfg<KS, S, MT, NT, TW, CK>;
    static constexpr int K16 = CK / 16;
    static constexpr int PSB = 3 * CK * 2 + 16;            // LDS bytes per pixel: 3 pieces x CK bf16 (+16 pad)
    static constexpr int A_BYTES = C::HR * C::HC * PSB;
    static constexpr int B_UNITS = C::TAPS * 3 * K16 * 2 * C::NW;      // 16-byte units: [tap][piece][k16][kg][NW]
    static constexpr int NB = (B_UNITS + 255) / 256;
    static constexpr int LDS_BYTES = A_BYTES + B_UNITS * 16 + 4 * C::NW * 4 + 16;
};

__device__ __forceinline__ void split3(float x, unsigned short& p1, unsigned short& p2, unsigned short& p3) {
    const __bf16 b1 = (__bf16)x;
    const float r1 = x - (float)b1;                        // exact
    const __bf16 b2 = (__bf16)r1;
    const float r2 = r1 - (float)b2;                       // exact
    const __bf16 b3 = (__bf16)r2;
    p1 = __builtin_bit_cast(unsigned short, b1);
    p2 = __builtin_bit_cast(unsigned short, b2);
    p3 = __builtin_bit_cast(unsigned short, b3);
}

template <int KS, int S, int MT, int NT, int TW, int CK>
__device__ __forceinline__ void mma_stage_bx3(const char* sA, const char* sB, const int (&xoff)[MT], int woff,
                                              f32x16 (&acc)[MT][NT]) {
    using C = ConvCfg<KS, S, MT, NT, TW, CK>;
    using X = BxCfg<KS, S, MT, NT, TW, CK>;
    constexpr int STEPS = C::TAPS * X::K16;
    constexpr bool DB = MT * NT <= 2;                  // register double-buffer of the fragments only for small tiles
    bf16x8 xf[DB ? 2 : 1][MT][3], wf[DB ? 2 : 1][NT][3];
    auto load = [&](int step, int buf) {
        const int tap = step / X::K16, k16 = step % X::K16;
        const int dy = tap / C::KW, dx = tap % C::KW;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                xf[buf][m][pc] = *reinterpret_cast<const bf16x8*>(sA + xoff[m] + (dy * C::HC + dx) * X::PSB + pc * (CK * 2) + k16 * 32);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                wf[buf][n][pc] = *reinterpret_cast<const bf16x8*>(sB + woff + ((((tap * 3 + pc) * X::K16 + k16) * 2) * C::NW + n * 32) * 16);
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
            for (int n = 0; n < NT; ++n) {                // smallest terms first
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][n][2], xf[cb][m][0], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][n][1], xf[cb][m][1], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][n][0], xf[cb][m][2], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][n][1], xf[cb][m][0], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][n][0], xf[cb][m][1], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][n][0], xf[cb][m][0], acc[m][n], 0, 0, 0);
            }
        if (DB) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int KS, int S, int MT, int NT, int TW, int CK>
__global__ __launch_bounds__(256, 2) void conv_bx3_kernel(ConvParams p) {
    if (p.dbg & 32) return;                            // ablation: launch cost only
    using C = ConvCfg<KS, S, MT, NT, TW, CK>;
    using X = BxCfg<KS, S, MT, NT, TW, CK>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sA = reinterpret_cast<char*>(smem);          // haloed pixels, 3 bf16 pieces per channel
    char* sB = sA + X::A_BYTES;                        // weight slab (pre-split)
    float* sS = reinterpret_cast<float*>(sB + X::B_UNITS * 16);
    int* sQ = reinterpret_cast<int*>(sS + 4 * C::NW);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int q = p.n_queues == 8 ? (blockIdx.x & 7) : 0;
    const int n_chunks = p.cin_pad / CK;

    const int nwg_q = gridDim.x / p.n_queues;
    const int j_cur0 = blockIdx.x / p.n_queues;
    if (j_cur0 >= p.per_queue) return;
    if (tid == 0) sQ[1] = atomicAdd(p.queue + q * QUEUE_STRIDE, 1) + nwg_q;
    int j_cur = j_cur0;

    float4 ra[C::NA];
    unsigned ra_ok = 0;
    uint4 rb[X::NB];
    float rs = 0.f;

    auto issue_loads = [&](const Item& it, int c0) {
        const float* in = p.in + (size_t)it.b * p.H * p.W * p.in_cs + p.in_co + it.g * p.in_gs;
        const uint4* wg = p.w3 + (size_t)it.g * (C::TAPS * (p.cin_pad >> 4) * 6 * p.cout_pad);
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
            const bool ok = idx < C::A_VEC && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && c < p.cin_valid;
            ra[k] = ldg4(in + (ok ? (unsigned)((iy * p.W + ix) * p.in_cs + c) : 0u));
            ra_ok = k == 0 ? (ok ? 1u : 0u) : (ra_ok | ((ok ? 1u : 0u) << k));
        }
#pragma unroll
        for (int k = 0; k < X::NB; ++k) {
            const int idx = tid + k * 256;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (idx < X::B_UNITS) {
                // LDS unit index: ((((tap*3 + pc)*K16 + k16)*2 + kg)*NW + j
                int r = idx;
                const int j = r % C::NW; r /= C::NW;
                const int kg = r & 1; r >>= 1;
                const int k16 = r % X::K16; r /= X::K16;
                const int pc = r % 3;
                const int tap = r / 3;
                // global: [tap][cin_pad/16][piece][kg][cout_pad] units
                v = wg[(unsigned)(((((tap * (p.cin_pad >> 4) + (c0 >> 4) + k16) * 3 + pc) * 2 + kg) * p.cout_pad) + it.n0 + j)];
            }
            rb[k] = v;
        }
        if (c0 == 0 && tid < 2 * C::NW) {
            const float* src = tid < C::NW ? p.scale : p.shift;
            rs = src[it.g * p.cout_pad + it.n0 + (tid & (C::NW - 1))];
        }
    };
    auto write_lds = [&](bool first_chunk, int slot) {
#pragma unroll
        for (int k = 0; k < C::NA; ++k) {
            const int idx = tid + k * 256;
            if (idx < C::A_VEC) {
                const int qq = idx % C::QC, pix = idx / C::QC;
                unsigned short h[4][3];
                const float4 av = ((ra_ok >> k) & 1u) ? ra[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                split3(av.x, h[0][0], h[0][1], h[0][2]);
                split3(av.y, h[1][0], h[1][1], h[1][2]);
                split3(av.z, h[2][0], h[2][1], h[2][2]);
                split3(av.w, h[3][0], h[3][1], h[3][2]);
                char* dst = sA + pix * X::PSB + qq * 8;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
                    uint2 u;
                    u.x = (unsigned)h[0][pc] | ((unsigned)h[1][pc] << 16);
                    u.y = (unsigned)h[2][pc] | ((unsigned)h[3][pc] << 16);
                    *reinterpret_cast<uint2*>(dst + pc * (CK * 2)) = u;
                }
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
        xoff[m] = ((row * S) * C::HC + col * S) * X::PSB + lh * 16;
    }
    const int woff = (lh * C::NW + li) * 16;

    Item cur = decode_item(p, q, j_cur, C::NW);
    issue_loads(cur, 0);
    write_lds(true, 0);
    __syncthreads();                                   // stage 0 in LDS; also publishes sQ[1]
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
        if (!(p.dbg & 8)) mma_stage_bx3<KS, S, MT, NT, TW, CK>(sA, sB, xoff, woff, acc);
        if (ch == 0 && tid == 0) sQ[0] = j_after;
        __syncthreads();
        if (pf && !(p.dbg & 2)) write_lds(last, slot ^ 1);
        if (last) {
            if (!(p.dbg & 4)) conv_epilogue<KS, S, MT, NT, TW, CK>(p, cur, acc, sS + slot * 2 * C::NW, wave, li, lh);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        }
        if (last && !have_next) break;
        __syncthreads();
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
}

// ------------------------------------------------------------------------------------------------
// bf16x3, second generation ("bxd"): same arithmetic as conv_bx3_kernel, different data movement.
// Measured on conv_bx3 (64->64 @64x64, B=32): LDS traffic (fragment reads + staging writes) ran at
// ~95 % of the LDS peak at the MFMA rate the kernel was aiming for, and the re-fetch of the pre-split
// weight slab by every 128-pixel workgroup tile drew ~10 TB/s from L2.  Here:
//   * the weight slab is staged one TAP ROW (KW taps) at a time, by LDS-DMA (global_load_lds_dwordx4:
//     no staging VGPRs, no ds_write pass), double-buffered: the DMA of sub-stage n+1 runs under the
//     MFMAs of sub-stage n and is retired (vmcnt(0)) before the barrier that ends sub-stage n;
//   * the 56 VGPRs the weight staging used are gone, so a wave can own a 2x2 block tile (64 pixels x
//     64 channels) at TWO workgroups per CU without spilling: 12 fragment reads per 24 MFMAs instead
//     of 9 per 12, and each weight byte fetched from L2 feeds twice the pixels;
//   * activations are still staged through registers (they must be split into bf16 pieces on the way).
// LDS per workgroup (MT=NT=2, TW=16, CK=16): 36.3 KB pixels + 2 x 18.4 KB weight rows = 74 KB.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int KS, int S, int MT, int NT, int TW, int CK>
struct BdCfg {
    using C = ConvCfg<KS, S, MT, NT, TW, CK>;
    static constexpr int K16 = CK / 16;
    static constexpr int PSB = 3 * CK * 2 + 16;
    static constexpr int A_BYTES = C::HR * C::HC * PSB;
    static constexpr int SUB_UNITS = C::KW * 3 * K16 * 2 * C::NW;      // 16-byte units of one tap row: [dx][piece][k16][kg][NW]
    static constexpr int NBD = (SUB_UNITS + 255) / 256;
    static constexpr int LDS_BYTES = A_BYTES + 2 * SUB_UNITS * 16 + 4 * C::NW * 4 + 32;
    static_assert(SUB_UNITS % 64 == 0, "a wave's LDS-DMA writes 64 consecutive units");
};

template <int KS, int S, int MT, int NT, int TW, int CK>
__global__ __launch_bounds__(256, 2) void conv_bxd_kernel(ConvParams p) {
    if (p.dbg & 32) return;                            // ablation: launch cost only
    using C = ConvCfg<KS, S, MT, NT, TW, CK>;
    using X = BdCfg<KS, S, MT, NT, TW, CK>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sA = reinterpret_cast<char*>(smem);
    char* sB = sA + X::A_BYTES;                                   // two tap-row buffers
    float* sS = reinterpret_cast<float*>(sB + 2 * X::SUB_UNITS * 16);
    int* sQ = reinterpret_cast<int*>(sS + 4 * C::NW);             // [0..1] first two items, [2..3] item-ahead mailbox

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int q = p.n_queues == 8 ? (blockIdx.x & 7) : 0;
    const int n_chunks = p.cin_pad / CK;
    const int cin16 = p.cin_pad >> 4;

    const int nwg_q = gridDim.x / p.n_queues;
    const int j_cur0 = blockIdx.x / p.n_queues;
    if (j_cur0 >= p.per_queue) return;
    if (tid == 0) sQ[1] = atomicAdd(p.queue + q * QUEUE_STRIDE, 1) + nwg_q;
    int j_cur = j_cur0;

    float4 ra[C::NA];
    unsigned ra_ok = 0;
    float rs = 0.f;

    // LDS-DMA of the weight units of tap row `row`, channel chunk c0, into buffer `buf`
    auto issue_B = [&](const Item& it, int c0, int row, int buf) {
        const uint4* wg = p.w3 + (size_t)it.g * (C::TAPS * cin16 * 6 * p.cout_pad);
        char* dst = sB + buf * (X::SUB_UNITS * 16);
#pragma unroll
        for (int k = 0; k < X::NBD; ++k) {
            if (k * 256 + wave * 64 < X::SUB_UNITS) {             // wave-uniform
                int r = k * 256 + tid;
                const int j = r % C::NW; r /= C::NW;
                const int kg = r & 1; r >>= 1;
                const int k16 = r % X::K16; r /= X::K16;
                const int pc = r % 3;
                const int dx = r / 3;
                const int tap = row * C::KW + dx;
                const uint4* src = wg + (unsigned)(((((tap * cin16 + (c0 >> 4) + k16) * 3 + pc) * 2 + kg) * p.cout_pad) + it.n0 + j);
                __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(dst + (k * 256 + wave * 64) * 16), 16, 0, 0);
            }
        }
    };
    auto issue_A = [&](const Item& it, int c0) {
        const float* in = p.in + (size_t)it.b * p.H * p.W * p.in_cs + p.in_co + it.g * p.in_gs;
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
            const bool ok = idx < C::A_VEC && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && c < p.cin_valid;
            ra[k] = ldg4(in + (ok ? (unsigned)((iy * p.W + ix) * p.in_cs + c) : 0u));
            ra_ok = k == 0 ? (ok ? 1u : 0u) : (ra_ok | ((ok ? 1u : 0u) << k));
        }
        if (c0 == 0 && tid < 2 * C::NW) {
            const float* src = tid < C::NW ? p.scale : p.shift;
            rs = src[it.g * p.cout_pad + it.n0 + (tid & (C::NW - 1))];
        }
    };
    auto write_A = [&](bool first_chunk, int slot) {
#pragma unroll
        for (int k = 0; k < C::NA; ++k) {
            const int idx = tid + k * 256;
            if (idx < C::A_VEC) {
                const int qq = idx % C::QC, pix = idx / C::QC;
                unsigned short h[4][3];
                const float4 av = ((ra_ok >> k) & 1u) ? ra[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                split3(av.x, h[0][0], h[0][1], h[0][2]);
                split3(av.y, h[1][0], h[1][1], h[1][2]);
                split3(av.z, h[2][0], h[2][1], h[2][2]);
                split3(av.w, h[3][0], h[3][1], h[3][2]);
                char* dst = sA + pix * X::PSB + qq * 8;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
                    uint2 u;
                    u.x = (unsigned)h[0][pc] | ((unsigned)h[1][pc] << 16);
                    u.y = (unsigned)h[2][pc] | ((unsigned)h[3][pc] << 16);
                    *reinterpret_cast<uint2*>(dst + pc * (CK * 2)) = u;
                }
            }
        }
        if (first_chunk && tid < 2 * C::NW) sS[slot * 2 * C::NW + tid] = rs;
    };

    int xoff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mb = wave * MT + m;
        const int row = mb * C::RPB + li / TW, col = li % TW;
        xoff[m] = ((row * S) * C::HC + col * S) * X::PSB + lh * 16;
    }
    const int woff = (lh * C::NW + li) * 16;

    Item cur = decode_item(p, q, j_cur, C::NW);
    issue_B(cur, 0, 0, 0);
    issue_A(cur, 0);
    write_A(true, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // stage 0 in LDS; also publishes sQ[1]
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
        if (!(p.dbg & 8)) {
            const char* sBc = sB + bbuf * (X::SUB_UNITS * 16);
#pragma unroll
            for (int dx = 0; dx < C::KW; ++dx)
#pragma unroll
                for (int k16 = 0; k16 < X::K16; ++k16) {
                    bf16x8 xf[MT][3], wf[NT][3];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc)
                            xf[m][pc] = *reinterpret_cast<const bf16x8*>(sA + xoff[m] + (row * C::HC + dx) * X::PSB + pc * (CK * 2) + k16 * 32);
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc)
                            wf[n][pc] = *reinterpret_cast<const bf16x8*>(sBc + woff + ((((dx * 3 + pc) * X::K16 + k16) * 2) * C::NW + n * 32) * 16);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {            // smallest terms first
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][2], xf[m][0], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][1], xf[m][1], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][0], xf[m][2], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][1], xf[m][0], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][0], xf[m][1], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][0], xf[m][0], acc[m][n], 0, 0, 0);
                        }
                }
        }
        if (ch == 0 && row == 0 && tid == 0) sQ[2 + par] = j_after;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's LDS-DMA has landed (and ra is in)
        __syncthreads();                                           // all waves: done reading bbuf / sA, DMA visible
        bbuf ^= 1;
        if (last_row) {
            if (pfA && !(p.dbg & 2)) write_A(last_ch, slot ^ 1);
            if (last_ch) {
                if (!(p.dbg & 4)) conv_epilogue<KS, S, MT, NT, TW, CK>(p, cur, acc, sS + slot * 2 * C::NW, wave, li, lh);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
                if (!have_next) break;
            }
            __syncthreads();                                       // next chunk's pixels visible
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
}

// Bring-up cross-check: one thread per output element, same packed weights, plain FMA loop.
__global__ void conv_naive_kernel(ConvParams p, int KS, int S, int B, int groups) {
    const size_t total = (size_t)B * p.Ho * p.Wo * p.Cout * groups;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        size_t r = t;
        const int co = r % p.Cout; r /= p.Cout;
        const int g = r % groups; r /= groups;
        const int ox = r % p.Wo; r /= p.Wo;
        const int oy = r % p.Ho;
        const int b = r / p.Ho;
        const float* in = p.in + (size_t)b * p.H * p.W * p.in_cs + p.in_co + g * p.in_gs;
        const float* wg = p.w + (size_t)g * p.w_gs;
        const int KH = KS == 13 ? 1 : KS, KW = KS == 13 ? 3 : KS;
        float acc = 0.f;
        for (int tap = 0; tap < KH * KW; ++tap) {
            const int iy = oy * S - p.pad_h + tap / KW, ix = ox * S - p.pad_w + tap % KW;
            if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.W) continue;
            const float* px = in + ((size_t)iy * p.W + ix) * p.in_cs;
            for (int c = 0; c < p.cin_valid; ++c)
                acc = fmaf(px[c], wg[(((size_t)tap * (p.cin_pad >> 2) + (c >> 2)) * p.cout_pad + co) * 4 + (c & 3)], acc);
        }
        float v = fmaf(acc, p.scale[g * p.cout_pad + co], p.shift[g * p.cout_pad + co]);
        const size_t pix = ((size_t)b * p.Ho + oy) * p.Wo + ox;
        if (p.res) v += p.res[pix * p.res_cs + p.res_co + g * p.res_gs + co];
        if (p.relu) v = fmaxf(v, 0.f);
        p.out[(size_t)b * p.out_bs + (size_t)oy * p.out_rs + (size_t)ox * p.out_cs + p.out_co + g * p.out_gs + co] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------------
typedef void (*conv_fn)(ConvParams);
struct ConvVariant { int ks, s, mt, nt, tw, ck; conv_fn fn; int lds; int th; int occ; int pp; int math; };

#define ROMP_CONV_VARIANT(KS, S, MT, NT, TW, CK)                                      \
    { KS, S, MT, NT, TW, CK, conv_mfma_kernel<KS, S, MT, NT, TW, CK>,                 \
      ConvCfg<KS, S, MT, NT, TW, CK>::LDS_BYTES, ConvCfg<KS, S, MT, NT, TW, CK>::TH, 0, 0, 0 }
#define ROMP_CONV_VARIANT_PP(KS, S, MT, NT, TW, CK)                                   \
    { KS, S, MT, NT, TW, CK, conv_pp_kernel<KS, S, MT, NT, TW, CK>,                   \
      2 * ConvCfg<KS, S, MT, NT, TW, CK>::LDS_BYTES, ConvCfg<KS, S, MT, NT, TW, CK>::TH, 0, 1, 0 }
#define ROMP_CONV_VARIANT_BXD(KS, S, MT, NT, TW, CK)                                  \
    { KS, S, MT, NT, TW, CK, conv_bxd_kernel<KS, S, MT, NT, TW, CK>,                  \
      BdCfg<KS, S, MT, NT, TW, CK>::LDS_BYTES, ConvCfg<KS, S, MT, NT, TW, CK>::TH, 0, 0, 2 }
#define ROMP_CONV_VARIANT_BX3(KS, S, MT, NT, TW, CK)                                  \
    { KS, S, MT, NT, TW, CK, conv_bx3_kernel<KS, S, MT, NT, TW, CK>,                  \
      BxCfg<KS, S, MT, NT, TW, CK>::LDS_BYTES, ConvCfg<KS, S, MT, NT, TW, CK>::TH, 0, 0, 1 }

static ConvVariant kVariants[] = {
    // 3x3 stride 1
    ROMP_CONV_VARIANT(3, 1, 2, 1, 32, 16), ROMP_CONV_VARIANT(3, 1, 2, 2, 32, 16),
    ROMP_CONV_VARIANT(3, 1, 1, 1, 32, 16), ROMP_CONV_VARIANT(3, 1, 1, 2, 32, 16),
    ROMP_CONV_VARIANT(3, 1, 4, 1, 32, 16),
    ROMP_CONV_VARIANT(3, 1, 1, 1, 16, 16), ROMP_CONV_VARIANT(3, 1, 1, 2, 16, 16),
    ROMP_CONV_VARIANT(3, 1, 2, 1, 16, 16), ROMP_CONV_VARIANT(3, 1, 2, 2, 16, 16),
    // 3x3 stride 2
    ROMP_CONV_VARIANT(3, 2, 1, 1, 32, 16), ROMP_CONV_VARIANT(3, 2, 1, 2, 32, 16),
    ROMP_CONV_VARIANT(3, 2, 1, 1, 16, 16), ROMP_CONV_VARIANT(3, 2, 1, 2, 16, 16),
    ROMP_CONV_VARIANT(3, 2, 1, 1, 32, 8), ROMP_CONV_VARIANT(3, 2, 1, 2, 32, 8),
    ROMP_CONV_VARIANT(3, 2, 1, 1, 16, 8), ROMP_CONV_VARIANT(3, 2, 1, 2, 16, 8),
    // 1x1
    ROMP_CONV_VARIANT(1, 1, 2, 1, 32, 32), ROMP_CONV_VARIANT(1, 1, 2, 2, 32, 32),
    ROMP_CONV_VARIANT(1, 1, 1, 1, 32, 32), ROMP_CONV_VARIANT(1, 1, 1, 2, 32, 32),
    ROMP_CONV_VARIANT(1, 1, 4, 1, 32, 32), ROMP_CONV_VARIANT(1, 1, 4, 2, 32, 32),
    ROMP_CONV_VARIANT(1, 1, 1, 1, 16, 32), ROMP_CONV_VARIANT(1, 1, 1, 2, 16, 32),
    ROMP_CONV_VARIANT(1, 1, 2, 2, 16, 32),
    ROMP_CONV_VARIANT(1, 1, 2, 1, 32, 16), ROMP_CONV_VARIANT(1, 1, 2, 2, 32, 16),
    // 1x1 stride 2 (ResNet-50 downsample branches) and 2x2 (the four output parities of ConvTranspose2d k4 s2 p1)
    ROMP_CONV_VARIANT(1, 2, 1, 2, 16, 32), ROMP_CONV_VARIANT(1, 2, 1, 1, 16, 32), ROMP_CONV_VARIANT(1, 2, 2, 2, 16, 32),
    ROMP_CONV_VARIANT(2, 1, 1, 2, 16, 16), ROMP_CONV_VARIANT(2, 1, 2, 2, 16, 16), ROMP_CONV_VARIANT(2, 1, 2, 2, 32, 16),
    ROMP_CONV_VARIANT(2, 1, 1, 1, 16, 16),
    // 1x3 (Conv1d k=3: BEV bird's-eye-view head, bev/model.py:24-45,179-182)
    ROMP_CONV_VARIANT(13, 1, 2, 2, 32, 16), ROMP_CONV_VARIANT(13, 1, 1, 2, 32, 16), ROMP_CONV_VARIANT(13, 1, 1, 1, 32, 16),
    ROMP_CONV_VARIANT(13, 1, 2, 1, 32, 16), ROMP_CONV_VARIANT(13, 1, 1, 2, 32, 32),
    // bf16x3 split (f32-accurate on the bf16 matrix pipe)
    ROMP_CONV_VARIANT_BX3(3, 1, 2, 1, 32, 16), ROMP_CONV_VARIANT_BX3(3, 1, 2, 1, 16, 16),
    ROMP_CONV_VARIANT_BX3(3, 1, 2, 2, 32, 16), ROMP_CONV_VARIANT_BX3(3, 1, 2, 2, 16, 16),
    ROMP_CONV_VARIANT_BX3(3, 1, 1, 2, 32, 16), ROMP_CONV_VARIANT_BX3(3, 1, 1, 2, 16, 16),
    ROMP_CONV_VARIANT_BX3(3, 1, 4, 1, 32, 16), ROMP_CONV_VARIANT_BX3(3, 1, 1, 1, 16, 16),
    ROMP_CONV_VARIANT_BX3(3, 2, 1, 2, 16, 16), ROMP_CONV_VARIANT_BX3(3, 2, 1, 1, 16, 16), ROMP_CONV_VARIANT_BX3(3, 2, 1, 2, 32, 16),
    ROMP_CONV_VARIANT_BX3(1, 1, 2, 2, 32, 32), ROMP_CONV_VARIANT_BX3(1, 1, 2, 1, 32, 32), ROMP_CONV_VARIANT_BX3(1, 1, 1, 2, 16, 32),
    ROMP_CONV_VARIANT_BX3(1, 1, 2, 2, 32, 16), ROMP_CONV_VARIANT_BX3(13, 1, 1, 2, 32, 16), ROMP_CONV_VARIANT_BX3(13, 1, 2, 2, 32, 16),
    ROMP_CONV_VARIANT_BX3(2, 1, 1, 2, 16, 16), ROMP_CONV_VARIANT_BX3(2, 1, 2, 2, 16, 16), ROMP_CONV_VARIANT_BX3(1, 1, 1, 2, 32, 32),
    // bf16x3 with LDS-DMA weight rows
    ROMP_CONV_VARIANT_BXD(3, 1, 2, 2, 16, 16), ROMP_CONV_VARIANT_BXD(3, 1, 2, 2, 32, 16),
    ROMP_CONV_VARIANT_BXD(3, 1, 2, 1, 16, 16), ROMP_CONV_VARIANT_BXD(3, 1, 2, 1, 32, 16),
    ROMP_CONV_VARIANT_BXD(3, 1, 1, 2, 16, 16), ROMP_CONV_VARIANT_BXD(3, 1, 4, 1, 32, 16),
    // ping-pong (8 waves, two alternating groups)
    ROMP_CONV_VARIANT_PP(3, 1, 2, 1, 32, 16), ROMP_CONV_VARIANT_PP(3, 1, 2, 1, 16, 16),
    ROMP_CONV_VARIANT_PP(3, 1, 1, 2, 32, 16), ROMP_CONV_VARIANT_PP(3, 1, 1, 2, 16, 16),
    ROMP_CONV_VARIANT_PP(3, 1, 2, 2, 32, 16), ROMP_CONV_VARIANT_PP(3, 1, 2, 2, 16, 16),
    ROMP_CONV_VARIANT_PP(3, 1, 1, 1, 16, 16), ROMP_CONV_VARIANT_PP(3, 1, 1, 1, 32, 16),
    ROMP_CONV_VARIANT_PP(3, 2, 1, 2, 16, 16), ROMP_CONV_VARIANT_PP(3, 2, 1, 1, 16, 16),
    ROMP_CONV_VARIANT_PP(3, 2, 1, 2, 32, 8), ROMP_CONV_VARIANT_PP(3, 2, 1, 1, 32, 16),
    ROMP_CONV_VARIANT_PP(1, 1, 2, 2, 32, 32), ROMP_CONV_VARIANT_PP(1, 1, 2, 1, 32, 32),
    ROMP_CONV_VARIANT_PP(1, 1, 1, 2, 16, 32), ROMP_CONV_VARIANT_PP(1, 1, 2, 2, 32, 16),
};
static const int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
static bool g_attr_done = false;
static int g_num_cu = 256;
static int* g_queue_scratch = nullptr;      // for romp_conv_forward callers without an arena

static const int kMaxLds = 160 * 1024;

static int ensure_attrs() {
    if (g_attr_done) return ROMP_OK;
    int dev = 0;
    ROMP_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    ROMP_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    g_num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    for (int i = 0; i < kNumVariants; ++i) {
        if (kVariants[i].lds > kMaxLds) continue;
        if (kVariants[i].lds > 48 * 1024)
            ROMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kVariants[i].fn),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kVariants[i].lds));
        int occ = 0;
        ROMP_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kVariants[i].fn),
                                                                    kVariants[i].pp ? 512 : 256, kVariants[i].lds));
        kVariants[i].occ = occ > 0 ? occ : 1;
    }
    ROMP_HIP_CHECK(hipMalloc((void**)&g_queue_scratch, QUEUE_INTS * sizeof(int)));
    g_attr_done = true;
    return ROMP_OK;
}

// One-time per-process setup (LDS attributes, occupancy, scratch queue).  romp_net_create calls it so
// that it never runs inside a stream capture (hipMalloc / hipFuncSetAttribute are illegal there).
int conv_init() { return ensure_attrs(); }

static bool variant_ok(const ConvVariant& v, const romp_op& op, int Ho, int Wo) {
    if (v.lds > kMaxLds) return false;
    if (v.math && (op.weight_aux == nullptr || (op.cin_pad & 15))) return false;
    if (v.ks != op.ksize || v.s != op.stride) return false;
    if (Wo % v.tw) return false;                     // rows may be partial (masked), columns may not
    if (op.cin_pad % v.ck || op.cout_pad % (v.nt * 32)) return false;
    return true;
}

// Heuristic choice (used until romp_net_autotune has measured the alternatives).
static int choose_variant(const romp_op& op, int Ho, int Wo, int B) {
    int best = -1;
    double best_score = -1;
    for (int i = 0; i < kNumVariants; ++i) {
        const ConvVariant& v = kVariants[i];
        if (!variant_ok(v, op, Ho, Wo) || v.math) continue;     // bf16x3 variants are chosen by autotune / explicitly
        const long items = (long)B * ((Ho + v.th - 1) / v.th) * (Wo / v.tw) * (op.cout_pad / (v.nt * 32)) * op.groups;
        const double eff = (double)Ho / (((Ho + v.th - 1) / v.th) * v.th);   // partial row tiles waste MFMA work
        double fill = items >= 512 ? 1.0 : (double)items / 512.0;
        double score = eff * fill * (1.0 + 0.25 * (v.mt * v.nt - 1)) * (v.ck >= 16 ? 1.0 : 0.8) * (v.tw == 32 ? 1.05 : 1.0);
        if (v.mt * v.nt > 4) score *= 0.5;
        if (score > best_score) { best_score = score; best = i; }
    }
    return best;
}

static void out_dims(const romp_op& op, int* Ho, int* Wo) {
    const int kh = op.ksize == 13 ? 1 : op.ksize, kw = op.ksize == 13 ? 3 : op.ksize;
    if (op.ksize == 2) { *Ho = op.H / op.stride; *Wo = op.W / op.stride; return; }   // 2x2: pad_h + (the other side) = 1 in total
    *Ho = (op.H + 2 * (kh / 2) - kh) / op.stride + 1;
    *Wo = (op.W + 2 * (kw / 2) - kw) / op.stride + 1;
}

int conv_num_variants() { return kNumVariants; }

bool conv_variant_valid(const romp_op& op, int variant) {
    int Ho, Wo;
    out_dims(op, &Ho, &Wo);
    return variant >= 0 && variant < kNumVariants && variant_ok(kVariants[variant], op, Ho, Wo);
}

int launch_conv(const romp_op& op, const float* in, const float* res, float* out, int B, int mode,
                int variant, int* queue, hipStream_t st, int wg_cap) {
    ROMP_REQUIRE(op.ksize == 1 || op.ksize == 2 || op.ksize == 3 || op.ksize == 13, "conv: ksize %d unsupported", op.ksize);
    ROMP_REQUIRE(op.stride == 1 || op.stride == 2, "conv: stride %d unsupported", op.stride);
    ROMP_REQUIRE(op.groups >= 1, "conv: groups must be >= 1");
    ROMP_REQUIRE((op.in_cstride & 3) == 0 && (op.in_coff & 3) == 0 && (op.in_gstride & 3) == 0 && (op.Cin & 3) == 0,
                 "conv: input channels must be float4 aligned (cs %d co %d Cin %d)", op.in_cstride, op.in_coff, op.Cin);
    ConvParams p;
    p.in = in; p.w = op.weight; p.scale = op.scale; p.shift = op.shift; p.res = res; p.out = out;
    p.w3 = reinterpret_cast<const uint4*>(op.weight_aux);
    p.H = op.H; p.W = op.W;
    out_dims(op, &p.Ho, &p.Wo);
    p.Cout = op.Cout; p.cin_valid = op.Cin; p.cin_pad = op.cin_pad; p.cout_pad = op.cout_pad;
    p.in_cs = op.in_cstride; p.in_co = op.in_coff; p.in_gs = op.in_gstride;
    p.out_cs = op.out_cstride; p.out_co = op.out_coff; p.out_gs = op.out_gstride;
    p.res_cs = op.res_cstride; p.res_co = op.res_coff; p.res_gs = op.res_gstride;
    p.relu = op.relu;
    p.w_gs = (op.ksize == 13 ? 3 : op.ksize * op.ksize) * op.cin_pad * op.cout_pad;
    const int kh = op.ksize == 13 ? 1 : op.ksize, kw = op.ksize == 13 ? 3 : op.ksize;
    p.pad_h = op.pad_h >= 0 ? op.pad_h : kh / 2;
    p.pad_w = op.pad_w >= 0 ? op.pad_w : kw / 2;
    p.out_rs = op.out_rstride > 0 ? op.out_rstride : p.Wo * op.out_cstride;
    p.out_bs = op.out_bstride > 0 ? op.out_bstride : p.Ho * p.Wo * op.out_cstride;
    p.tiles_x = p.tiles_y = p.tiles_total = 1;
    p.nslices = p.ns_total = p.n_queues = p.per_queue = 1;
    p.queue = nullptr;
    { const char* e = getenv("ROMP_CONV_DEBUG"); p.dbg = e ? atoi(e) : 0; }
    p.vec_io = (op.Cout == op.cout_pad && (op.Cout & 3) == 0 && (op.out_cstride & 3) == 0 && (op.out_coff & 3) == 0 && (op.out_gstride & 3) == 0 &&
                (!res || ((op.res_cstride & 3) == 0 && (op.res_coff & 3) == 0 && (op.res_gstride & 3) == 0))) ? 1 : 0;
    if (mode == 1) {
        const size_t total = (size_t)B * p.Ho * p.Wo * op.Cout * op.groups;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 65535) blocks = 65535;
        hipLaunchKernelGGL(conv_naive_kernel, dim3(blocks), dim3(256), 0, st, p, op.ksize, op.stride, B, op.groups);
        ROMP_HIP_CHECK(hipGetLastError());
        return ROMP_OK;
    }
    int rc = ensure_attrs();
    if (rc) return rc;
    if (variant < 0) variant = choose_variant(op, p.Ho, p.Wo, B);
    ROMP_REQUIRE(variant >= 0 && variant < kNumVariants && variant_ok(kVariants[variant], op, p.Ho, p.Wo),
                 "conv: no kernel variant (%d) for k%d s%d Cin %d(pad %d) Cout %d(pad %d) out %dx%d", variant,
                 op.ksize, op.stride, op.Cin, op.cin_pad, op.Cout, op.cout_pad, p.Ho, p.Wo);
    const ConvVariant& v = kVariants[variant];
    p.tiles_x = p.Wo / v.tw;
    p.tiles_y = (p.Ho + v.th - 1) / v.th;
    p.tiles_total = B * p.tiles_x * p.tiles_y;
    p.nslices = op.cout_pad / (v.nt * 32);
    p.ns_total = p.nslices * op.groups;
    p.n_queues = (p.tiles_total % 8 == 0) ? 8 : 1;
    p.per_queue = (p.tiles_total / p.n_queues) * p.ns_total;
    if (queue == nullptr) {
        queue = g_queue_scratch;
        ROMP_HIP_CHECK(hipMemsetAsync(queue, 0, QUEUE_INTS * sizeof(int), st));
    }
    p.queue = queue;
    const long items = (long)p.tiles_total * p.ns_total;
    long grid = (long)g_num_cu * ((wg_cap > 0 && wg_cap < v.occ) ? wg_cap : v.occ);   // wg_cap: leave room for a co-resident kernel
    const long want = v.pp ? (items + 1) / 2 : items;         // a ping-pong workgroup runs two item streams
    if (grid > want) grid = want;
    if (p.n_queues == 8) grid = grid >= 8 ? (grid / 8) * 8 : 8;    // same number of workgroups per queue
    hipLaunchKernelGGL(v.fn, dim3((unsigned)grid), dim3(v.pp ? 512 : 256), v.lds, st, p);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

int describe_conv(const romp_op& op, int B, int variant, char* out, int n) {
    int Ho, Wo;
    out_dims(op, &Ho, &Wo);
    if (variant < 0) variant = choose_variant(op, Ho, Wo, B);
    ROMP_REQUIRE(variant >= 0 && variant < kNumVariants, "describe: no variant");
    const ConvVariant& v = kVariants[variant];
    snprintf(out, n, "%s_k%ds%d_mt%d_nt%d_tw%d_ck%d", v.math == 2 ? "conv_bxd" : v.math ? "conv_bx3" : (v.pp ? "conv_pp" : "conv_mfma"), v.ks, v.s, v.mt, v.nt,
             v.tw, v.ck);
    return ROMP_OK;
}

}  // namespace romp
