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
#include "conv_common.h"

namespace romp {

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
    char* sE = reinterpret_cast<char*>(sA) + C::LDS_MAIN + wave * EPI_WAVE;      // this wave's epilogue staging tile
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
            if (!(p.dbg & 4)) conv_epilogue<KS, S, MT, NT, TW, CK>(p, cur, acc, sS + slot * 2 * C::NW, sE, wave, li, lh);
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
    char* sE = reinterpret_cast<char*>(sA) + C::LDS_MAIN + wave * EPI_WAVE;      // this wave's epilogue staging tile
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
                    conv_epilogue<KS, S, MT, NT, TW, CK>(p, cur, acc, sS + slot * 2 * C::NW, sE, wave, li, lh);
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

#define ROMP_CONV_VARIANT(KS, S, MT, NT, TW, CK)                                      \
    { KS, S, MT, NT, TW, CK, conv_mfma_kernel<KS, S, MT, NT, TW, CK>,                 \
      ConvCfg<KS, S, MT, NT, TW, CK>::LDS_BYTES, ConvCfg<KS, S, MT, NT, TW, CK>::TH, 0, 0, 0, 0 }
#define ROMP_CONV_VARIANT_PP(KS, S, MT, NT, TW, CK)                                   \
    { KS, S, MT, NT, TW, CK, conv_pp_kernel<KS, S, MT, NT, TW, CK>,                   \
      2 * ConvCfg<KS, S, MT, NT, TW, CK>::LDS_BYTES, ConvCfg<KS, S, MT, NT, TW, CK>::TH, 0, 1, 0, 0 }
// math: 0 f32 MFMA, 1 bf16x3 (register-staged weights), 2 bf16x3 (LDS-DMA weight rows), 3 / 4 the same two for f16x2
static ConvVariant kVariantsF32[] = {
    ROMP_CONV_VARIANT(3, 1, 2, 1, 32, 16), ROMP_CONV_VARIANT(3, 1, 2, 2, 32, 16),
    ROMP_CONV_VARIANT(3, 1, 1, 1, 32, 16), ROMP_CONV_VARIANT(3, 1, 1, 2, 32, 16),
    ROMP_CONV_VARIANT(3, 1, 4, 1, 32, 16),
    ROMP_CONV_VARIANT(3, 1, 1, 1, 16, 16), ROMP_CONV_VARIANT(3, 1, 1, 2, 16, 16),
    ROMP_CONV_VARIANT(3, 1, 2, 1, 16, 16), ROMP_CONV_VARIANT(3, 1, 2, 2, 16, 16),
    ROMP_CONV_VARIANT(3, 2, 1, 1, 32, 16), ROMP_CONV_VARIANT(3, 2, 1, 2, 32, 16),
    ROMP_CONV_VARIANT(3, 2, 1, 1, 16, 16), ROMP_CONV_VARIANT(3, 2, 1, 2, 16, 16),
    ROMP_CONV_VARIANT(3, 2, 1, 1, 32, 8), ROMP_CONV_VARIANT(3, 2, 1, 2, 32, 8),
    ROMP_CONV_VARIANT(3, 2, 1, 1, 16, 8), ROMP_CONV_VARIANT(3, 2, 1, 2, 16, 8),
    ROMP_CONV_VARIANT(1, 1, 2, 1, 32, 32), ROMP_CONV_VARIANT(1, 1, 2, 2, 32, 32),
    ROMP_CONV_VARIANT(1, 1, 1, 1, 32, 32), ROMP_CONV_VARIANT(1, 1, 1, 2, 32, 32),
    ROMP_CONV_VARIANT(1, 1, 4, 1, 32, 32), ROMP_CONV_VARIANT(1, 1, 4, 2, 32, 32),
    ROMP_CONV_VARIANT(1, 1, 1, 1, 16, 32), ROMP_CONV_VARIANT(1, 1, 1, 2, 16, 32),
    ROMP_CONV_VARIANT(1, 1, 2, 2, 16, 32),
    ROMP_CONV_VARIANT(1, 1, 2, 1, 32, 16), ROMP_CONV_VARIANT(1, 1, 2, 2, 32, 16),
    ROMP_CONV_VARIANT(1, 2, 1, 2, 16, 32), ROMP_CONV_VARIANT(1, 2, 1, 1, 16, 32), ROMP_CONV_VARIANT(1, 2, 2, 2, 16, 32),
    ROMP_CONV_VARIANT(2, 1, 1, 2, 16, 16), ROMP_CONV_VARIANT(2, 1, 2, 2, 16, 16), ROMP_CONV_VARIANT(2, 1, 2, 2, 32, 16),
    ROMP_CONV_VARIANT(2, 1, 1, 1, 16, 16),
    ROMP_CONV_VARIANT(13, 1, 2, 2, 32, 16), ROMP_CONV_VARIANT(13, 1, 1, 2, 32, 16), ROMP_CONV_VARIANT(13, 1, 1, 1, 32, 16),
    ROMP_CONV_VARIANT(13, 1, 2, 1, 32, 16), ROMP_CONV_VARIANT(13, 1, 1, 2, 32, 32),
    ROMP_CONV_VARIANT_PP(3, 1, 2, 1, 32, 16), ROMP_CONV_VARIANT_PP(3, 1, 2, 1, 16, 16),
    ROMP_CONV_VARIANT_PP(3, 1, 1, 2, 32, 16), ROMP_CONV_VARIANT_PP(3, 1, 1, 2, 16, 16),
    ROMP_CONV_VARIANT_PP(3, 1, 2, 2, 32, 16), ROMP_CONV_VARIANT_PP(3, 1, 2, 2, 16, 16),
    ROMP_CONV_VARIANT_PP(3, 1, 1, 1, 16, 16), ROMP_CONV_VARIANT_PP(3, 1, 1, 1, 32, 16),
    ROMP_CONV_VARIANT_PP(3, 2, 1, 2, 16, 16), ROMP_CONV_VARIANT_PP(3, 2, 1, 1, 16, 16),
    ROMP_CONV_VARIANT_PP(3, 2, 1, 2, 32, 8), ROMP_CONV_VARIANT_PP(3, 2, 1, 1, 32, 16),
    ROMP_CONV_VARIANT_PP(1, 1, 2, 2, 32, 32), ROMP_CONV_VARIANT_PP(1, 1, 2, 1, 32, 32),
    ROMP_CONV_VARIANT_PP(1, 1, 1, 2, 16, 32), ROMP_CONV_VARIANT_PP(1, 1, 2, 2, 32, 16),
};
ConvVariant* conv_variants_f32(int* n) { *n = (int)(sizeof(kVariantsF32) / sizeof(kVariantsF32[0])); return kVariantsF32; }

}  // namespace romp
