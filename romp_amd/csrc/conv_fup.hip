// conv_fup.hip -- a fuse-layer OUTPUT of an HRNet module (simple_romp/romp/model.py:233-244) in one kernel ("fuseup", round 4):
//
//     y_i = relu( sum_{j <= i} T_ij  +  sum_{j > i} nearest_up_{2^(j-i)}( bn_ij( W_ij . x_j ) ) )
//
// T_ij: tensors already at output i's resolution (the branch output x_i and the stride-2 chains' results); the second sum: the 1x1
// convs from the lower-resolution branches (model.py:186-196).  Rounds 1-3 ran every W_ij . x_j as its own conv launch (30 per forward,
// 15-20 us each for 0.03-0.13 GFLOP: all fixed cost) writing a small tensor that fusesum_kernel read back.  Here the workgroup that sums
// an output tile computes the up-terms of that tile itself: the tile's footprint in x_j (8 x 64 output pixels <-> 4 x 32, 2 x 16, 1 x 8
// source pixels) arrives by LDS-DMA in the planar H2 layout of conv_h2x.hip, `v_mfma_f32_16x16x32_f16` with the three f16x2 products
// and REGISTER-RESIDENT weights (the whole of W_ij for the wave's 16-channel group(s): 96-128 registers, loaded once per persistent
// workgroup) gives bn(W x) for the footprint, the results wait in LDS as float32, and the sum phase -- the old fusesum_kernel: one
// (pixel, channel octet) unit per thread and step -- reads them with the upsample folded into the index.  30 conv launches and their
// 60 tensor round trips disappear; the source tiles are read exactly once (footprints of different tiles are disjoint).
//
// Shapes are HRNet's: output channels CO in {32, 64, 128}; up-term k (k = 0 .. NUP-1) comes from the branch k + 1 levels down:
// CO << (k + 1) channels at 1 / 2^(k+1) of the resolution.  Tile = TH x TW output pixels (8 x 64, 8 x 32, 4 x 32: 64 KB of output each).
#include "conv_split.h"
#include "conv_fuse.h"
#include <string.h>

namespace romp {

// geometry helpers (free functions: a class's own constexpr members cannot feed its static data members)
constexpr int fup_th(int co) { return co == 128 ? 4 : 8; }
constexpr int fup_tw(int co) { return co == 32 ? 64 : 32; }
constexpr int fup_cin(int co, int k) { return co << (k + 1); }
constexpr int fup_lh(int co, int k) { return fup_th(co) >> (k + 1); }
constexpr int fup_lw(int co, int k) { return fup_tw(co) >> (k + 1); }
constexpr int fup_npl(int co, int k) { return fup_lh(co, k) * fup_lw(co, k); }                        // source pixels of a tile
constexpr int fup_npp(int co, int k) { return fup_npl(co, k) < 16 ? 16 : fup_npl(co, k); }            // ... padded to whole 16-pixel MFMA blocks
constexpr int fup_src_bytes(int co, int k) { return fup_npp(co, k) * fup_cin(co, k) * 4; }
constexpr int fup_res_off(int co, int k) { return k == 0 ? 0 : fup_res_off(co, k - 1) + fup_npp(co, k - 1) * (co + 4); }   // floats
constexpr int fup_max(int a, int b) { return a > b ? a : b; }

template <int CO, int NUP>
struct UCfg {
    static constexpr int TH = fup_th(CO), TW = fup_tw(CO);
    static constexpr int NPX = TH * TW;
    static constexpr int O8 = CO / 8;                          // channel octets of the output
    static constexpr int UNITS = NPX * O8;                     // (pixel, octet) units per tile: 2048
    static constexpr int GW = CO >= 64 ? CO / 64 : 1;          // 16-channel groups per wave (CO = 32: two waves share a group, halving the blocks)
    static constexpr int BSTEP = CO >= 64 ? 1 : 2;             // stride of a wave's 16-pixel blocks
    static constexpr int RS = CO + 4;                          // floats per row of the result tiles (16-byte rows, conflict-free b128 columns)
    static constexpr int KMAX = fup_cin(CO, NUP - 1) / 32;     // 32-channel chunks of the widest up-term
    static constexpr int MAXD = CO == 32 ? 1 : CO == 64 ? 2 : 3;   // direct terms: output i of an HRNet module has i chains + the branch itself
    // LDS: source buffer A (terms 0 and 2), source buffer B (term 1), the result tiles
    static constexpr int BUF_A = fup_max(fup_src_bytes(CO, 0), NUP > 2 ? fup_src_bytes(CO, 2) : 0);
    static constexpr int BUF_B = NUP > 1 ? fup_src_bytes(CO, 1) : 0;
    static constexpr int OFF_R = BUF_A + BUF_B;
    static constexpr int LDS_BYTES = OFF_R + fup_res_off(CO, NUP) * 4 + 64;
    static constexpr int NKC = (fup_cin(CO, 0) + (NUP > 1 ? fup_cin(CO, 1) : 0) + (NUP > 2 ? fup_cin(CO, 2) : 0)) / 32;   // chunks over all up-terms
    static_assert(CO == 32 || CO == 64 || CO == 128, "HRNet branch widths");
    static_assert(NUP >= 1 && NUP <= 3 && fup_lh(CO, NUP - 1) >= 1, "an up-term's footprint is at least one source row");
    static_assert(UNITS % 256 == 0, "whole passes of the sum phase");
};

struct FupParams {
    const float* dt[3]; int d_cs[3];       // direct terms (H2; pointer includes the channel offset), at most 3, summed first and in order
    int n_direct;
    const float* ux[3]; int u_cs[3];       // up sources (H2, dense channels), NUP of them
    const uint4* uw;                       // their weights: per term the wave16 pack [group CO/16][kc cin/32][piece 2][lane 64] 16-byte units, concatenated
    const float* us; const float* ub;      // f16x2 epilogue scale and shift, [term][CO]
    float* out; int out_cs, out_co;
    int H, W;                              // output spatial size
    int relu; float act_scale;
    int tiles_x, tiles_y, tiles_total;
    int* sat;
};

typedef float f32x4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_u;
typedef const __attribute__((address_space(1))) void glb_void_u;

template <int CO, int NUP>
__global__ __launch_bounds__(256, 2) void fuseup_kernel(FupParams p) {
    using X = UCfg<CO, NUP>;
    using frag = f16x8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sBuf = reinterpret_cast<char*>(smem);
    float* sR = reinterpret_cast<float*>(sBuf + X::OFF_R);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, q = lane >> 4;                   // MFMA B / D operand: pixel px of a block; A: channel px of the group; k-quarter / channel quad q
    const int g0 = CO >= 64 ? wv * X::GW : (wv & 1);           // this wave's first 16-channel group
    const int b0 = CO >= 64 ? 0 : (wv >> 1);                   // ... and first pixel block

    // ---- this wave's weights, all up-terms: [term][group of the wave][kc][piece], loaded once per workgroup
    frag wreg[X::NKC * X::GW][2];
    {
        int at = 0;                                            // running 16-byte-unit offset of a term's pack
#pragma unroll
        for (int k = 0, r = 0; k < NUP; ++k) {
            const int nkc = fup_cin(CO, k) / 32;
#pragma unroll
            for (int gi = 0; gi < X::GW; ++gi)
#pragma unroll
                for (int kc = 0; kc < X::KMAX; ++kc)     // (upper bound of the largest term; the guard below folds at compile time)
                    if (kc < nkc) {
#pragma unroll
                        for (int pc = 0; pc < 2; ++pc)
                            wreg[r][pc] = __builtin_bit_cast(frag, p.uw[at + ((((g0 + gi) * nkc + kc) * 2 + pc) * 64) + lane]);
                        ++r;
                    }
            at += (CO / 16) * nkc * 2 * 64;
        }
    }
    // ---- DMA of up-term k's footprint of tile (b, ty, tx) into `buf`: LDS unit (plane, pixel) = plane * npp + pixel, plane = 2 * octet + piece
    auto fetch_src = [&](int k, int b, int ty, int tx, char* buf) __attribute__((always_inline)) {
        const int npp = fup_npp(CO, k), npl = fup_npl(CO, k), lw = fup_lw(CO, k), cin = fup_cin(CO, k);
        const int Hl = p.H >> (k + 1), Wl = p.W >> (k + 1);
        const float* src = p.ux[k] + ((size_t)(b * Hl + ty * fup_lh(CO, k)) * Wl + tx * lw) * p.u_cs[k];
        const int pieces = npp * (cin / 4) / 64;               // wave-instructions of 64 units
#pragma unroll 2
        for (int i = wv; i < pieces; i += 4) {
            const int U = i * 64 + lane;
            const int plane = U / npp, pix = U % npp;          // (powers of two)
            const int pv = pix < npl ? pix : 0;                // padded block lanes re-read pixel 0 (their results are never used)
            const float* a = src + (size_t)((pv / lw) * Wl + pv % lw) * p.u_cs[k] + (plane >> 1) * 8 + (plane & 1) * 4;
            __builtin_amdgcn_global_load_lds((glb_void_u*)a, (lds_void_u*)(buf + i * 1024), 16, 0, 0);
        }
    };
    auto tile_of = [&](int t, int& b, int& ty, int& tx) __attribute__((always_inline)) {
        tx = t % p.tiles_x; t /= p.tiles_x;
        ty = t % p.tiles_y;
        b = t / p.tiles_y;
    };
    char* bufA = sBuf;
    char* bufB = sBuf + X::BUF_A;

    int t = blockIdx.x;
    if (t >= p.tiles_total) return;
    int b, ty, tx;
    tile_of(t, b, ty, tx);
    fetch_src(0, b, ty, tx, bufA);
    if (NUP > 1) fetch_src(1, b, ty, tx, bufB);
    float sat_mx = 0.f;

#pragma unroll 1
    for (;;) {
        // ================= the up-terms: bn(W x) for the tile's footprint -> result tiles in LDS (float32, scaled domain)
#pragma unroll
        for (int k = 0, wr = 0; k < NUP; ++k) {
            const char* buf = (k & 1) ? bufB : bufA;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's DMA pieces (and the previous tile's stores) are done
            __builtin_amdgcn_s_barrier();                      // every wave's pieces of term k are in LDS
            constexpr int MAXB = 8;
            const int nkc = fup_cin(CO, k) / 32, npp = fup_npp(CO, k), nblk = (fup_npp(CO, k) / 16);
            f32x4u acc[X::GW][MAXB / X::BSTEP];
#pragma unroll
            for (int gi = 0; gi < X::GW; ++gi)
#pragma unroll
                for (int j = 0; j < MAXB / X::BSTEP; ++j) acc[gi][j] = (f32x4u){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < X::KMAX; ++kc)
                if (kc < nkc) {
#pragma unroll
                    for (int j = 0; j < MAXB / X::BSTEP; ++j) {
                        const int blk = b0 + j * X::BSTEP;      // (b0 is run time for CO = 32: a wave-uniform guard)
                        if (j * X::BSTEP < nblk && blk < nblk) {
                            frag x[2];
#pragma unroll
                            for (int pc = 0; pc < 2; ++pc)
                                x[pc] = *reinterpret_cast<const frag*>(buf + (((2 * (4 * kc + q) + pc) * npp) + 16 * blk + px) * 16);
#pragma unroll
                            for (int gi = 0; gi < X::GW; ++gi) {
                                const frag (&w)[2] = wreg[wr + gi * nkc + kc];
                                acc[gi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[1], x[0], acc[gi][j], 0, 0, 0);
                                acc[gi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], x[1], acc[gi][j], 0, 0, 0);
                                acc[gi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], x[0], acc[gi][j], 0, 0, 0);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);          // (keeps the fragment reads of later chunks from being hoisted: registers)
                }
            // BN in the scaled domain (the H2 terms of the sum are x * 2^act_shift), to the result tile: lane (px, q) holds channels
            // 16 g + 4 q .. + 3 of pixel 16 blk + px
#pragma unroll
            for (int gi = 0; gi < X::GW; ++gi) {
                const int c = 16 * (g0 + gi) + 4 * q;
                const float4 sc = *reinterpret_cast<const float4*>(p.us + k * CO + c), sh = *reinterpret_cast<const float4*>(p.ub + k * CO + c);
#pragma unroll
                for (int j = 0; j < MAXB / X::BSTEP; ++j) {
                    const int blk = b0 + j * X::BSTEP;
                    if (j * X::BSTEP < nblk && blk < nblk) {
                        f32x4u v;
                        v[0] = fmaf(acc[gi][j][0], sc.x * p.act_scale, sh.x * p.act_scale);
                        v[1] = fmaf(acc[gi][j][1], sc.y * p.act_scale, sh.y * p.act_scale);
                        v[2] = fmaf(acc[gi][j][2], sc.z * p.act_scale, sh.z * p.act_scale);
                        v[3] = fmaf(acc[gi][j][3], sc.w * p.act_scale, sh.w * p.act_scale);
                        *reinterpret_cast<f32x4u*>(sR + fup_res_off(CO, k) + (16 * blk + px) * X::RS + c) = v;
                    }
                }
            }
            wr += X::GW * nkc;
            if (k + 2 < NUP) {                                  // term k + 2 takes over this term's source buffer
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                  // every wave is done reading it
                fetch_src(k + 2, b, ty, tx, (k & 1) ? bufB : bufA);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // result tiles complete; both source buffers are free
        // the next tile's first footprints fly under this tile's sum phase
        const int tn = t + gridDim.x;
        const bool has_next = tn < p.tiles_total;
        int bn = b, tyn = ty, txn = tx;
        if (has_next) {
            tile_of(tn, bn, tyn, txn);
            fetch_src(0, bn, tyn, txn, bufA);
            if (NUP > 1) fetch_src(1, bn, tyn, txn, bufB);
        }
        // ================= the sum: one (pixel, octet) unit per thread and pass, terms in the reference's order (j ascending)
        const size_t pix0 = ((size_t)b * p.H + ty * X::TH) * p.W + tx * X::TW;
#pragma unroll 1                                          // (two units in flight spill in the 3-term instantiation: 112 weight registers stay live)
        for (int u = tid; u < X::UNITS; u += 256) {
            const int o = u % X::O8, pl = u / X::O8;
            const int ly = pl / X::TW, lx = pl % X::TW;
            const size_t pix = pix0 + (size_t)ly * p.W + lx;
            float4 va, vb;
#pragma unroll
            for (int d = 0; d < X::MAXD; ++d)
                if (d < p.n_direct) {
                    const float* a = p.dt[d] + pix * p.d_cs[d] + o * 8;
                    const uint4 hi = *reinterpret_cast<const uint4*>(a), lo = *reinterpret_cast<const uint4*>(a + 4);
                    const float4 ta = h2_unpack(make_uint2(hi.x, hi.y), make_uint2(lo.x, lo.y), 1.f);
                    const float4 tb = h2_unpack(make_uint2(hi.z, hi.w), make_uint2(lo.z, lo.w), 1.f);
                    if (d == 0) { va = ta; vb = tb; }
                    else {
                        va.x += ta.x; va.y += ta.y; va.z += ta.z; va.w += ta.w;
                        vb.x += tb.x; vb.y += tb.y; vb.z += tb.z; vb.w += tb.w;
                    }
                }
#pragma unroll
            for (int k = 0; k < NUP; ++k) {
                const float* r = sR + fup_res_off(CO, k) + ((ly >> (k + 1)) * fup_lw(CO, k) + (lx >> (k + 1))) * X::RS + o * 8;
                const float4 ta = *reinterpret_cast<const float4*>(r), tb = *reinterpret_cast<const float4*>(r + 4);
                va.x += ta.x; va.y += ta.y; va.z += ta.z; va.w += ta.w;
                vb.x += tb.x; vb.y += tb.y; vb.z += tb.z; vb.w += tb.w;
            }
            if (p.relu) {
                va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f); va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
                vb.x = fmaxf(vb.x, 0.f); vb.y = fmaxf(vb.y, 0.f); vb.z = fmaxf(vb.z, 0.f); vb.w = fmaxf(vb.w, 0.f);
            }
            uint2 ha, la, hb, lb;
            h2_pack(va, 1.f, ha, la, sat_mx);
            h2_pack(vb, 1.f, hb, lb, sat_mx);
            float* op_ = p.out + pix * p.out_cs + p.out_co + o * 8;
            *reinterpret_cast<uint4*>(op_) = make_uint4(ha.x, ha.y, hb.x, hb.y);
            *reinterpret_cast<uint4*>(op_ + 4) = make_uint4(la.x, la.y, lb.x, lb.y);
        }
        if (!has_next) break;
        t = tn; b = bn; ty = tyn; tx = txn;
    }
    sat_report(p.sat, sat_mx);
}

template <int CO, int NUP>
static int launch_fup(FupParams& p, int B, hipStream_t st) {
    using X = UCfg<CO, NUP>;
    static bool attr = false;
    static int num_cu = 256;
    if (!attr) {                                               // (romp_net_create calls this path's set-up outside any stream capture)
        ROMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fuseup_kernel<CO, NUP>), hipFuncAttributeMaxDynamicSharedMemorySize, X::LDS_BYTES));
        int dev = 0;
        hipDeviceProp_t prop;
        ROMP_HIP_CHECK(hipGetDevice(&dev));
        ROMP_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        attr = true;
    }
    if (p.out == nullptr) return ROMP_OK;                      // set-up only
    ROMP_REQUIRE(p.n_direct <= X::MAXD, "fuseup: %d direct terms for a %d-channel output (at most %d)", p.n_direct, CO, X::MAXD);
    ROMP_REQUIRE(p.H % X::TH == 0 && p.W % X::TW == 0, "fuseup: %dx%d is not a multiple of the %dx%d tile", p.H, p.W, X::TH, X::TW);
    p.tiles_x = p.W / X::TW; p.tiles_y = p.H / X::TH; p.tiles_total = B * p.tiles_x * p.tiles_y;
    const int cap = conv_wg_cap();
    const int per_cu = (cap == 1 || 160 * 1024 / X::LDS_BYTES < 2) ? 1 : 2;
    long grid = (long)num_cu * per_cu;
    if (grid > p.tiles_total) grid = p.tiles_total;
    hipLaunchKernelGGL((fuseup_kernel<CO, NUP>), dim3((unsigned)grid), dim3(256), X::LDS_BYTES, st, p);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

// `op`: ROMP_OP_FUSEUP (include/romp_hip.h).  terms[k]: the resolved tensors in op order -- the direct terms (term_shift 0) first, then
// the up sources (term_shift 1, 2, ..).  terms == nullptr: one-time set-up of the kernel this op needs.
int launch_fuseup(const romp_op& op, const FuseTerm* terms, float* out, int B, hipStream_t st) {
    int n_direct = 0, n_up = 0;
    for (int k = 0; k < op.n_terms; ++k) {
        if (op.term_shift[k] == 0) { ROMP_REQUIRE(n_up == 0, "fuseup: direct terms come first"); ++n_direct; }
        else { ROMP_REQUIRE(op.term_shift[k] == n_up + 1, "fuseup: up-term %d must come from the branch %d levels down", n_up, n_up + 1); ++n_up; }
    }
    ROMP_REQUIRE(n_direct >= 1 && n_direct <= 3 && n_up >= 1 && n_up <= 3, "fuseup: %d direct + %d up terms", n_direct, n_up);
    ROMP_REQUIRE(op.weight_aux && op.scale_h2 && op.shift && (op.flags & ROMP_OPF_WAVE16), "fuseup: per-group f16x2 weight packs expected");
    ROMP_REQUIRE(op.out_fmt == ROMP_FMT_H2 && ((op.out_cstride | op.out_coff) & 7) == 0, "fuseup: H2 output expected");
    FupParams p;
    memset(&p, 0, sizeof(p));
    p.n_direct = n_direct;
    for (int k = 0; k < op.n_terms; ++k) {
        ROMP_REQUIRE(op.term_fmt[k] == ROMP_FMT_H2 && ((op.term_cstride[k] | op.term_coff[k]) & 7) == 0, "fuseup: term %d must be an octet-aligned H2 tensor", k);
        if (k < n_direct) { p.dt[k] = terms ? terms[k].ptr : nullptr; p.d_cs[k] = op.term_cstride[k]; }
        else {
            const int u = k - n_direct;
            ROMP_REQUIRE(op.term_cstride[k] == (op.Cout << (u + 1)) && op.term_coff[k] == 0, "fuseup: up source %d must be a dense %d-channel tensor", u, op.Cout << (u + 1));
            p.ux[u] = terms ? terms[k].ptr : nullptr; p.u_cs[u] = op.term_cstride[k];
        }
    }
    p.uw = reinterpret_cast<const uint4*>(op.weight_aux); p.us = op.scale_h2; p.ub = op.shift;
    p.out = out; p.out_cs = op.out_cstride; p.out_co = op.out_coff;
    p.H = op.H; p.W = op.W; p.relu = op.relu; p.act_scale = ldexpf(1.f, op.act_shift);
    p.sat = conv_sat_counter();
#define ROMP_FUP(CO_, NUP_) if (op.Cout == CO_ && n_up == NUP_) return launch_fup<CO_, NUP_>(p, B, st)
    ROMP_FUP(32, 1); ROMP_FUP(32, 2); ROMP_FUP(32, 3); ROMP_FUP(64, 1); ROMP_FUP(64, 2); ROMP_FUP(128, 1);
#undef ROMP_FUP
    ROMP_REQUIRE(false, "fuseup: no kernel for %d channels with %d up-terms", op.Cout, n_up);
}

}  // namespace romp
