// conv_h2x.hip -- the seam between two Bottlenecks of HRNet's layer1 (simple_romp/romp/model.py:103-123) as ONE kernel
// (plan.fuse_bottleneck_seams, ROMP_OP_SEAM1X1; round 3, the tile loop re-done in round 5).
//     t = relu(bn3(conv1x1_{64->256}(m)) + x)        the last conv of Bottleneck i (residual x, 256 channels)
//     u = relu(bn1'(conv1x1_{256->64}(t)))           the first conv of Bottleneck i + 1
// Both are 1x1 convs on 256-channel 128^2 tensors and HBM-bound (4.3-4.5 TB/s as separate launches, 1.7 ms of an 11.8 ms forward
// at B = 32); t has to be written anyway (it is the next block's residual) but need not be READ back: 1.87 -> 1.34 GB per seam.
// No halo (1x1): a tile is 64 consecutive pixels of the flattened batch.  One 256-thread workgroup per CU, one wave per SIMD;
// 16-channel MFMA rows as in conv_h2c.h (v_mfma_f32_16x16x32_f16): GEMM 1, wave w computes channel groups 4 w .. 4 w + 3 of
// t for the tile's four 16-pixel blocks from m in LDS (planes, DMA); its epilogue adds the residual (prefetched into registers a
// tile ahead, 8-byte pieces in the D-operand ownership), writes t to HBM (16-byte units after a permlane16 swap) AND to LDS
// planes; barrier; GEMM 2, wave w computes channel group w of u from t in LDS (K = 256); epilogue; next tile's m by DMA under it.
// ConvParams: in = m, res = x, out = t, out2 = u; w3 / wh = the two convs' weights repacked per 16-channel group
// (plan.pack_h2_wave16); scale / w = conv 1's f16x2 scale and shift (256), scale_h / shift = conv 2's (64).
// Round 5: no store is waited for inside the tile loop.  The first version drained EVERY memory operation at the end of a tile --
// its own u stores, issued a few cycles earlier, included: one exposed store round trip per tile with nothing to compute, and the
// next m asked for only half a tile ahead.  Now the m tile of tile i + 1 is the FIRST memory operation of tile i (a second m buffer
// in LDS), and the tile ends on a COUNTED wait, vmcnt(52): a wave's memory operations retire in issue order and 16 t stores + 32
// residual loads + 4 u stores were issued after the DMA, so "at most 52 outstanding" means the next m has landed while the stores
// and the next residual (used after the next GEMM 1; hipcc counts that wait itself) stay in flight across the barrier.
// (scripts/check_counted_waits.py -- a CPU test runs it -- counts the vector-memory instructions of the COMPILED loop against that 52.)
//
// DS = 1 (round 5, ROMP_OPF_SEAM_DS): the seam behind Bottleneck 0, whose residual is not a tensor but a conv of its own -- the
// `downsample` branch, bn_d(conv1x1_{64->256}(x0)) on the block input x0 (model.py:289-301) -- folded in:
//     t = relu(bn3(conv1x1(m)) + bn_d(conv1x1(x0)))
// The x0 tile arrives by DMA beside m's, GEMM 1 runs both products on two accumulator sets (two BN scales), no residual is
// loaded: the 256-channel downsample output (537 MB at B = 32) is neither written nor read, and its launch (206 us) is gone.
#include "conv_split.h"
#include "conv_fuse.h"
#include <string.h>

namespace romp {

struct XCfg {
    static constexpr int N = 64;                               // pixels per tile
    static constexpr int C0 = 64, C1 = 256, C2 = 64;           // channels of m (and x0), t, u
    static constexpr int MPLN = C0 / 4, TPLN = C1 / 4;         // planes (octet, piece) of 64 units
    static constexpr int MBYTES = MPLN * N * 16;               // 16 384: one m (x0) tile
    static constexpr int OFF_T = MBYTES;
    static constexpr int OFF_M1 = OFF_T + TPLN * N * 16;       // 81 920: the second m buffer
    static constexpr int OFF_X = OFF_M1 + MBYTES;              // 98 304: (DS) the two x0 buffers
    static constexpr int lds_bytes(int ds) { return ds ? OFF_X + 2 * MBYTES : OFF_X; }   // 98 304 / 131 072
    // memory operations a wave issues per tile AFTER the tile DMAs: 16 t stores, (32 residual loads,) 4 u stores
    static constexpr int after_dma(int ds) { return ds ? 16 + 4 : 16 + 32 + 4; }
};

typedef float f32x4x __attribute__((ext_vector_type(4)));

template <int DS, int DRAIN = 0>
__global__ __launch_bounds__(256, 1) void seam1x1_kernel(ConvParams p) {
    conv_args_now(p);
    using X = XCfg;
    using frag = f16x8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sBuf = reinterpret_cast<char*>(smem);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void_f*)sBuf;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, q = lane >> 4;
    const int n_tiles = p.tiles_total;
    const int k0 = blockIdx.x, kstep = gridDim.x;
    if (k0 >= n_tiles) return;

    // ---- weights: GEMM 1 groups 4 wv + gi (K = 64: 2 chunks; DS: the downsample conv's too), GEMM 2 group wv (K = 256: 8 chunks)
    frag w3[4][2][2], wd[DS ? 4 : 1][2][2], w1[8][2];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                w3[gi][kc][pc] = __builtin_bit_cast(frag, p.w3[(((4 * wv + gi) * 2 + kc) * 2 + pc) * 64 + lane]);
                if (DS) wd[gi][kc][pc] = __builtin_bit_cast(frag, p.wx[(((4 * wv + gi) * 2 + kc) * 2 + pc) * 64 + lane]);
            }
#pragma unroll
    for (int kc = 0; kc < 8; ++kc)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) w1[kc][pc] = __builtin_bit_cast(frag, p.wh[((wv * 8 + kc) * 2 + pc) * 64 + lane]);
    // scale / shift of this lane's channels, pre-multiplied by 2^act_shift (DS: b3 = bn3's shift + bn_d's)
    f32x4x s3[4], b3[4], sd[DS ? 4 : 1], s1, b1;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 16 * (4 * wv + gi) + 4 * q + i;
            s3[gi][i] = p.scale[c] * p.act_scale; b3[gi][i] = p.w[c] * p.act_scale;
            if (DS) { sd[gi][i] = p.scale_x[c] * p.act_scale; b3[gi][i] = (p.w[c] + p.shift_x[c]) * p.act_scale; }
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 16 * wv + 4 * q + i;
        s1[i] = p.scale_h[c] * p.act_scale; b1[i] = p.shift[c] * p.act_scale;
    }

    // ---- m (and x0) tile by DMA: plane pl = 4 k + wv (k < 4) is this wave's k-th piece; lane = pixel
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    auto make_rsrc = [&](const float* base_p, unsigned bytes) __attribute__((always_inline)) {
        const unsigned long long base = (unsigned long long)base_p;
        i32x4_t r;
        r[0] = (int)(unsigned)base;
        r[1] = (int)(unsigned)(base >> 32) & 0xffff;
        r[2] = (int)bytes;
        r[3] = 0x00020000;
        return r;
    };
    const i32x4_t rsrc = make_rsrc(p.in + p.in_co, p.in_bytes);
    const i32x4_t rsrc_x = DS ? make_rsrc(p.res + p.res_co, p.res_bytes) : rsrc;
    auto fetch_tile = [&](const i32x4_t& rs, int cs, int tile, int off) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pl = 4 * k + wv;
            const int voff = ((tile * X::N + lane) * cs + (pl >> 1) * 8 + (pl & 1) * 4) * 4;
            const unsigned dst = lds0 + (unsigned)(off + pl * X::N * 16);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rs), "s"(dst) : "memory");
        }
    };
    auto fetch_m = [&](int tile, int mb) __attribute__((always_inline)) {
        fetch_tile(rsrc, p.in_cs, tile, mb ? X::OFF_M1 : 0);
        if (DS) fetch_tile(rsrc_x, p.res_cs, tile, X::OFF_X + mb * X::MBYTES);
    };
    // (DS = 0) the residual of a tile in the D-operand ownership: for channel group gi and pixel block pb the lane's 4 channels' high
    // and low pieces (8 bytes each): octet 2 g + q / 2, half q & 1
    auto load_res = [&](int tile, auto& r) __attribute__((always_inline)) {
#pragma unroll
        for (int gi = 0; gi < 4; ++gi)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const float* a = p.res + (size_t)(tile * X::N + 16 * pb + px) * p.res_cs + p.res_co + (2 * (4 * wv + gi) + (q >> 1)) * 8 + (q & 1) * 2;
                r[gi][pb][0] = *reinterpret_cast<const uint2*>(a);
                r[gi][pb][1] = *reinterpret_cast<const uint2*>(a + 4);
            }
    };
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    auto pack_hi = [&](float a, float b) __attribute__((always_inline)) {
        const f32x2_t v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
    };
    const int xa = (2 * q * X::N + px) * 16;                    // GEMM 1 fragment: + tile buffer + ((8 kc + pc) * 64 + 16 pb) * 16
    const int ta = X::OFF_T + (2 * q * X::N + px) * 16;        // GEMM 2 fragment: + ((8 kc + pc) * 64 + 16 pb) * 16

    float sat_mx = 0.f;                                         // largest value handed to the fp16 split (post-clamp: == H2_MAX iff clamped)
    uint2 res[DS ? 1 : 4][4][2];
    fetch_m(k0, 0);
    if constexpr (!DS) load_res(k0, res);
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            asm volatile("" : "+v"(w3[gi][kc][0]), "+v"(w3[gi][kc][1]));
            if (DS) asm volatile("" : "+v"(wd[gi][kc][0]), "+v"(wd[gi][kc][1]));
        }
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) asm volatile("" : "+v"(w1[kc][0]), "+v"(w1[kc][1]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int mb = 0;
#pragma unroll 1
    for (int tile = k0; tile < n_tiles; tile += kstep) {
        const int nxt = tile + kstep;
        const bool has_next = nxt < n_tiles;
        // ---- the next tile's m (x0), a whole tile ahead: buffer mb ^ 1 was last read by GEMM 1 of the previous tile (every wave is
        // past that tile's barriers).  The OLDEST memory operations of this tile
        if (has_next) fetch_m(nxt, mb ^ 1);
        // ---- GEMM 1: t groups 4 wv .. 4 wv + 3, four pixel blocks (DS: two products, two accumulator sets)
        f32x4x acc[4][4], accd[DS ? 4 : 1][4];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                acc[gi][pb] = (f32x4x){0.f, 0.f, 0.f, 0.f};
                if (DS) accd[gi][pb] = (f32x4x){0.f, 0.f, 0.f, 0.f};
            }
        const char* sM = sBuf + (mb ? X::OFF_M1 : 0);
        const char* sX = sBuf + X::OFF_X + mb * X::MBYTES;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                frag x[2], xx[2];
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    x[pc] = *reinterpret_cast<const frag*>(sM + xa + ((8 * kc + pc) * X::N + 16 * pb) * 16);
                    if (DS) xx[pc] = *reinterpret_cast<const frag*>(sX + xa + ((8 * kc + pc) * X::N + 16 * pb) * 16);
                }
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int gi = 0; gi < 4; ++gi) {
                        acc[gi][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3[gi][kc][pr == 0 ? 1 : 0], x[pr == 1 ? 1 : 0], acc[gi][pb], 0, 0, 0);
                        if (DS) accd[gi][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wd[gi][kc][pr == 0 ? 1 : 0], xx[pr == 1 ? 1 : 0], accd[gi][pb], 0, 0, 0);
                    }
            }
        // ---- epilogue 1: t = relu(bn3 + x) (DS: relu(bn3 + bn_d)), to HBM and to the LDS planes GEMM 2 reads
        float* tout = p.out + (size_t)(tile * X::N + px) * p.out_cs + p.out_co + 4 * q;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int g = 4 * wv + gi;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y = fmaf(acc[gi][pb][e], s3[gi][e], b3[gi][e]);
                    if constexpr (DS) {
                        v[e] = h2_sat(fmaxf(fmaf(accd[gi][pb][e], sd[gi][e], y), 0.f));
                    } else {
                        const unsigned wh = e < 2 ? res[gi][pb][0].x : res[gi][pb][0].y, wl = e < 2 ? res[gi][pb][1].x : res[gi][pb][1].y;
                        v[e] = (e & 1) ? add_pieces_relu<1>(y, wh, wl, H2_MAX) : add_pieces_relu<0>(y, wh, wl, H2_MAX);
                    }
                }
                sat_track(sat_mx, v[0], v[1]);
                sat_track(sat_mx, v[2], v[3]);
                unsigned hh[2] = {pack_hi(v[0], v[1]), pack_hi(v[2], v[3])};
                unsigned hl[2] = {h2_low_pair(hh[0], v[0], v[1]), h2_low_pair(hh[1], v[2], v[3])};
                char* tl = sBuf + X::OFF_T + ((2 * (2 * g + (q >> 1))) * X::N + 16 * pb + px) * 16 + (q & 1) * 8;
                *reinterpret_cast<uint2*>(tl) = make_uint2(hh[0], hh[1]);
                *reinterpret_cast<uint2*>(tl + X::N * 16) = make_uint2(hl[0], hl[1]);
                const u32x2_t a = __builtin_amdgcn_permlane16_swap(hh[0], hl[0], false, false);
                const u32x2_t b = __builtin_amdgcn_permlane16_swap(hh[1], hl[1], false, false);
                *reinterpret_cast<uint4*>(tout + (size_t)(16 * pb) * p.out_cs + 16 * g) = make_uint4(a[0], b[0], a[1], b[1]);
            }
        if constexpr (!DS) { if (has_next) load_res(nxt, res); }               // the next tile's residual: a GEMM 2 + a GEMM 1 to arrive
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // t is complete
        // ---- GEMM 2: u group wv from t (K = 256)
        f32x4x acc2[4];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) acc2[pb] = (f32x4x){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 8; ++kc)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                frag x[2];
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) x[pc] = *reinterpret_cast<const frag*>(sBuf + ta + ((8 * kc + pc) * X::N + 16 * pb) * 16);
                acc2[pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[kc][1], x[0], acc2[pb], 0, 0, 0);
                acc2[pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[kc][0], x[1], acc2[pb], 0, 0, 0);
                acc2[pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[kc][0], x[0], acc2[pb], 0, 0, 0);
            }
        float* uout = p.out2 + (size_t)(tile * X::N + px) * p.out2_cs + p.out2_co + 16 * wv + 4 * q;
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = h2_sat(fmaxf(fmaf(acc2[pb][e], s1[e], b1[e]), 0.f));
            sat_track(sat_mx, v[0], v[1]);
            sat_track(sat_mx, v[2], v[3]);
            unsigned hh[2] = {pack_hi(v[0], v[1]), pack_hi(v[2], v[3])};
            unsigned hl[2] = {h2_low_pair(hh[0], v[0], v[1]), h2_low_pair(hh[1], v[2], v[3])};
            const u32x2_t a = __builtin_amdgcn_permlane16_swap(hh[0], hl[0], false, false);
            const u32x2_t b = __builtin_amdgcn_permlane16_swap(hh[1], hl[1], false, false);
            *reinterpret_cast<uint4*>(uout + (size_t)(16 * pb) * p.out2_cs) = make_uint4(a[0], b[0], a[1], b[1]);
        }
        // ---- the next m (x0) has landed (see the header): everything older than the operations issued after the DMAs has completed.
        // Barrier: every wave's share of it is there, t may be overwritten
        static_assert(X::after_dma(0) == 52 && X::after_dma(1) == 20, "the counted waits below");
        // (DRAIN = 1: the full drain of round 3 -- its own instantiation, chosen by ROMP_CONV_DEBUG=1024, so that a GPU test can compare
        // the counted waits bit for bit against it; a RUN-TIME switch here made hipcc spill the DS = 1 tile loop)
        if (DRAIN) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (DS) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(52) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        mb ^= 1;
    }
    sat_report(p.sat, sat_mx);
}

// `opa`: the 64 -> 256 conv (+ residual + ReLU), `opb`: the 256 -> 64 conv (+ ReLU) reading its output.  `opd` (or nullptr): the
// 64 -> 256 downsample conv whose output WAS opa's residual (ROMP_OPF_SEAM_DS on opb; plan.fuse_bottleneck_seams); `x` is then
// opd's INPUT tensor (64 channels)
int launch_seam1x1(const romp_op& opa, const romp_op& opb, const romp_op* opd, const float* m, const float* x, float* t, float* u, int B, hipStream_t st) {
    ROMP_REQUIRE(opa.ksize == 1 && opa.stride == 1 && opa.Cin == 64 && opa.Cout == 256 && opa.groups == 1 && opa.relu &&
                 opb.ksize == 1 && opb.stride == 1 && opb.Cin == 256 && opb.Cout == 64 && opb.groups == 1 && opb.relu,
                 "seam1x1: a 1x1 64 -> 256 conv + residual + ReLU followed by a 1x1 256 -> 64 conv + ReLU expected");
    ROMP_REQUIRE(opa.weight_aux && opa.scale_h2 && opb.weight_aux && opb.scale_h2 && (opa.flags & opb.flags & ROMP_OPF_WAVE16),
                 "seam1x1: per-group f16x2 weight packs (ROMP_OPF_WAVE16) expected");
    ROMP_REQUIRE(opa.in_fmt == ROMP_FMT_H2 && opa.res_fmt == ROMP_FMT_H2 && opa.out_fmt == ROMP_FMT_H2 && opb.in_fmt == ROMP_FMT_H2 &&
                 opb.out_fmt == ROMP_FMT_H2 && opa.act_shift == opb.act_shift, "seam1x1: H2 tensors expected");
    ROMP_REQUIRE(((long)B * opa.H * opa.W) % XCfg::N == 0 && opa.H == opb.H && opa.W == opb.W, "seam1x1: pixel count not a multiple of 64");
    ROMP_REQUIRE(((opa.in_cstride | opa.in_coff | opa.res_cstride | opa.res_coff | opa.out_cstride | opa.out_coff | opb.out_cstride | opb.out_coff) & 7) == 0 &&
                 opa.out_rstride == 0 && opa.out_bstride == 0 && opb.out_rstride == 0 && opb.out_bstride == 0, "seam1x1: dense, octet-aligned tensors expected");
    if (opd)
        ROMP_REQUIRE(opd->ksize == 1 && opd->stride == 1 && opd->Cin == 64 && opd->Cout == 256 && opd->groups == 1 && !opd->relu && opd->res_buf < 0 &&
                     opd->weight_aux && opd->scale_h2 && (opd->flags & ROMP_OPF_WAVE16) && opd->in_fmt == ROMP_FMT_H2 && opd->act_shift == opa.act_shift &&
                     opd->H == opa.H && opd->W == opa.W && ((opd->in_cstride | opd->in_coff) & 7) == 0 && opd->out_buf == opa.res_buf,
                     "seam1x1: the folded downsample must be a plain 1x1 64 -> 256 conv on an H2 tensor whose output was the residual");
    static bool attr = false;
    static int num_cu = 256;
    if (!attr) {
        ROMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(seam1x1_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, XCfg::lds_bytes(0)));
        ROMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(seam1x1_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, XCfg::lds_bytes(1)));
        ROMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(seam1x1_kernel<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, XCfg::lds_bytes(0)));
        ROMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(seam1x1_kernel<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, XCfg::lds_bytes(1)));
        int dev = 0;
        hipDeviceProp_t prop;
        ROMP_HIP_CHECK(hipGetDevice(&dev));
        ROMP_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        attr = true;
    }
    if (m == nullptr && t == nullptr) return ROMP_OK;          // set-up only
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.in = m; p.res = x; p.out = t; p.out2 = u;
    p.w3 = reinterpret_cast<const uint4*>(opa.weight_aux);
    p.wh = reinterpret_cast<const uint4*>(opb.weight_aux);
    p.scale = opa.scale_h2; p.w = opa.shift;
    p.scale_h = opb.scale_h2; p.shift = opb.shift;
    p.act_scale = ldexpf(1.f, opa.act_shift);
    p.sat = conv_sat_counter();
    { const char* e = getenv("ROMP_CONV_DEBUG"); p.dbg = e ? (atoi(e) & 1024) : 0; }      // 1024: the full-drain builds (tests)
    {
        const unsigned long long bytes = ((unsigned long long)B * opa.H * opa.W * opa.in_cstride - opa.in_coff) * 4ull;
        ROMP_REQUIRE(bytes < 0x80000000ull, "seam1x1: input tensor of %llu bytes: beyond the 31-bit offsets of the DMA", bytes);
        p.in_bytes = (unsigned)bytes;
    }
    p.in_cs = opa.in_cstride; p.in_co = opa.in_coff;
    p.res_cs = opa.res_cstride; p.res_co = opa.res_coff;
    if (opd) {
        p.wx = reinterpret_cast<const uint4*>(opd->weight_aux);
        p.scale_x = opd->scale_h2; p.shift_x = opd->shift;
        p.res_cs = opd->in_cstride; p.res_co = opd->in_coff;
        const unsigned long long bytes = ((unsigned long long)B * opd->H * opd->W * opd->in_cstride - opd->in_coff) * 4ull;
        ROMP_REQUIRE(bytes < 0x80000000ull, "seam1x1: downsample input tensor of %llu bytes: beyond the 31-bit offsets of the DMA", bytes);
        p.res_bytes = (unsigned)bytes;
    }
    p.out_cs = opa.out_cstride; p.out_co = opa.out_coff;
    p.out2_cs = opb.out_cstride; p.out2_co = opb.out_coff;
    p.tiles_total = (int)(((long)B * opa.H * opa.W) / XCfg::N);
    long grid = num_cu;
    if (grid > p.tiles_total) grid = p.tiles_total;
    if (p.dbg & 1024) {
        if (opd) hipLaunchKernelGGL((seam1x1_kernel<1, 1>), dim3((unsigned)grid), dim3(256), XCfg::lds_bytes(1), st, p);
        else hipLaunchKernelGGL((seam1x1_kernel<0, 1>), dim3((unsigned)grid), dim3(256), XCfg::lds_bytes(0), st, p);
    } else if (opd) hipLaunchKernelGGL(seam1x1_kernel<1>, dim3((unsigned)grid), dim3(256), XCfg::lds_bytes(1), st, p);
    else hipLaunchKernelGGL(seam1x1_kernel<0>, dim3((unsigned)grid), dim3(256), XCfg::lds_bytes(0), st, p);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

}  // namespace romp
