// stem2.hip -- HRNet's whole stem as ONE kernel (round 6; ROMP_OP_STEM2, plan.fuse_stem2; simple_romp/romp/model.py:384-390):
//     m = relu(bn1(conv3x3_s2_{3->64}(x / 255 * 2 - 1)))        512^2 -> 256^2     (the stem: stem_fuse.hip stem_mfma_kernel)
//     y = relu(bn2(conv3x3_s2_{64->64}(m)))                     256^2 -> 128^2     (conv2: conv_h2s.hip as a launch of its own)
// As two launches the 64-channel 256^2 tensor m -- 16.8 MB per image, the largest activation of the network after layer1's -- is
// written (0.18 ms, 3.5 TB/s) and read straight back (0.22 ms, 3.0 TB/s): 1.07 GB of a forward's 25 GB at B = 32, both on the serial
// head of the graph where a saved microsecond is a saved microsecond.  Here m never leaves the CU: per 4 x 16 tile of y a workgroup
//   A. gathers the stem's im2col columns from the normalised 19 x 67 x 3 image halo in LDS (K = 27 padded to ONE 32-wide f16x2 MFMA
//      step, as stem_mfma_kernel) for the 9 x 33 pixels of m the tile needs -- 19 blocks of 16 pixels over the four waves -- applies
//      bn1 + ReLU, zeroes what lies outside the 256^2 map (conv2's padding), splits into fp16 pieces and parks them in LDS in
//      (octet, piece) PLANES with the columns de-interleaved by parity (a stride-2 tap then reads 16 consecutive units);
//   B. runs conv2 from those planes the way conv_h2c.h runs its second conv: wave w owns output channels 16 w .. 16 w + 15 with
//      its 9 x 2 x 2 weight fragments (144 registers) resident for the whole launch, v_mfma_f32_16x16x32_f16, 216 MFMAs per wave
//      and tile; bn2 + ReLU, split, v_permlane16_swap, 16-byte stores of y in the H2 format.
// The stem's halo recompute is (9 x 33) / (8 x 32) = 1.16 x of a layer that is 0.23 of the pair's 1.43 GFLOP per image.  The next
// tile's image halo is fetched into registers under phase B and written to the second halo buffer before the tile's last barrier.
// Both places that form fp16 pieces count their clamps (conv_common.h sat_track_pk: post-ReLU values).
// Measured (B = 32, same box, profiles/r06_stem2_ab.txt): 0.247 ms against 0.178 + 0.226 ms for the two launches; the job 3 055-3 080 ->
// 3 101-3 108 images/s (+1.1 %).  Two re-arrangements were measured and dropped (profiles/r06_stem2_variants_ab.txt): eight waves (two per
// SIMD; conv2's 144 weight registers leave too few of the 256: 48 spilled dwords, 0.345 ms) and two m blocks in flight per wave with
// the stem's operands in LDS (0.268 ms): phase A is not bound by one wave's own latencies.
#include "conv_common.h"
#include "conv_split.h"
#include "conv_fuse.h"
#include <string.h>

namespace romp {

struct Stem2Params {
    const float* image;                                        // (B, H, W, 3) float 0..255
    const float* w1; const float* scale1; const float* shift1; // the stem: [tap 9][cin 3][cout 64] float32, BN scale / shift
    const uint4* w2; const float* scale2; const float* shift2; // conv2: plan.pack_h2_wave16 f16x2 pack, its f16x2 epilogue scale, shift
    float* out;
    int H, W, Hm, Wm, Ho, Wo;                                  // image 512, m 256, y 128
    int out_cs, out_co, out_rs, out_bs;
    float act_scale;
    int tiles_x, tiles_y, tiles_total;
    int* sat;
};

struct S2Cfg {
    static constexpr int TH = 4, TW = 16;                      // output tile
    static constexpr int MR = 2 * TH + 1, MC = 2 * TW + 1;     // 9 x 33 pixels of m
    static constexpr int IR = 2 * MR + 1, IC = 2 * MC + 1;     // 19 x 67 image pixels
    static constexpr int MPC = (MC + 1) / 2;                   // 17 columns per parity
    static constexpr int MROW = 2 * MPC;                       // units per m row of a plane: [parity][col / 2]
    static constexpr int MPL = 320;                            // units per plane: 9 x 34 = 306 used, padded to 0 mod 16
    static constexpr int NPL = 16;                             // planes: 8 octets x {high, low}; plane = 2 * octet + piece
    static constexpr int NPIX = MR * MC;                       // 297
    static constexpr int NBLK = (NPIX + 15) / 16;              // 19 blocks of 16 m pixels
    static constexpr int INF = IR * IC * 3;                    // 3 819 floats of normalised halo
    static constexpr int INP = 3840;                           // padded
    static constexpr int NL = (INF + 255) / 256;               // 15 loads per thread
    static constexpr int OFF_IN = NPL * MPL * 16;              // 81 920
    static constexpr int LDS_BYTES = OFF_IN + 2 * INP * 4;     // 112 640
    static_assert(MR * MROW <= MPL && MPL % 16 == 0, "m planes");
};

typedef float f32x4t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 1) void stem2_kernel(Stem2Params p) {
    using X = S2Cfg;
    using frag = f16x8;
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sM = reinterpret_cast<char*>(smem);
    float* sIn = reinterpret_cast<float*>(sM + X::OFF_IN);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, q = lane >> 4;
    int tile = blockIdx.x;
    if (tile >= p.tiles_total) return;
    auto pack_hi = [&](float a, float c) __attribute__((always_inline)) {
        const f32x2_t v = {a, c};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
    };
    // ---- image halo of a tile: batched, branch-free loads (normalised and scaled by 16 -- the fp16 pieces' scale -- when written)
    float raw[X::NL];
    bool ok[X::NL];
    auto load_halo = [&](int t) __attribute__((always_inline)) {
        const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, b = t / (p.tiles_x * p.tiles_y);
        const float* img = p.image + (size_t)b * p.H * p.W * 3;
        const int iy0 = 4 * ty * X::TH - 3, ix0 = 4 * tx * X::TW - 3;
#pragma unroll
        for (int k = 0; k < X::NL; ++k) {
            const int idx = tid + k * 256;
            const int idc = idx < X::INF ? idx : 0;
            const int e = idc % (X::IC * 3), hy = idc / (X::IC * 3);
            const int iy = iy0 + hy, ix = ix0 + e / 3;
            ok[k] = idx < X::INF && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            raw[k] = img[ok[k] ? ((size_t)iy * p.W + ix0) * 3 + e : 0];
        }
    };
    auto store_halo = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < X::NL; ++k) {
            const int idx = tid + k * 256;                      // zero padding is applied AFTER normalisation
            if (idx < X::INF) sIn[buf * X::INP + idx] = ok[k] ? ((raw[k] / 255.0f) * 2.0f - 1.0f) * 16.0f : 0.f;
        }
    };
    load_halo(tile);

    // ---- the stem's A operands: channel 16 g + px, k = 8 q .. 8 q + 7 (zero beyond 26), 256 w split into fp16 pairs (stem_mfma_kernel)
    int koff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * q + j < 27 ? 8 * q + j : 26;
        const int tap = k / 3, ci = k % 3;
        koff[j] = ((tap / 3) * X::IC + tap % 3) * 3 + ci;
    }
    frag wa[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int k0 = 8 * q + 2 * jj, k1 = k0 + 1;
            const float w0 = k0 < 27 ? h2_sat(p.w1[k0 * 64 + 16 * g + px] * 256.0f) : 0.f;
            const float w1 = k1 < 27 ? h2_sat(p.w1[k1 * 64 + 16 * g + px] * 256.0f) : 0.f;
            hi[jj] = pack_hi(w0, w1);
            lo[jj] = h2_low_pair(hi[jj], w0, w1);
        }
        wa[g][0] = __builtin_bit_cast(frag, make_uint4(hi[0], hi[1], hi[2], hi[3]));
        wa[g][1] = __builtin_bit_cast(frag, make_uint4(lo[0], lo[1], lo[2], lo[3]));
    }
    // bn1 of this lane's channels 16 g + 4 q .. + 3, in the scaled domain ((256 w)(16 x) = 4096 w x)
    const float prod_scale = p.act_scale * (1.0f / 4096.0f);
    f32x4t s1[4], b1[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s1[g][e] = p.scale1[16 * g + 4 * q + e] * prod_scale;
            b1[g][e] = p.shift1[16 * g + 4 * q + e] * p.act_scale;
        }
    // ---- conv2: this wave's weights (output channels 16 wv ..), bn2 of this lane's channels 16 wv + 4 q .. + 3
    frag w2[9][2][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) w2[tap][kc][pc] = __builtin_bit_cast(frag, p.w2[(((wv * 9 + tap) * 2 + kc) * 2 + pc) * 64 + lane]);
    f32x4t s2, b2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s2[e] = p.scale2[16 * wv + 4 * q + e] * p.act_scale;
        b2[e] = p.shift2[16 * wv + 4 * q + e] * p.act_scale;
    }
    // fragment base of conv2: lane (px, q) reads octet 4 kc + q, piece pc of m pixel (2 r + dy, 2 px + dx): plane 8 kc + 2 q + pc,
    // unit (2 r + dy) * 34 + (dx & 1) * 17 + px + (dx >> 1)
    const int mb = (2 * q * X::MPL + px) * 16;
    store_halo(0);
    unsigned sat_pk = 0u;
    __syncthreads();

    int buf = 0;
#pragma unroll 1
    for (; tile < p.tiles_total; tile += gridDim.x) {
        const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
        const int nxt = tile + gridDim.x;
        const bool has_next = nxt < p.tiles_total;
        // ================= A. the stem for the tile's 9 x 33 pixels of m, block after block (wave wv: blocks wv, wv + 4, ..)
        const float* sI = sIn + buf * X::INP;
        const int my0 = 2 * ty * X::TH - 1, mx0 = 2 * tx * X::TW - 1;       // m coordinates of the tile's m pixel (0, 0)
#pragma unroll 1
        for (int blk = wv; blk < X::NBLK; blk += 4) {
            const int i = blk * 16 + px;
            const bool act = i < X::NPIX;
            const int ic = act ? i : X::NPIX - 1;
            const int r = ic / X::MC, c = ic % X::MC;
            const float* base = sI + ((2 * r) * X::IC + 2 * c) * 3;
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = base[koff[j]];
            unsigned hi[4], lo[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                hi[jj] = pack_hi(x[2 * jj], x[2 * jj + 1]);
                lo[jj] = h2_low_pair(hi[jj], x[2 * jj], x[2 * jj + 1]);
            }
            const frag xh = __builtin_bit_cast(frag, make_uint4(hi[0], hi[1], hi[2], hi[3]));
            const frag xl = __builtin_bit_cast(frag, make_uint4(lo[0], lo[1], lo[2], lo[3]));
            f32x4t acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[g] = (f32x4t){0.f, 0.f, 0.f, 0.f};
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][1], xh, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][0], xl, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][0], xh, acc[g], 0, 0, 0);
            }
            // D: lane (px, q) holds channels 16 g + 4 q .. + 3 of ITS OWN pixel: half (q & 1) of octet 2 g + (q >> 1)
            const bool inside = act && (unsigned)(my0 + r) < (unsigned)p.Hm && (unsigned)(mx0 + c) < (unsigned)p.Wm;
            char* dst = sM + ((r * X::MROW + (c & 1) * X::MPC + (c >> 1)) * 16 + (q & 1) * 8);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = h2_sat(fmaxf(fmaf(acc[g][e], s1[g][e], b1[g][e]), 0.f));
                    v[e] = inside ? t : 0.f;
                }
                const unsigned hh[2] = {pack_hi(v[0], v[1]), pack_hi(v[2], v[3])};
                const unsigned hl[2] = {h2_low_pair(hh[0], v[0], v[1]), h2_low_pair(hh[1], v[2], v[3])};
                sat_pk = sat_track_pk(sat_pk, hh[0], hh[1]);
                if (act) {
                    const int o = 2 * g + (q >> 1);
                    *reinterpret_cast<uint2*>(dst + (2 * o) * X::MPL * 16) = make_uint2(hh[0], hh[1]);
                    *reinterpret_cast<uint2*>(dst + (2 * o + 1) * X::MPL * 16) = make_uint2(hl[0], hl[1]);
                }
            }
        }
        if (has_next) load_halo(nxt);                              // lands under phase B
        __syncthreads();                                           // m is complete
        // ================= B. conv2 from the planes: output rows 0..3 x 16 pixels, this wave's 16 channels
        f32x4t acc2[X::TH];
#pragma unroll
        for (int r = 0; r < X::TH; ++r) acc2[r] = (f32x4t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < X::TH; ++r)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap % 3;
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    frag xm[2];
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc)
                        xm[pc] = *reinterpret_cast<const frag*>(sM + mb + ((8 * kc + pc) * X::MPL + (2 * r + dy) * X::MROW + (dx & 1) * X::MPC + (dx >> 1)) * 16);
                    acc2[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[tap][kc][1], xm[0], acc2[r], 0, 0, 0);
                    acc2[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[tap][kc][0], xm[1], acc2[r], 0, 0, 0);
                    acc2[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[tap][kc][0], xm[0], acc2[r], 0, 0, 0);
                }
            }
        // ---- bn2 + ReLU, split; lanes (px, q) and (px, q ^ 1) trade halves so that each stores one whole 16-byte unit
        float* out = p.out + (size_t)b * p.out_bs + p.out_co + 16 * wv + 4 * q;
#pragma unroll
        for (int r = 0; r < X::TH; ++r) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = h2_sat(fmaxf(fmaf(acc2[r][e], s2[e], b2[e]), 0.f));
            const unsigned hh[2] = {pack_hi(v[0], v[1]), pack_hi(v[2], v[3])};
            const unsigned hl[2] = {h2_low_pair(hh[0], v[0], v[1]), h2_low_pair(hh[1], v[2], v[3])};
            sat_pk = sat_track_pk(sat_pk, hh[0], hh[1]);
            const u32x2_t a = __builtin_amdgcn_permlane16_swap(hh[0], hl[0], false, false);
            const u32x2_t c = __builtin_amdgcn_permlane16_swap(hh[1], hl[1], false, false);
            const int oy = ty * X::TH + r, ox = tx * X::TW + px;
            *reinterpret_cast<uint4*>(out + (size_t)oy * p.out_rs + (size_t)ox * p.out_cs) = make_uint4(a[0], c[0], a[1], c[1]);
        }
        if (has_next) store_halo(buf ^ 1);
        buf ^= 1;
        __syncthreads();                                           // every wave is done with m; the next halo is in place
    }
    sat_report_pk(p.sat, sat_pk);
}

// `ops`: the stem op (kind NOP by now, fields intact) and conv2's op (kind ROMP_OP_STEM2): plan.fuse_stem2
int launch_stem2(const romp_op& stem, const romp_op& op, const float* image, float* out, int B, hipStream_t st) {
    ROMP_REQUIRE(stem.Cin == 3 && stem.Cout == 64 && stem.ksize == 3 && stem.stride == 2 && stem.weight && stem.scale && stem.shift,
                 "stem2: the op in front must hold the 3 -> 64 k3 s2 stem");
    ROMP_REQUIRE(op.ksize == 3 && op.stride == 2 && op.Cin == 64 && op.Cout == 64 && op.groups == 1 && op.relu && op.res_buf == ROMP_BUF_NONE &&
                 op.weight_aux && op.scale_h2 && op.shift && (op.flags & ROMP_OPF_WAVE16) && op.out_fmt == ROMP_FMT_H2,
                 "stem2: a 3x3 stride-2 64 -> 64 conv + ReLU with a per-wave f16x2 weight pack and an H2 output expected");
    ROMP_REQUIRE(op.H == stem.H / 2 && op.W == stem.W / 2 && stem.H % 64 == 0 && stem.W % 64 == 0, "stem2: %dx%d image: a multiple of 64 expected", stem.H, stem.W);
    ROMP_REQUIRE(((op.out_cstride | op.out_coff) & 7) == 0 && stem.act_shift == op.act_shift, "stem2: octet-aligned H2 output expected");
    static bool attr = false;
    static int num_cu = 256;
    if (!attr) {
        ROMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, S2Cfg::LDS_BYTES));
        int dev = 0;
        hipDeviceProp_t prop;
        ROMP_HIP_CHECK(hipGetDevice(&dev));
        ROMP_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        attr = true;
    }
    if (image == nullptr && out == nullptr) return ROMP_OK;    // set-up only
    Stem2Params p;
    memset(&p, 0, sizeof(p));
    p.image = image; p.w1 = stem.weight; p.scale1 = stem.scale; p.shift1 = stem.shift;
    p.w2 = reinterpret_cast<const uint4*>(op.weight_aux); p.scale2 = op.scale_h2; p.shift2 = op.shift;
    p.out = out;
    p.H = stem.H; p.W = stem.W; p.Hm = op.H; p.Wm = op.W; p.Ho = op.H / 2; p.Wo = op.W / 2;
    p.out_cs = op.out_cstride; p.out_co = op.out_coff;
    p.out_rs = op.out_rstride > 0 ? op.out_rstride : p.Wo * op.out_cstride;
    p.out_bs = op.out_bstride > 0 ? op.out_bstride : p.Ho * p.Wo * op.out_cstride;
    p.act_scale = ldexpf(1.f, op.act_shift);
    p.tiles_x = p.Wo / S2Cfg::TW; p.tiles_y = p.Ho / S2Cfg::TH; p.tiles_total = B * p.tiles_x * p.tiles_y;
    p.sat = conv_sat_counter();
    const int grid = p.tiles_total < num_cu ? p.tiles_total : num_cu;
    hipLaunchKernelGGL(stem2_kernel, dim3((unsigned)grid), dim3(256), S2Cfg::LDS_BYTES, st, p);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

}  // namespace romp
