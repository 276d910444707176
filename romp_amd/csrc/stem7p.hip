// stem7p.hip -- ResNet-50's whole stem as ONE kernel on the matrix cores (round 6; ROMP_OP_STEM7P, resnet_plan._stem7p;
// romp/lib/models/resnet_50.py:32-45,56):
//     m = relu(bn1(conv7x7_s2_p3_{3->64}((x / 255 - mean) / std)))        512^2 -> 256^2
//     y = maxpool3x3_s2_p1(m)                                             256^2 -> 128^2
// As two launches (stem_fuse.hip stem7_conv_kernel + maxpool3s2_kernel) the float32 VALU conv ran at 78 TFLOP/s (0.54 ms at B = 32:
// 9 408 FMAs per thread) and the 256^2 x 64 tensor m -- 16.8 MB per image -- was written and read straight back (0.15 ms more):
// 0.69 ms of an 8.3-ms call.  Here:
//   * K = 147 = 7 rows x 21 (dx, colour) values runs as FIVE 32-wide f16x2 MFMA steps (v_mfma_f32_16x16x32_f16, the three piece
//     products hi*lo + lo*hi + hi*hi, float32 accumulate -- every other conv's arithmetic): a kernel row is padded to 22 so that an fp16
//     PAIR never straddles two rows, k' = 22 dy + 3 dx + c, 154 of 160 slots used (zero weights elsewhere);
//   * the normalised image halo of a tile sits in LDS ALREADY SPLIT into the high and low fp16 pieces of 16 x, in 8-byte units of two
//     values (hi0 hi1 lo0 lo1), rows 118 values apart: the im2col gather of a pixel is 20 aligned ds_read_b64 -- each the high AND the
//     low pair of one MFMA operand dword -- and no conversion work (a pixel's taps start 6 values = 24 bytes behind its neighbour's);
//   * wave g owns output channels 16 g .. 16 g + 15 (A operand: 5 steps x 2 pieces = 40 registers, formed once per workgroup from the
//     float32 weights x 256) and walks ALL pixels of the tile's m region, two 16-pixel blocks in flight (independent accumulators);
//   * a tile is 4 x 8 pooled pixels = 9 x 17 pixels of m (153 = 10 blocks; 1.25 x recompute for the pool's halo): BN + ReLU in float32,
//     zero outside the 256^2 map (the pool pads with -inf; everything valid is >= 0 after the ReLU and every window holds a valid pixel),
//     parked per wave in LDS, pooled by the same wave (no workgroup barrier: a wave's LDS operations execute in order), stored as
//     float32 or H2;
//   * persistent workgroups (two per CU), the next tile's raw image values fetched into registers under the MFMAs and normalised /
//     split / written to the second halo buffer before the tile's one barrier.
// The float32 VALU pair stays for the float32 / calibration programs and whenever 256 |w| would leave the fp16 pieces (plan side).
// Measured (B = 32, per-op HIP events, profiles/r06s_stem7p*_ab.txt): 0.53 + 0.15 ms as two launches -> 0.315 ms (first form: piece planes,
// 40 ds_read_b32 per block) -> 0.278 ms (this form); 141-145 TFLOP/s algorithmic.  The kernel is latency-bound at two waves per SIMD
// (~1 500 instructions per wave and tile in ~18 000 clocks): neither the DS instruction count (halved: -12 %) nor the VALU count
// (-40 %: reciprocal normalisation, per-thread slot tables) nor unrolling the block loop moved it further.
#include "conv_common.h"
#include "conv_split.h"
#include "conv_fuse.h"
#include <string.h>

namespace romp {

struct Stem7pParams {
    const float* image;                                        // (B, H, W, 3) float 0..255
    const float* w; const float* scale; const float* shift;    // [ky 7][kx 7][cin 3][cout 64] float32, BN scale / shift
    float* out;
    int H, W, Hm, Wm, Ho, Wo;                                  // image 512, m 256, y 128
    int out_cs, out_co;
    int out_h2; float act_scale;
    int tiles_x, tiles_y, tiles_total;
    int* sat;
};

struct S7Cfg {
    static constexpr int TH = 4, TW = 8;                       // pooled tile
    static constexpr int MR = 2 * TH + 1, MC = 2 * TW + 1;     // 9 x 17 pixels of m
    static constexpr int NPIX = MR * MC;                       // 153
    static constexpr int NBLK = (NPIX + 15) / 16;              // 10 blocks of 16
    static constexpr int IR = 2 * MR + 5, IC = 2 * MC + 5;     // 23 x 39 image pixels
    static constexpr int RS = 118;                             // halves per halo row: 117 values + 1 (even: aligned pairs)
    static constexpr int PLANE = (IR + 1) * RS + 8;            // values of a halo buffer: a zero row behind the last (k' rows 7 of the bottom pixels) + slack
    static constexpr int BUF_BYTES = PLANE * 4;                // a value = its high and its low fp16 piece, in 8-byte units of two values: hi0 hi1 lo0 lo1
    static constexpr int NVAL = IR * IC * 3;                   // 2 691 values per tile
    static constexpr int NL = (NVAL + 255) / 256;              // 11 loads per thread
    static constexpr int KROW = 22, NSTEP = 5;                 // k' = 22 dy + 3 dx + c; 5 x 32 >= 7 x 22
    static constexpr int OFF_P = 2 * BUF_BYTES;                // two halo buffers
    static constexpr int PARK = NBLK * 16 * 16 * 4;            // per wave: 160 pixels x 16 channels float32
    static constexpr int LDS_BYTES = OFF_P + 4 * PARK;         // 22 720 + 40 960
    static_assert(OFF_P % 16 == 0 && PLANE % 2 == 0, "alignment");
    static_assert(NBLK % 2 == 0, "blocks go in pairs");
};

typedef float f32x4p __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void stem7p_kernel(Stem7pParams p) {
    using X = S7Cfg;
    using frag = f16x8;
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sBuf = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, q = lane >> 4;
    float* sP = reinterpret_cast<float*>(sBuf + X::OFF_P + wave * X::PARK);
    auto pack_hi = [&](float a, float c) __attribute__((always_inline)) {
        const f32x2_t v = {a, c};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
    };

    // ---- zero both halo buffers once: the pad half of every row, the pad row and the slack are read (against zero weights) and
    // must hold finite values
    for (int i = tid; i < X::OFF_P / 16; i += 256) reinterpret_cast<uint4*>(sBuf)[i] = make_uint4(0u, 0u, 0u, 0u);

    // ---- the halo fetch: value idx of a tile = (halo row hy, e = 3 column + colour); raw values of the next tile wait in registers
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    float raw[X::NL];
    unsigned okm = 0u;
    int slot[X::NL];                                           // this thread's values of a tile, the same for every tile: halo row | column << 5 | colour << 11 | LDS half index << 13
#pragma unroll
    for (int k = 0; k < X::NL; ++k) {
        const int idx = tid + k * 256 < X::NVAL ? tid + k * 256 : 0;
        const int hy = idx / (X::IC * 3), e = idx % (X::IC * 3);
        slot[k] = hy | ((e / 3) << 5) | ((e % 3) << 11) | ((hy * X::RS + e) << 13);
    }
    auto tile_of = [&](int t, int& b, int& ty, int& tx) __attribute__((always_inline)) {
        tx = t % p.tiles_x; t /= p.tiles_x;
        ty = t % p.tiles_y;
        b = t / p.tiles_y;
    };
    auto fetch = [&](int t) __attribute__((always_inline)) {
        int b, ty, tx;
        tile_of(t, b, ty, tx);
        const float* img = p.image + (size_t)b * p.H * p.W * 3;
        const int iy0 = 4 * (ty * X::TH) - 5, ix0 = 4 * (tx * X::TW) - 5;       // m row 2 * (TH ty) - 1, image row 2 * that - 3
        okm = 0u;
#pragma unroll
        for (int k = 0; k < X::NL; ++k) {
            const int hy = slot[k] & 31, col = (slot[k] >> 5) & 63, c = (slot[k] >> 11) & 3;
            const int iy = iy0 + hy, ix = ix0 + col;
            const bool ok = tid + k * 256 < X::NVAL && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            okm |= ok ? 1u << k : 0u;
            raw[k] = img[ok ? (unsigned)((iy * p.W + ix) * 3 + c) : 0u];       // (unsigned: a 32-bit offset from the image's scalar base)
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
        _Float16* hv = reinterpret_cast<_Float16*>(sBuf + buf * X::BUF_BYTES);   // value v: high piece at half 4 (v / 2) + (v & 1), low piece 2 halves on
#pragma unroll
        for (int k = 0; k < X::NL; ++k) {
            if (tid + k * 256 < X::NVAL) {
                const int c = (slot[k] >> 11) & 3, at = (unsigned)slot[k] >> 13;
                const float mc = c == 0 ? mean[0] : c == 1 ? mean[1] : mean[2], rs = c == 0 ? 16.0f / stdv[0] : c == 1 ? 16.0f / stdv[1] : 16.0f / stdv[2];
                // (x / 255 - mean) / std as the reference normalises (resnet_50.py:41-44), with reciprocal multiplies: within an ulp of
                // its two divisions -- below the 2^-22 of the split that follows; 22 IEEE divisions per thread and tile were a quarter of
                // the kernel's VALU work -- times 16, the fp16 pieces' scale; zero padding AFTER normalisation
                const float v = ((okm >> k) & 1u) ? (raw[k] * (1.0f / 255.0f) - mc) * rs : 0.f;
                const _Float16 h = (_Float16)v;
                hv[4 * (at >> 1) + (at & 1)] = h;
                hv[4 * (at >> 1) + (at & 1) + 2] = (_Float16)(v - (float)h);
            }
        }
    };

    // ---- this lane's 20 pair offsets (values, relative to the pixel's first tap): pair jj of step s is k' = 32 s + 8 q + 2 jj.  A pair
    // is ONE aligned ds_read_b64 -- {high pair, low pair} -- because the halo is stored in 8-byte units of two values
    // (hi0 hi1 lo0 lo1) and every pair starts at an even value: a pixel's taps start 6 values = 24 bytes behind its neighbour's.
    // (Measured on the way, profiles/r06s_stem7p2_ab.txt: piece PLANES read by ds_read_b32 -- 40 reads per block -- 0.315 ms, the kernel
    // bound by its DS instruction count; rows of 24 = three 8-half groups on misaligned ds_read_b128 0.99 ms -- legal here, but lane
    // by lane: 21 B/clk/CU, scripts/micro/lds_unaligned_tp.hip -- and on four b32 with immediate offsets 0.36 ms: a sixth K step.)
    int koff[X::NSTEP][4];
#pragma unroll
    for (int s = 0; s < X::NSTEP; ++s)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int k = 32 * s + 8 * q + 2 * jj;
            koff[s][jj] = ((k / X::KROW) * X::RS + k % X::KROW) * 4;            // bytes (k' >= 154: the zero row / finite neighbours, zero weights)
        }
    // ---- and the A operands: channel 16 wave + px, the same k' (zero where k' is padding), 256 w split into fp16 pairs
    frag wa[X::NSTEP][2];
#pragma unroll
    for (int s = 0; s < X::NSTEP; ++s) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            float wv[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = 32 * s + 8 * q + 2 * jj + h;
                const int dy = k / X::KROW, r = k % X::KROW;
                const bool real = dy < 7 && r < 21;
                wv[h] = real ? h2_sat(p.w[(dy * 21 + r) * 64 + 16 * wave + px] * 256.0f) : 0.f;
            }
            hi[jj] = pack_hi(wv[0], wv[1]);
            lo[jj] = h2_low_pair(hi[jj], wv[0], wv[1]);
        }
        wa[s][0] = __builtin_bit_cast(frag, make_uint4(hi[0], hi[1], hi[2], hi[3]));
        wa[s][1] = __builtin_bit_cast(frag, make_uint4(lo[0], lo[1], lo[2], lo[3]));
    }
    // BN of this lane's 4 channels (16 wave + 4 q ..): the accumulator holds 16 x * 256 w
    f32x4p sc, sh;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sc[e] = p.scale[16 * wave + 4 * q + e] * (1.0f / 4096.0f);
        sh[e] = p.shift[16 * wave + 4 * q + e];
    }
    // this lane's pixels of the m region, block by block: (row, column) packed, and the halo offset of their first tap
    int pbase[X::NBLK];                                        // halves: (2 ry * RS + 6 rx) | ry << 20 | rx << 25  (pixels beyond 152: pixel 152's)

#pragma unroll
    for (int blk = 0; blk < X::NBLK; ++blk) {
        const int i = blk * 16 + px < X::NPIX ? blk * 16 + px : X::NPIX - 1;
        const int ry = i / X::MC, rx = i % X::MC;
        pbase[blk] = (2 * ry * X::RS + 6 * rx) | (ry << 20) | (rx << 25);
    }

    int t = blockIdx.x;
    if (t < p.tiles_total) fetch(t);
    __syncthreads();                                           // (the zero fill is complete)
    if (t < p.tiles_total) commit(0);
    __syncthreads();
    float sat_mx = 0.f;
    int buf = 0;
#pragma unroll 1
    for (; t < p.tiles_total; t += gridDim.x) {
        const int tn = t + gridDim.x;
        const bool has_next = tn < p.tiles_total;
        if (has_next) fetch(tn);
        int b, ty, tx;
        tile_of(t, b, ty, tx);
        const int my0 = 2 * ty * X::TH - 1, mx0 = 2 * tx * X::TW - 1;           // m pixel of region pixel (0, 0)
        // ---- A. the conv, two blocks of 16 m pixels at a time
#pragma unroll 1                                                // (fully unrolled: 0.278-0.292 ms against 0.274-0.283, same box)
        for (int bp = 0; bp < X::NBLK; bp += 2) {
            f32x4p acc[2] = {(f32x4p){0.f, 0.f, 0.f, 0.f}, (f32x4p){0.f, 0.f, 0.f, 0.f}};
            int pb[2] = {pbase[0], pbase[1]};                  // (the table is indexed by a run-time block pair of a rolled loop: selects, not scratch)
#pragma unroll
            for (int k = 2; k < X::NBLK; k += 2)
                if (k == bp) { pb[0] = pbase[k]; pb[1] = pbase[k + 1]; }
            const char* hb[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) hb[u] = sBuf + buf * X::BUF_BYTES + (pb[u] & 0xfffff) * 4;
#pragma unroll
            for (int st = 0; st < X::NSTEP; ++st) {
                frag xh[2], xl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    uint2 t[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) t[jj] = *reinterpret_cast<const uint2*>(hb[u] + koff[st][jj]);
                    xh[u] = __builtin_bit_cast(frag, make_uint4(t[0].x, t[1].x, t[2].x, t[3].x));
                    xl[u] = __builtin_bit_cast(frag, make_uint4(t[0].y, t[1].y, t[2].y, t[3].y));
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[st][1], xh[u], acc[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[st][0], xl[u], acc[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[st][0], xh[u], acc[u], 0, 0, 0);
            }
            // BN + ReLU, zero outside the map, parked: pixel i of the region at sP[i * 16 + 4 q ..]
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ry = (pb[u] >> 20) & 31, rx = (pb[u] >> 25) & 31;
                const bool in = (unsigned)(my0 + ry) < (unsigned)p.Hm && (unsigned)(mx0 + rx) < (unsigned)p.Wm;
                float4 v;
                v.x = in ? fmaxf(fmaf(acc[u][0], sc[0], sh[0]), 0.f) : 0.f;
                v.y = in ? fmaxf(fmaf(acc[u][1], sc[1], sh[1]), 0.f) : 0.f;
                v.z = in ? fmaxf(fmaf(acc[u][2], sc[2], sh[2]), 0.f) : 0.f;
                v.w = in ? fmaxf(fmaf(acc[u][3], sc[3], sh[3]), 0.f) : 0.f;
                *reinterpret_cast<float4*>(sP + ((bp + u) * 16 + px) * 16 + 4 * q) = v;      // (pixels 153 .. 159: duplicates of pixel 152, never read)
            }
        }
        // ---- B. the pool: item = (pooled pixel, channel quad) of this wave's 16 channels, two per lane
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = lane + 64 * it;
            const int cq = item & 3, pp = item >> 2;
            const int ply = pp / X::TW, plx = pp % X::TW;
            float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float4 v = *reinterpret_cast<const float4*>(sP + ((2 * ply + dy) * X::MC + 2 * plx + dx) * 16 + 4 * cq);
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
            const int oy = ty * X::TH + ply, ox = tx * X::TW + plx;
            float* o = p.out + (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.out_cs + p.out_co + 16 * wave;
            if (p.out_h2) {
                // lanes (item, item ^ 1) hold the two quads of an octet: each ends up with one whole 16-byte unit (even: the 8 high
                // pieces, odd: the 8 low pieces)
                const float s0 = m.x * p.act_scale, s1 = m.y * p.act_scale, s2 = m.z * p.act_scale, s3 = m.w * p.act_scale;
                sat_track(sat_mx, s0, s1);
                sat_track(sat_mx, s2, s3);
                const float v0 = h2_sat(s0), v1 = h2_sat(s1), v2 = h2_sat(s2), v3 = h2_sat(s3);
                const unsigned h0 = pack_hi(v0, v1), h1 = pack_hi(v2, v3);
                const unsigned l0 = h2_low_pair(h0, v0, v1), l1 = h2_low_pair(h1, v2, v3);
                const bool odd = cq & 1;
                const unsigned sx = odd ? h0 : l0, sy = odd ? h1 : l1;          // what the partner lane needs
                const unsigned rx_ = __shfl_xor(sx, 1), ry_ = __shfl_xor(sy, 1);
                const uint4 unit = odd ? make_uint4(rx_, ry_, l0, l1) : make_uint4(h0, h1, rx_, ry_);
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(o + (cq >> 1) * 8) + (odd ? 16 : 0)) = unit;
            } else {
                *reinterpret_cast<float4*>(o + 4 * cq) = m;
            }
        }
        // ---- C. the next tile's halo into the other buffer; one barrier: it is complete, and this one free to be overwritten a tile on
        if (has_next) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (p.out_h2) sat_report(p.sat, sat_mx);
}

int launch_stem7p(const romp_op& op, const float* image, float* out, int B, hipStream_t st) {
    using X = S7Cfg;
    ROMP_REQUIRE(op.Cin == 3 && op.Cout == 64 && op.ksize == 7 && op.stride == 2, "stem7p: expects 3->64 k7 s2 (+ max-pool 3 s2)");
    ROMP_REQUIRE(op.H % 32 == 0 && op.W % 32 == 0, "stem7p: input %dx%d must be a multiple of 32", op.H, op.W);
    ROMP_REQUIRE((op.out_cstride & 3) == 0 && (op.out_coff & 3) == 0, "stem7p: output channels must be float4 aligned");
    static bool attr = false;
    static int num_cu = 256;
    if (!attr) {
        ROMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem7p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, X::LDS_BYTES));
        int dev = 0;
        hipDeviceProp_t prop;
        ROMP_HIP_CHECK(hipGetDevice(&dev));
        ROMP_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        attr = true;
    }
    if (image == nullptr && out == nullptr) return ROMP_OK;    // set-up only (outside any stream capture)
    Stem7pParams p;
    memset(&p, 0, sizeof(p));
    p.image = image; p.w = op.weight; p.scale = op.scale; p.shift = op.shift; p.out = out;
    p.H = op.H; p.W = op.W; p.Hm = op.H / 2; p.Wm = op.W / 2; p.Ho = op.H / 4; p.Wo = op.W / 4;
    p.out_cs = op.out_cstride; p.out_co = op.out_coff;
    p.out_h2 = op.out_fmt == ROMP_FMT_H2; p.act_scale = ldexpf(1.f, op.act_shift);
    ROMP_REQUIRE(!p.out_h2 || ((op.out_cstride | op.out_coff) & 7) == 0, "stem7p: H2 output needs octet-aligned channels");
    p.sat = conv_sat_counter();
    p.tiles_x = p.Wo / X::TW; p.tiles_y = p.Ho / X::TH; p.tiles_total = B * p.tiles_x * p.tiles_y;
    long grid = 2L * num_cu;
    if (grid > p.tiles_total) grid = p.tiles_total;
    hipLaunchKernelGGL(stem7p_kernel, dim3((unsigned)grid), dim3(256), X::LDS_BYTES, st, p);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

}  // namespace romp
