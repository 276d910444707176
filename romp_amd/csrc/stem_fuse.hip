// stem_fuse.hip -- the two non-GEMM layer kernels of the network:
//   * stem conv1: input normalisation x/255*2-1 (model.py:384) fused into the 3->64 3x3 stride-2
//     convolution + BN + ReLU (model.py:385-387).  Two kernels: stem_conv_kernel (VALU, float32 products: the float32 /
//     calibration programs) and stem_mfma_kernel (K = 27 padded to one 32-wide f16x2 MFMA step, H2 output: the default path;
//     the VALU form was instruction-bound at 0.33 ms for 0.12 ms worth of HBM traffic).
//   * fuse-sum: y_i = relu(sum_j up_nearest(T_j))  (HighResolutionModule.forward model.py:233-244)
//     -- replaces the reference's nearest-upsample kernels and the chain of adds with one pass that
//     reads each term once and writes y once.  Summation order is the reference's (j ascending).
#include "conv_common.h"      // h2_pack / h2_unpack: the H2 activation format
#include "conv_split.h"       // f16x8
#include "conv_fuse.h"        // h2_low_pair
#include <stdlib.h>
#include <string.h>

namespace romp {

struct StemParams {
    const float* image; const float* w; const float* scale; const float* shift; float* out;
    int H, W, Ho, Wo, out_cs, out_co, tiles_x, tiles_y;
    int out_h2; float act_scale;      // output in the H2 format (conv_common.h)
    int* sat;                         // saturation counter (conv_common.h sat_report), or nullptr
};

// packed stem weight: [tap 9][cin 3][cout 64]
__global__ __launch_bounds__(256) void stem_conv_kernel(StemParams p) {
    constexpr int T = 16, HS = 2 * T + 1;           // 16x16 output tile, 33x33 input halo
    __shared__ __attribute__((aligned(16))) float s_in[HS * HS * 3];
    __shared__ __attribute__((aligned(16))) float s_w[27 * 64];
    const int tid = threadIdx.x;
    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x; bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const float* img = p.image + (size_t)b * p.H * p.W * 3;
    const int iy0 = ty * T * 2 - 1, ix0 = tx * T * 2 - 1;
    {   // halo fill, branch-free and batched: all loads of a thread are in flight together (a conditional load makes
        // hipcc wait inside the branch: 13 serialized round trips per thread, which had this kernel at 0.54 ms)
        constexpr int NL = (HS * HS * 3 + 255) / 256;
        float raw[NL];
        bool ok[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int idx = tid + k * 256;
            const int idc = idx < HS * HS * 3 ? idx : 0;
            const int e = idc % (HS * 3), hy = idc / (HS * 3);
            const int iy = iy0 + hy, ix = ix0 + e / 3;
            ok[k] = idx < HS * HS * 3 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            raw[k] = img[ok[k] ? ((size_t)iy * p.W + ix0) * 3 + e : 0];
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int idx = tid + k * 256;               // zero padding is applied AFTER normalisation
            if (idx < HS * HS * 3) s_in[idx] = ok[k] ? (raw[k] / 255.0f) * 2.0f - 1.0f : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < (27 * 64 + 255) / 256; ++k)
        if (tid + k * 256 < 27 * 64) s_w[tid + k * 256] = p.w[tid + k * 256];
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int cg = lane & 15, ps = lane >> 4;
    float4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const float4 w4 = *reinterpret_cast<const float4*>(s_w + (tap * 3 + ci) * 64 + cg * 4);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int row = wave * 4 + (t >> 2), col = (t & 3) * 4 + ps;
                const float x = s_in[((row * 2 + dy) * HS + col * 2 + dx) * 3 + ci];
                acc[t].x = fmaf(x, w4.x, acc[t].x); acc[t].y = fmaf(x, w4.y, acc[t].y);
                acc[t].z = fmaf(x, w4.z, acc[t].z); acc[t].w = fmaf(x, w4.w, acc[t].w);
            }
        }
    }
    const float4 sc = *reinterpret_cast<const float4*>(p.scale + cg * 4);
    const float4 sh = *reinterpret_cast<const float4*>(p.shift + cg * 4);
    float* out = p.out + (size_t)b * p.Ho * p.Wo * p.out_cs + p.out_co;
    float sat_mx = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int oy = ty * T + wave * 4 + (t >> 2), ox = tx * T + (t & 3) * 4 + ps;
        float4 v;
        v.x = fmaxf(fmaf(acc[t].x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(acc[t].y, sc.y, sh.y), 0.f);
        v.z = fmaxf(fmaf(acc[t].z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(acc[t].w, sc.w, sh.w), 0.f);
        if (p.out_h2) {                                   // channels 4cg..4cg+3 = half (cg & 1) of octet cg >> 1
            // The two lanes of an octet trade halves so that each stores ONE whole 16-byte unit (even lane: the eight high
            // pieces, odd lane: the eight low pieces): a wave's store covers 4 pixels x 256 contiguous bytes, instead of two
            // instructions of 8-byte pieces on alternating 16-byte slots (the 8-byte stores had this kernel at 2.2 TB/s)
            uint2 hi, lo;
            h2_pack(v, p.act_scale, hi, lo, sat_mx);
            const bool odd = cg & 1;
            const uint2 send = odd ? hi : lo;
            uint2 recv;
            recv.x = __shfl_xor(send.x, 1);
            recv.y = __shfl_xor(send.y, 1);
            const uint4 unit = odd ? make_uint4(recv.x, recv.y, lo.x, lo.y) : make_uint4(hi.x, hi.y, recv.x, recv.y);
            char* o = reinterpret_cast<char*>(out + ((size_t)oy * p.Wo + ox) * p.out_cs + (cg >> 1) * 8) + (odd ? 16 : 0);
            *reinterpret_cast<uint4*>(o) = unit;
        } else {
            *reinterpret_cast<float4*>(out + ((size_t)oy * p.Wo + ox) * p.out_cs + cg * 4) = v;
        }
    }
    if (p.out_h2) sat_report(p.sat, sat_mx);
}

// The same layer on the matrix cores (round 3, H2 output only): the VALU kernel above is instruction-bound (1 728 FMAs per thread,
// SQ_ACTIVE_INST_ANY 69 % of its wave cycles, 0.33 ms for 0.12 ms worth of HBM traffic at B = 32).  K = 27 pads to ONE 32-wide MFMA
// step: v_mfma_f32_16x16x32_f16 with A = 16 output channels x (tap, colour) as fp16 pairs of 256 w (registers, split once per
// workgroup) and B = the im2col column of a pixel, gathered from the normalised halo in LDS and split into fp16 pairs of 16 x on the
// fly -- the three f16x2 products, float32 accumulate, like every other conv.  Lane (px, kq) gathers k = 8 kq .. 8 kq + 7 of pixel
// px of a 16-pixel row block once for all 64 output channels (4 groups x 3 products = 12 MFMAs per gather); k = 27 .. 31 read any
// valid halo value against zero weights.  Wave w: rows 4 w .. 4 w + 3 of the 16 x 16 tile.
__global__ __launch_bounds__(256) void stem_mfma_kernel(StemParams p) {
    constexpr int T = 16, HS = 2 * T + 1;
    using frag = f16x8;
    typedef float f32x4s __attribute__((ext_vector_type(4)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) float s_in[HS * HS * 3];
    const int tid = threadIdx.x;
    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x; bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const float* img = p.image + (size_t)b * p.H * p.W * 3;
    const int iy0 = ty * T * 2 - 1, ix0 = tx * T * 2 - 1;
    {   // halo fill, branch-free and batched (see stem_conv_kernel)
        constexpr int NL = (HS * HS * 3 + 255) / 256;
        float raw[NL];
        bool ok[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int idx = tid + k * 256;
            const int idc = idx < HS * HS * 3 ? idx : 0;
            const int e = idc % (HS * 3), hy = idc / (HS * 3);
            const int iy = iy0 + hy, ix = ix0 + e / 3;
            ok[k] = idx < HS * HS * 3 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            raw[k] = img[ok[k] ? ((size_t)iy * p.W + ix0) * 3 + e : 0];
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int idx = tid + k * 256;               // zero padding is applied AFTER normalisation; x 16: the fp16 pieces' scale
            if (idx < HS * HS * 3) s_in[idx] = ok[k] ? ((raw[k] / 255.0f) * 2.0f - 1.0f) * 16.0f : 0.f;
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
    const int px = lane & 15, q = lane >> 4;
    auto pack_hi = [&](float a, float c) __attribute__((always_inline)) {
        const f32x2_t v = {a, c};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
    };
    // this lane's 8 k indices -> offsets into the halo (floats, relative to the pixel's top-left tap) ...
    int koff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * q + j < 27 ? 8 * q + j : 26;
        const int tap = k / 3, ci = k % 3;
        koff[j] = ((tap / 3) * HS + tap % 3) * 3 + ci;
    }
    // ... and the A operands: channel 16 g + px, the same 8 k (zero beyond 26), 256 w split into fp16 pairs
    frag wa[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int k0 = 8 * q + 2 * jj, k1 = k0 + 1;
            const float w0 = k0 < 27 ? h2_sat(p.w[k0 * 64 + 16 * g + px] * 256.0f) : 0.f;
            const float w1 = k1 < 27 ? h2_sat(p.w[k1 * 64 + 16 * g + px] * 256.0f) : 0.f;
            hi[jj] = pack_hi(w0, w1);
            lo[jj] = h2_low_pair(hi[jj], w0, w1);
        }
        wa[g][0] = __builtin_bit_cast(frag, make_uint4(hi[0], hi[1], hi[2], hi[3]));
        wa[g][1] = __builtin_bit_cast(frag, make_uint4(lo[0], lo[1], lo[2], lo[3]));
    }
    __syncthreads();

    f32x4s acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[r][g] = (f32x4s){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float* base = s_in + ((2 * (4 * wave + r)) * HS + 2 * px) * 3;
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
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][1], xh, acc[r][g], 0, 0, 0);
            acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][0], xl, acc[r][g], 0, 0, 0);
            acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][0], xh, acc[r][g], 0, 0, 0);
        }
    }
    // epilogue: lane (px, q) holds channels 16 g + 4 q .. + 3 of pixel px of row block r: BN + ReLU in the scaled domain, split, the
    // lanes of an octet trade halves (v_permlane16_swap) and each stores one 16-byte unit
    const float prod_scale = p.act_scale * (1.0f / 4096.0f);
    float sat_mx = 0.f;
    float* out = p.out + (size_t)b * p.Ho * p.Wo * p.out_cs + p.out_co + ((size_t)(ty * T + 4 * wave) * p.Wo + tx * T + px) * p.out_cs + 4 * q;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 sc = *reinterpret_cast<const float4*>(p.scale + 16 * g + 4 * q);
        const float4 sh = *reinterpret_cast<const float4*>(p.shift + 16 * g + 4 * q);
        const float s4[4] = {sc.x * prod_scale, sc.y * prod_scale, sc.z * prod_scale, sc.w * prod_scale};
        const float b4[4] = {sh.x * p.act_scale, sh.y * p.act_scale, sh.z * p.act_scale, sh.w * p.act_scale};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(acc[r][g][e], s4[e], b4[e]), 0.f);
            sat_track(sat_mx, v[0], v[1]);
            sat_track(sat_mx, v[2], v[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = h2_sat(v[e]);
            unsigned hh[2] = {pack_hi(v[0], v[1]), pack_hi(v[2], v[3])};
            unsigned hl[2] = {h2_low_pair(hh[0], v[0], v[1]), h2_low_pair(hh[1], v[2], v[3])};
            const u32x2_t a = __builtin_amdgcn_permlane16_swap(hh[0], hl[0], false, false);
            const u32x2_t c = __builtin_amdgcn_permlane16_swap(hh[1], hl[1], false, false);
            *reinterpret_cast<uint4*>(out + (size_t)r * p.Wo * p.out_cs + 16 * g) = make_uint4(a[0], c[0], a[1], c[1]);
        }
    }
    sat_report(p.sat, sat_mx);
}

int launch_stem(const romp_op& op, const float* image, float* out, int B, hipStream_t st) {
    ROMP_REQUIRE(op.Cin == 3 && op.Cout == 64 && op.ksize == 3 && op.stride == 2, "stem: expects 3->64 k3 s2");
    ROMP_REQUIRE(op.H % 32 == 0 && op.W % 32 == 0, "stem: input %dx%d must be a multiple of 32", op.H, op.W);
    ROMP_REQUIRE((op.out_cstride & 3) == 0 && (op.out_coff & 3) == 0, "stem: output channels must be float4 aligned");
    StemParams p;
    p.image = image; p.w = op.weight; p.scale = op.scale; p.shift = op.shift; p.out = out;
    p.H = op.H; p.W = op.W; p.Ho = op.H / 2; p.Wo = op.W / 2;
    p.out_cs = op.out_cstride; p.out_co = op.out_coff;
    p.out_h2 = op.out_fmt == ROMP_FMT_H2; p.act_scale = ldexpf(1.f, op.act_shift);
    p.sat = conv_sat_counter();
    ROMP_REQUIRE(!p.out_h2 || ((op.out_cstride | op.out_coff) & 7) == 0, "stem: H2 output needs octet-aligned channels");
    p.tiles_x = p.Wo / 16; p.tiles_y = p.Ho / 16;
    // ROMP_OPF_STEM_VALU (plan.Program.stem): the float32 VALU kernel for the H2 output too -- set by env ROMP_STEM=valu (A/B runs,
    // tests) and whenever 256 |w| would leave the fp16 pieces of the MFMA form (its 16x input / 256x weight scales are internal to
    // the kernel: x/255*2-1 lies in [-1, 1] whatever act_shift the OUTPUT tensor uses).
    const bool use_mfma = !(op.flags & ROMP_OPF_STEM_VALU);
    if (p.out_h2 && use_mfma) hipLaunchKernelGGL(stem_mfma_kernel, dim3((unsigned)(B * p.tiles_x * p.tiles_y)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(stem_conv_kernel, dim3((unsigned)(B * p.tiles_x * p.tiles_y)), dim3(256), 0, st, p);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

// ---- ResNet-50 stem (romp/lib/models/resnet_50.py:32-38,41-44,56): ImageNet normalisation (x/255 - mean)/std fused
// into the 3->64 7x7 stride-2 pad-3 convolution + BN + ReLU, then MaxPool2d(3, 2, 1).  Same VALU structure as the
// 3x3 stem: a 16x16 output tile per workgroup, the 37x37x3 normalised halo and the 147x64 weights in LDS.
struct Stem7Params {
    const float* image; const float* w; const float* scale; const float* shift; float* out;
    int H, W, Ho, Wo, out_cs, out_co, tiles_x, tiles_y;
};

__global__ __launch_bounds__(256) void stem7_conv_kernel(Stem7Params p) {
    constexpr int T = 16, HS = 2 * T + 5;
    __shared__ __attribute__((aligned(16))) float s_in[HS * HS * 3];
    __shared__ __attribute__((aligned(16))) float s_w[147 * 64];
    const int tid = threadIdx.x;
    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x; bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const float* img = p.image + (size_t)b * p.H * p.W * 3;
    const int iy0 = ty * T * 2 - 3, ix0 = tx * T * 2 - 3;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    {   // halo fill, branch-free and batched (see stem_conv_kernel)
        constexpr int NL = (HS * HS * 3 + 255) / 256;
        float raw[NL];
        bool ok[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int idx = tid + k * 256;
            const int idc = idx < HS * HS * 3 ? idx : 0;
            const int c = idc % 3, hx = (idc / 3) % HS, hy = idc / (HS * 3);
            const int iy = iy0 + hy, ix = ix0 + hx;
            ok[k] = idx < HS * HS * 3 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            raw[k] = img[ok[k] ? ((size_t)iy * p.W + ix) * 3 + c : 0];
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int idx = tid + k * 256;               // zero padding is applied AFTER normalisation
            const int c = idx % 3;
            if (idx < HS * HS * 3) s_in[idx] = ok[k] ? (raw[k] / 255.0f - mean[c]) / stdv[c] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < (147 * 64 + 255) / 256; ++k)
        if (tid + k * 256 < 147 * 64) s_w[tid + k * 256] = p.w[tid + k * 256];
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int cg = lane & 15, ps = lane >> 4;
    float4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int dy = 0; dy < 7; ++dy)
#pragma unroll 1
        for (int dx = 0; dx < 7; ++dx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float4 w4 = *reinterpret_cast<const float4*>(s_w + ((dy * 7 + dx) * 3 + ci) * 64 + cg * 4);
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int row = wave * 4 + (t >> 2), col = (t & 3) * 4 + ps;
                    const float x = s_in[((row * 2 + dy) * HS + col * 2 + dx) * 3 + ci];
                    acc[t].x = fmaf(x, w4.x, acc[t].x); acc[t].y = fmaf(x, w4.y, acc[t].y);
                    acc[t].z = fmaf(x, w4.z, acc[t].z); acc[t].w = fmaf(x, w4.w, acc[t].w);
                }
            }
    const float4 sc = *reinterpret_cast<const float4*>(p.scale + cg * 4);
    const float4 sh = *reinterpret_cast<const float4*>(p.shift + cg * 4);
    float* out = p.out + (size_t)b * p.Ho * p.Wo * p.out_cs + p.out_co;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int oy = ty * T + wave * 4 + (t >> 2), ox = tx * T + (t & 3) * 4 + ps;
        float4 v;
        v.x = fmaxf(fmaf(acc[t].x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(acc[t].y, sc.y, sh.y), 0.f);
        v.z = fmaxf(fmaf(acc[t].z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(acc[t].w, sc.w, sh.w), 0.f);
        *reinterpret_cast<float4*>(out + ((size_t)oy * p.Wo + ox) * p.out_cs + cg * 4) = v;
    }
}

int launch_stem7(const romp_op& op, const float* image, float* out, int B, hipStream_t st) {
    ROMP_REQUIRE(op.Cin == 3 && op.Cout == 64 && op.ksize == 7 && op.stride == 2, "stem7: expects 3->64 k7 s2");
    ROMP_REQUIRE(op.H % 32 == 0 && op.W % 32 == 0, "stem7: input %dx%d must be a multiple of 32", op.H, op.W);
    ROMP_REQUIRE((op.out_cstride & 3) == 0 && (op.out_coff & 3) == 0, "stem7: output channels must be float4 aligned");
    Stem7Params p;
    p.image = image; p.w = op.weight; p.scale = op.scale; p.shift = op.shift; p.out = out;
    p.H = op.H; p.W = op.W; p.Ho = op.H / 2; p.Wo = op.W / 2;
    p.out_cs = op.out_cstride; p.out_co = op.out_coff;
    p.tiles_x = p.Wo / 16; p.tiles_y = p.Ho / 16;
    hipLaunchKernelGGL(stem7_conv_kernel, dim3((unsigned)(B * p.tiles_x * p.tiles_y)), dim3(256), 0, st, p);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

// MaxPool2d(kernel 3, stride 2, padding 1) on NHWC float32 (padding = -inf, like torch)
__global__ void maxpool3s2_kernel(const float* __restrict__ in, int H, int W, int C4, int in_cs, float* __restrict__ out, int out_cs,
                                  size_t total) {
    const int Ho = H / 2, Wo = W / 2;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int c = (int)(r % C4) * 4; r /= C4;
        const int x = (int)(r % Wo); r /= Wo;
        const int y = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float4 m = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int iy = 2 * y + dy, ix = 2 * x + dx;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    const float4 v = *reinterpret_cast<const float4*>(in + (((size_t)b * H + iy) * W + ix) * in_cs + c);
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
            }
        *reinterpret_cast<float4*>(out + (((size_t)b * Ho + y) * Wo + x) * out_cs + c) = m;
    }
}

int launch_maxpool(const romp_op& op, const float* in, float* out, int B, hipStream_t st) {
    ROMP_REQUIRE(in && out && (op.Cin & 3) == 0 && (op.in_cstride & 3) == 0 && (op.out_cstride & 3) == 0 && !(op.H & 1) && !(op.W & 1),
                 "maxpool: bad shape");
    const size_t total = (size_t)B * (op.H / 2) * (op.W / 2) * (op.Cin / 4);
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(maxpool3s2_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, op.H, op.W, op.Cin / 4, op.in_cstride, out,
                       op.out_cstride, total);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

struct FuseParams {
    const float* t[4]; int shift[4]; int cs[4]; int h2[4];        // (a term's channel offset is folded into its pointer)
    float* out; int n_terms, H, W, C8, out_cs, out_co, relu, out_h2; float act_scale, inv_act_scale;
    int rows_total, c8_shift;                                     // B * H output rows; log2(C8) when C8 and W are powers of two
    int* sat;
};

// One thread per pixel and channel OCTET (32 bytes in either format).  H2 terms are summed in the scaled domain
// (x * 2^act_shift: exact, a power of two commutes with every f32 rounding), float32 terms are scaled on the way in.
// Work mapping (round 4): a workgroup owns whole output ROWS (blockIdx -> (image, row): one scalar division per row), a thread's
// (column, octet) comes from shifts -- W and C / 8 are powers of two in every HRNet fuse layer (`pow2` = 0: the general form with
// divisions).  Round 3 derived (octet, x, y, image) from a flat index with four runtime integer divisions per 32 output bytes:
// ~150 VALU instructions of index arithmetic per element had this streaming kernel at 3.7-4.0 TB/s.
template <int POW2>
__global__ __launch_bounds__(256) void fusesum_kernel(FuseParams p) {
    float sat_mx = 0.f;
    const int per_row = p.W * p.C8;
    const bool scaled = p.out_h2 || p.h2[0] || p.h2[1] || p.h2[2] || p.h2[3];
    const float in_scale = scaled ? p.act_scale : 1.f;
    for (int row = blockIdx.x; row < p.rows_total; row += gridDim.x) {
        const int b = row / p.H, y = row - b * p.H;
        const float* tp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                 // the row of term k this output row reads
            const int s = p.shift[k];
            tp[k] = p.t[k] + ((size_t)b * (p.H >> s) + (y >> s)) * (size_t)(p.W >> s) * p.cs[k];
        }
        float* orow = p.out + ((size_t)b * p.H + y) * (size_t)p.W * p.out_cs + p.out_co;
#pragma unroll 2
        for (int i = threadIdx.x; i < per_row; i += blockDim.x) {
            const int x = POW2 ? i >> p.c8_shift : i / p.C8;
            const int c = (POW2 ? i & (p.C8 - 1) : i - x * p.C8) * 8;
            float4 va, vb;                             // channels c..c+3, c+4..c+7 (scaled by act_scale when any H2 is involved)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < p.n_terms) {
                    const float* q = tp[k] + (size_t)(x >> p.shift[k]) * p.cs[k] + c;
                    const float4 u0 = *reinterpret_cast<const float4*>(q), u1 = *reinterpret_cast<const float4*>(q + 4);
                    float4 ta, tb;
                    if (p.h2[k]) {                     // u0 = eight high pieces, u1 = eight low pieces
                        const uint4 hi = __builtin_bit_cast(uint4, u0), lo = __builtin_bit_cast(uint4, u1);
                        ta = h2_unpack(make_uint2(hi.x, hi.y), make_uint2(lo.x, lo.y), 1.f);
                        tb = h2_unpack(make_uint2(hi.z, hi.w), make_uint2(lo.z, lo.w), 1.f);
                    } else {
                        ta = make_float4(u0.x * in_scale, u0.y * in_scale, u0.z * in_scale, u0.w * in_scale);
                        tb = make_float4(u1.x * in_scale, u1.y * in_scale, u1.z * in_scale, u1.w * in_scale);
                    }
                    if (k == 0) { va = ta; vb = tb; }
                    else {
                        va.x += ta.x; va.y += ta.y; va.z += ta.z; va.w += ta.w;
                        vb.x += tb.x; vb.y += tb.y; vb.z += tb.z; vb.w += tb.w;
                    }
                }
            }
            if (p.relu) {
                va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f); va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
                vb.x = fmaxf(vb.x, 0.f); vb.y = fmaxf(vb.y, 0.f); vb.z = fmaxf(vb.z, 0.f); vb.w = fmaxf(vb.w, 0.f);
            }
            float* op_ = orow + (size_t)x * p.out_cs + c;
            if (p.out_h2) {
                uint2 ha, la, hb, lb;
                h2_pack(va, 1.f, ha, la, sat_mx);
                h2_pack(vb, 1.f, hb, lb, sat_mx);
                *reinterpret_cast<uint4*>(op_) = make_uint4(ha.x, ha.y, hb.x, hb.y);
                *reinterpret_cast<uint4*>(op_ + 4) = make_uint4(la.x, la.y, lb.x, lb.y);
            } else {
                const float os = scaled ? p.inv_act_scale : 1.f;
                *reinterpret_cast<float4*>(op_) = make_float4(va.x * os, va.y * os, va.z * os, va.w * os);
                *reinterpret_cast<float4*>(op_ + 4) = make_float4(vb.x * os, vb.y * os, vb.z * os, vb.w * os);
            }
        }
    }
    if (p.out_h2) sat_report(p.sat, sat_mx);
}

int launch_fusesum(const FuseTerm* terms, int n_terms, float* out, int B, int H, int W, int C,
                   int out_cstride, int out_coff, int relu, hipStream_t st, int out_fmt, int act_shift) {
    ROMP_REQUIRE(n_terms >= 1 && n_terms <= 4, "fusesum: %d terms unsupported", n_terms);
    ROMP_REQUIRE((C & 7) == 0 && (out_cstride & 3) == 0 && (out_coff & 3) == 0, "fusesum: channels must come in octets");
    FuseParams p;
    for (int k = 0; k < 4; ++k) { p.t[k] = nullptr; p.shift[k] = 0; p.cs[k] = 0; p.h2[k] = 0; }
    for (int k = 0; k < n_terms; ++k) {
        ROMP_REQUIRE((terms[k].cstride & 3) == 0, "fusesum: term stride must be float4 aligned");
        ROMP_REQUIRE((H >> terms[k].shift) << terms[k].shift == H, "fusesum: term %d shift %d does not divide H", k, terms[k].shift);
        ROMP_REQUIRE(terms[k].fmt != ROMP_FMT_H2 || (terms[k].cstride & 7) == 0, "fusesum: H2 term stride must be octet aligned");
        p.t[k] = terms[k].ptr; p.shift[k] = terms[k].shift; p.cs[k] = terms[k].cstride; p.h2[k] = terms[k].fmt == ROMP_FMT_H2;
    }
    p.out = out; p.n_terms = n_terms; p.H = H; p.W = W; p.C8 = C / 8;
    p.out_cs = out_cstride; p.out_co = out_coff; p.relu = relu;
    p.out_h2 = out_fmt == ROMP_FMT_H2;
    ROMP_REQUIRE(!p.out_h2 || ((out_cstride | out_coff) & 7) == 0, "fusesum: H2 output needs octet-aligned channels");
    p.act_scale = ldexpf(1.f, act_shift); p.inv_act_scale = ldexpf(1.f, -act_shift);
    p.sat = conv_sat_counter();
    p.rows_total = B * H;
    const bool pow2 = (p.C8 & (p.C8 - 1)) == 0 && (W & (W - 1)) == 0;
    p.c8_shift = 0;
    while ((1 << p.c8_shift) < p.C8) ++p.c8_shift;
    // rows of fewer than 256 (pixel, octet) units: smaller workgroups, so that no thread idles; every CU gets several workgroups
    const int per_row = W * p.C8;
    const int threads = per_row >= 256 ? 256 : (per_row >= 128 ? 128 : 64);
    int blocks = p.rows_total;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (pow2) hipLaunchKernelGGL(fusesum_kernel<1>, dim3((unsigned)blocks), dim3(threads), 0, st, p);
    else hipLaunchKernelGGL(fusesum_kernel<0>, dim3((unsigned)blocks), dim3(threads), 0, st, p);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

// ---- split-K tail ------------------------------------------------------------------------------------------------
// A layer with few pixels and many input channels (16x16x256 at batch 1) gives the conv kernels a handful of work items with a
// long serial channel loop.  The single-image plans run it as a grouped conv over G input-channel slices (G times the items,
// 1/G of the loop) into G float32 partial tensors; this kernel adds them in slice order and applies the layer's epilogue.
struct KsumParams {
    const float* part; const float* res; const float* scale; const float* shift; float* out;
    int G, C, C8, part_cs, res_cs, res_co, out_cs, out_co, relu, relu_from, res_h2, out_h2;
    float act_scale, inv_act_scale; size_t total;
    int* sat;
};

__global__ __launch_bounds__(256) void ksum_kernel(KsumParams p) {
    float sat_mx = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < p.total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % p.C8) * 8;
        const size_t pix = i / p.C8;
        const float* pp = p.part + pix * p.part_cs + c;
        // eight slices per round, every load of a round issued before the first add (a rolled loop waited for each slice in
        // turn: 4.7 us per launch on the single-image critical path); slices are added in order, as before
        float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
        for (int g0 = 0; g0 < p.G; g0 += 8) {
            float4 ta[8], tb[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int gg = min(g0 + g, p.G - 1);
                ta[g] = ldg4(pp + (size_t)gg * p.C);
                tb[g] = ldg4(pp + (size_t)gg * p.C + 4);
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g0 + g < p.G) {
                    if (g0 + g == 0) { va = ta[0]; vb = tb[0]; }
                    else {
                        va.x += ta[g].x; va.y += ta[g].y; va.z += ta[g].z; va.w += ta[g].w;
                        vb.x += tb[g].x; vb.y += tb[g].y; vb.z += tb[g].z; vb.w += tb[g].w;
                    }
                }
            }
        }
        const float4 sa = ldg4(p.scale + c), sb = ldg4(p.scale + c + 4), ha = ldg4(p.shift + c), hb = ldg4(p.shift + c + 4);
        va = make_float4(fmaf(va.x, sa.x, ha.x), fmaf(va.y, sa.y, ha.y), fmaf(va.z, sa.z, ha.z), fmaf(va.w, sa.w, ha.w));
        vb = make_float4(fmaf(vb.x, sb.x, hb.x), fmaf(vb.y, sb.y, hb.y), fmaf(vb.z, sb.z, hb.z), fmaf(vb.w, sb.w, hb.w));
        if (p.res) {
            const float* rp = p.res + pix * p.res_cs + p.res_co + c;
            const float4 u0 = ldg4(rp), u1 = ldg4(rp + 4);
            float4 ra = u0, rb = u1;
            if (p.res_h2) {
                const uint4 hi = __builtin_bit_cast(uint4, u0), lo = __builtin_bit_cast(uint4, u1);
                ra = h2_unpack(make_uint2(hi.x, hi.y), make_uint2(lo.x, lo.y), p.inv_act_scale);
                rb = h2_unpack(make_uint2(hi.z, hi.w), make_uint2(lo.z, lo.w), p.inv_act_scale);
            }
            va.x += ra.x; va.y += ra.y; va.z += ra.z; va.w += ra.w;
            vb.x += rb.x; vb.y += rb.y; vb.z += rb.z; vb.w += rb.w;
        }
        if (p.relu && c >= p.relu_from) {
            va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f); va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
            vb.x = fmaxf(vb.x, 0.f); vb.y = fmaxf(vb.y, 0.f); vb.z = fmaxf(vb.z, 0.f); vb.w = fmaxf(vb.w, 0.f);
        }
        float* op_ = p.out + pix * p.out_cs + p.out_co + c;
        if (p.out_h2) {
            uint2 h0, l0, h1, l1;
            h2_pack(va, p.act_scale, h0, l0, sat_mx);
            h2_pack(vb, p.act_scale, h1, l1, sat_mx);
            *reinterpret_cast<uint4*>(op_) = make_uint4(h0.x, h0.y, h1.x, h1.y);
            *reinterpret_cast<uint4*>(op_ + 4) = make_uint4(l0.x, l0.y, l1.x, l1.y);
        } else {
            *reinterpret_cast<float4*>(op_) = va;
            *reinterpret_cast<float4*>(op_ + 4) = vb;
        }
    }
    if (p.out_h2) sat_report(p.sat, sat_mx);
}

int launch_ksum(const romp_op& op, const float* partial, const float* res, float* out, int B, hipStream_t st) {
    ROMP_REQUIRE(partial && out && op.scale && op.shift, "ksum: null buffer");
    ROMP_REQUIRE(op.groups >= 1 && (op.Cout & 7) == 0 && op.in_cstride == op.groups * op.Cout, "ksum: partials must be %d x %d channels", op.groups, op.Cout);
    ROMP_REQUIRE(((op.out_cstride | op.out_coff) & 7) == 0 && (!res || ((op.res_cstride | op.res_coff) & 7) == 0), "ksum: channels must come in octets");
    KsumParams p;
    p.part = partial; p.res = res; p.scale = (const float*)op.scale; p.shift = (const float*)op.shift; p.out = out;
    p.G = op.groups; p.C = op.Cout; p.C8 = op.Cout / 8; p.part_cs = op.in_cstride;
    p.res_cs = op.res_cstride; p.res_co = op.res_coff; p.out_cs = op.out_cstride; p.out_co = op.out_coff; p.relu = op.relu;
    p.relu_from = op.relu ? op.relu_from : 0;
    p.res_h2 = op.res_fmt == ROMP_FMT_H2; p.out_h2 = op.out_fmt == ROMP_FMT_H2;
    p.act_scale = ldexpf(1.f, op.act_shift); p.inv_act_scale = ldexpf(1.f, -op.act_shift);
    p.sat = conv_sat_counter();
    p.total = (size_t)B * op.H * op.W * p.C8;
    size_t blocks = (p.total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(ksum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

}  // namespace romp
