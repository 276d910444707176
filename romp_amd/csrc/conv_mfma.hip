// conv_mfma.hip -- NHWC float32 convolution (1x1 / 3x3, stride 1 / 2) as an implicit GEMM on the
// gfx950 f32-input matrix cores, with the inference BatchNorm (folded to scale/shift), the
// residual add and the ReLU fused into the epilogue.
//
// Replaces, per layer, the conv2d + batch_norm + add + relu op sequence the reference launches
// (BasicBlock.forward model.py:67-83, Bottleneck.forward :103-123, transition / fuse / head convs).
//
// GEMM view:  M = output pixels (B*Ho*Wo), N = Cout, K = taps*Cin.
//   v_mfma_f32_32x32x2_f32: A[i=lane&31][k=lane>>5], B[k=lane>>5][j=lane&31],
//   D[row][col]: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
//   A rows are pixels, B columns are output channels, so one lane owns ONE output channel for 16
//   pixels: the BN scale/shift is two scalars per lane and every store instruction writes two
//   128-byte channel runs.
//   The arithmetic is exact f32 (one rounding per product, f32 accumulate) -- the parity mode the
//   1e-4 gate needs; gfx950 has no TF32-like shortcut.
//
// Workgroup = 4 waves (one per SIMD).  A workgroup owns a TH x TW output-pixel tile of one image
// and NT*32 output channels; each wave owns MT M-blocks (32 pixels each) x NT N-blocks.
// Per input-channel chunk (CK channels) the haloed input tile and the weight slab are staged in
// LDS; the next chunk's global loads are issued BEFORE the MFMA loop of the current chunk and
// written to LDS after it (issue-early / write-late), so HBM/L2 latency hides under the MFMAs.
// LDS pixel stride is CK+4 floats: the ds_read_b128 A-fragment reads of 16 consecutive pixels then
// hit 16 distinct 16-byte bank slots (conflict-free at stride 1, 2-way at stride 2).
#include "common.h"

namespace romp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float* in; const float* w; const float* scale; const float* shift; const float* res;
    float* out;
    int H, W, Ho, Wo;
    int Cout;                 // valid output channels per group (store mask)
    int cin_valid;            // channels physically present in the input (loader mask)
    int cin_pad, cout_pad;    // packed weight dims
    int in_cs, in_co, in_gs;
    int out_cs, out_co, out_gs;
    int res_cs, res_co, res_gs;
    int relu;
    int tiles_x, tiles_y;
    int w_gs;                 // floats per group in the packed weight
};

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int KS, int S, int MT, int NT, int TW, int CK>
struct ConvCfg {
    static constexpr int TAPS = KS * KS;
    static constexpr int PAD = KS / 2;
    static constexpr int RPB = 32 / TW;              // tile rows per 32-pixel M-block
    static constexpr int TH = 4 * MT * RPB;          // output tile rows
    static constexpr int HR = (TH - 1) * S + KS;     // haloed input rows
    static constexpr int HC = (TW - 1) * S + KS;
    static constexpr int PS = CK + 4;                // LDS floats per pixel (padded)
    static constexpr int NW = NT * 32;               // output channels per workgroup
    static constexpr int QC = CK / 4;                // float4 per pixel per chunk
    static constexpr int A_VEC = HR * HC * QC;
    static constexpr int B_VEC = TAPS * QC * NW;
    static constexpr int NA = (A_VEC + 255) / 256;
    static constexpr int NB = (B_VEC + 255) / 256;
    static constexpr int LDS_BYTES = (HR * HC * PS + TAPS * CK * NW) * 4;
};

template <int KS, int S, int MT, int NT, int TW, int CK>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvParams p) {
    using C = ConvCfg<KS, S, MT, NT, TW, CK>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + C::HR * C::HC * C::PS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x; bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int n0 = blockIdx.y * C::NW;
    const int g = blockIdx.z;

    const float* in = p.in + (size_t)b * p.H * p.W * p.in_cs + p.in_co + g * p.in_gs;
    const float* wg = p.w + (size_t)g * p.w_gs;
    const int iy0 = ty * C::TH * S - C::PAD, ix0 = tx * TW * S - C::PAD;

    float4 ra[C::NA], rb[C::NB];

    auto issue_loads = [&](int c0) {
#pragma unroll
        for (int k = 0; k < C::NA; ++k) {
            const int idx = tid + k * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < C::A_VEC) {
                const int q = idx % C::QC, pix = idx / C::QC;
                const int hx = pix % C::HC, hy = pix / C::HC;
                const int iy = iy0 + hy, ix = ix0 + hx, c = c0 + q * 4;
                if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && c < p.cin_valid)
                    v = ldg4(in + ((size_t)iy * p.W + ix) * p.in_cs + c);
            }
            ra[k] = v;
        }
#pragma unroll
        for (int k = 0; k < C::NB; ++k) {
            const int idx = tid + k * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < C::B_VEC) {
                const int j = idx % C::NW, tq = idx / C::NW;
                const int q = tq % C::QC, tap = tq / C::QC;
                v = ldg4(wg + (((size_t)tap * (p.cin_pad >> 2) + (c0 >> 2) + q) * p.cout_pad + n0 + j) * 4);
            }
            rb[k] = v;
        }
    };
    auto write_lds = [&]() {
#pragma unroll
        for (int k = 0; k < C::NA; ++k) {
            const int idx = tid + k * 256;
            if (idx < C::A_VEC) {
                const int q = idx % C::QC, pix = idx / C::QC;
                *reinterpret_cast<float4*>(sA + pix * C::PS + q * 4) = ra[k];
            }
        }
#pragma unroll
        for (int k = 0; k < C::NB; ++k) {
            const int idx = tid + k * 256;
            if (idx < C::B_VEC) *reinterpret_cast<float4*>(sB + idx * 4) = rb[k];
        }
    };

    // per-wave fragment base addresses
    int aoff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mb = wave * MT + m;
        const int row = mb * C::RPB + li / TW, col = li % TW;
        aoff[m] = ((row * S) * C::HC + col * S) * C::PS + lh * 4;
    }
    const int boff = (lh * C::NW + li) * 4;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int n_chunks = p.cin_pad / CK;
    issue_loads(0);
    for (int ch = 0; ch < n_chunks; ++ch) {
        __syncthreads();                 // everyone finished reading the previous chunk
        write_lds();
        __syncthreads();
        if (ch + 1 < n_chunks) issue_loads((ch + 1) * CK);
#pragma unroll
        for (int tap = 0; tap < C::TAPS; ++tap) {
            const int dy = tap / KS, dx = tap % KS;
#pragma unroll
            for (int q8 = 0; q8 < CK / 8; ++q8) {
                float4 af[MT], bf[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    af[m] = *reinterpret_cast<const float4*>(sA + aoff[m] + (dy * C::HC + dx) * C::PS + q8 * 8);
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    bf[n] = *reinterpret_cast<const float4*>(sB + boff + ((tap * C::QC + q8 * 2) * C::NW + n * 32) * 4);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].x, bf[n].x, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].y, bf[n].y, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].z, bf[n].z, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].w, bf[n].w, acc[m][n], 0, 0, 0);
                    }
            }
        }
    }

    // epilogue: BN scale/shift (+ residual) (+ ReLU); lane owns channel n0+n*32+li
    float* out = p.out + (size_t)b * p.Ho * p.Wo * p.out_cs + p.out_co + g * p.out_gs;
    const float* res = p.res ? p.res + (size_t)b * p.Ho * p.Wo * p.res_cs + p.res_co + g * p.res_gs : nullptr;
    const int oy0 = ty * C::TH, ox0 = tx * TW;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n0 + n * 32 + li;
        const float sc = p.scale[g * p.cout_pad + co], sh = p.shift[g * p.cout_pad + co];
        const bool ok = co < p.Cout;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int mb = wave * MT + m;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int oy = oy0 + mb * C::RPB + pr / TW, ox = ox0 + pr % TW;
                const size_t pix = (size_t)oy * p.Wo + ox;
                float v = fmaf(acc[m][n][r], sc, sh);
                if (ok) {
                    if (res) v += res[pix * p.res_cs + co];
                    if (p.relu) v = fmaxf(v, 0.f);
                    out[pix * p.out_cs + co] = v;
                }
            }
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
        const int pad = KS / 2;
        float acc = 0.f;
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int iy = oy * S - pad + tap / KS, ix = ox * S - pad + tap % KS;
            if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.W) continue;
            const float* px = in + ((size_t)iy * p.W + ix) * p.in_cs;
            for (int c = 0; c < p.cin_valid; ++c)
                acc = fmaf(px[c], wg[(((size_t)tap * (p.cin_pad >> 2) + (c >> 2)) * p.cout_pad + co) * 4 + (c & 3)], acc);
        }
        float v = fmaf(acc, p.scale[g * p.cout_pad + co], p.shift[g * p.cout_pad + co]);
        const size_t pix = ((size_t)b * p.Ho + oy) * p.Wo + ox;
        if (p.res) v += p.res[pix * p.res_cs + p.res_co + g * p.res_gs + co];
        if (p.relu) v = fmaxf(v, 0.f);
        p.out[pix * p.out_cs + p.out_co + g * p.out_gs + co] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------------
typedef void (*conv_fn)(ConvParams);
struct ConvVariant { int ks, s, mt, nt, tw, ck; conv_fn fn; int lds; int th; };

#define ROMP_CONV_VARIANT(KS, S, MT, NT, TW, CK)                                      \
    { KS, S, MT, NT, TW, CK, conv_mfma_kernel<KS, S, MT, NT, TW, CK>,                 \
      ConvCfg<KS, S, MT, NT, TW, CK>::LDS_BYTES, ConvCfg<KS, S, MT, NT, TW, CK>::TH }

static const ConvVariant kVariants[] = {
    // 3x3 stride 1
    ROMP_CONV_VARIANT(3, 1, 2, 1, 32, 16), ROMP_CONV_VARIANT(3, 1, 2, 2, 32, 16),
    ROMP_CONV_VARIANT(3, 1, 1, 1, 32, 16), ROMP_CONV_VARIANT(3, 1, 1, 2, 32, 16),
    ROMP_CONV_VARIANT(3, 1, 1, 1, 16, 16), ROMP_CONV_VARIANT(3, 1, 1, 2, 16, 16),
    ROMP_CONV_VARIANT(3, 1, 2, 2, 16, 16),
    // 3x3 stride 2
    ROMP_CONV_VARIANT(3, 2, 1, 1, 32, 16), ROMP_CONV_VARIANT(3, 2, 1, 2, 32, 16),
    ROMP_CONV_VARIANT(3, 2, 1, 1, 16, 16), ROMP_CONV_VARIANT(3, 2, 1, 2, 16, 16),
    ROMP_CONV_VARIANT(3, 2, 1, 1, 32, 8), ROMP_CONV_VARIANT(3, 2, 1, 2, 32, 8),
    ROMP_CONV_VARIANT(3, 2, 1, 1, 16, 8), ROMP_CONV_VARIANT(3, 2, 1, 2, 16, 8),
    // 1x1
    ROMP_CONV_VARIANT(1, 1, 2, 1, 32, 32), ROMP_CONV_VARIANT(1, 1, 2, 2, 32, 32),
    ROMP_CONV_VARIANT(1, 1, 1, 1, 32, 32), ROMP_CONV_VARIANT(1, 1, 1, 2, 32, 32),
    ROMP_CONV_VARIANT(1, 1, 1, 1, 16, 32), ROMP_CONV_VARIANT(1, 1, 1, 2, 16, 32),
    ROMP_CONV_VARIANT(1, 1, 2, 2, 16, 32),
    ROMP_CONV_VARIANT(1, 1, 2, 1, 32, 16), ROMP_CONV_VARIANT(1, 1, 2, 2, 32, 16),
};
static const int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
static bool g_attr_done = false;

static int ensure_attrs() {
    if (g_attr_done) return ROMP_OK;
    for (int i = 0; i < kNumVariants; ++i)
        if (kVariants[i].lds > 48 * 1024)
            ROMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kVariants[i].fn),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kVariants[i].lds));
    g_attr_done = true;
    return ROMP_OK;
}

// Pick the variant: widest tile that divides the output map, then the largest per-workgroup
// footprint that still yields >= ~2 workgroups per CU at this batch size.
static const ConvVariant* choose_variant(const romp_op& op, int Ho, int Wo, int B) {
    const ConvVariant* best = nullptr;
    double best_score = -1;
    for (int i = 0; i < kNumVariants; ++i) {
        const ConvVariant& v = kVariants[i];
        if (v.ks != op.ksize || v.s != op.stride) continue;
        if (Wo % v.tw || Ho % v.th) continue;
        if (op.cin_pad % v.ck || op.cout_pad % (v.nt * 32)) continue;
        const long wgs = (long)B * (Ho / v.th) * (Wo / v.tw) * (op.cout_pad / (v.nt * 32)) * op.groups;
        // work per WG ~ mt*nt ; prefer big tiles while the grid still fills 256 CUs twice
        double fill = wgs >= 512 ? 1.0 : (double)wgs / 512.0;
        double score = fill * (1.0 + 0.25 * (v.mt * v.nt - 1)) * (v.ck >= 16 ? 1.0 : 0.8) * (v.tw == 32 ? 1.05 : 1.0);
        if (score > best_score) { best_score = score; best = &v; }
    }
    return best;
}

int launch_conv(const romp_op& op, const float* in, const float* res, float* out, int B, int mode,
                hipStream_t st) {
    ROMP_REQUIRE(op.ksize == 1 || op.ksize == 3, "conv: ksize %d unsupported", op.ksize);
    ROMP_REQUIRE(op.stride == 1 || op.stride == 2, "conv: stride %d unsupported", op.stride);
    ROMP_REQUIRE(op.groups >= 1, "conv: groups must be >= 1");
    ROMP_REQUIRE((op.in_cstride & 3) == 0 && (op.in_coff & 3) == 0 && (op.in_gstride & 3) == 0 && (op.Cin & 3) == 0,
                 "conv: input channels must be float4 aligned (cs %d co %d Cin %d)", op.in_cstride, op.in_coff, op.Cin);
    ConvParams p;
    p.in = in; p.w = op.weight; p.scale = op.scale; p.shift = op.shift; p.res = res; p.out = out;
    p.H = op.H; p.W = op.W;
    p.Ho = (op.H + 2 * (op.ksize / 2) - op.ksize) / op.stride + 1;
    p.Wo = (op.W + 2 * (op.ksize / 2) - op.ksize) / op.stride + 1;
    p.Cout = op.Cout; p.cin_valid = op.Cin; p.cin_pad = op.cin_pad; p.cout_pad = op.cout_pad;
    p.in_cs = op.in_cstride; p.in_co = op.in_coff; p.in_gs = op.in_gstride;
    p.out_cs = op.out_cstride; p.out_co = op.out_coff; p.out_gs = op.out_gstride;
    p.res_cs = op.res_cstride; p.res_co = op.res_coff; p.res_gs = op.res_gstride;
    p.relu = op.relu;
    p.w_gs = op.ksize * op.ksize * op.cin_pad * op.cout_pad;
    p.tiles_x = p.tiles_y = 1;
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
    const ConvVariant* v = choose_variant(op, p.Ho, p.Wo, B);
    ROMP_REQUIRE(v != nullptr, "conv: no kernel variant for k%d s%d Cin %d(pad %d) Cout %d(pad %d) out %dx%d",
                 op.ksize, op.stride, op.Cin, op.cin_pad, op.Cout, op.cout_pad, p.Ho, p.Wo);
    p.tiles_x = p.Wo / v->tw;
    p.tiles_y = p.Ho / v->th;
    dim3 grid((unsigned)(B * p.tiles_x * p.tiles_y), (unsigned)(op.cout_pad / (v->nt * 32)), (unsigned)op.groups);
    hipLaunchKernelGGL(v->fn, grid, dim3(256), v->lds, st, p);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

int describe_conv(const romp_op& op, int B, char* out, int n) {
    const int Ho = (op.H + 2 * (op.ksize / 2) - op.ksize) / op.stride + 1;
    const int Wo = (op.W + 2 * (op.ksize / 2) - op.ksize) / op.stride + 1;
    const ConvVariant* v = choose_variant(op, Ho, Wo, B);
    ROMP_REQUIRE(v != nullptr, "describe: no variant");
    snprintf(out, n, "conv_mfma_k%ds%d_mt%d_nt%d_tw%d_ck%d", v->ks, v->s, v->mt, v->nt, v->tw, v->ck);
    return ROMP_OK;
}

}  // namespace romp
