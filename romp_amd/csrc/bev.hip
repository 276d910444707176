// bev.hip -- BEV head pieces that are not plain 2-D convolutions (BASELINE config 4).
//
// Reference: simple_romp/bev/model.py  BEVv1.fv_conditioned_bv_estimation :188-197,
// coarse2fine_localization :199-215 (BasicBlock_3D refiners :52-75,184-186),
// mesh_parameter_regression :225-230 (+ :89-102, :217-223, transformer MLP :131-140);
// simple_romp/bev/post_parser.py  CenterMap3D.parse_3dcentermap :44-66, pack_params_dict :240-253,
// denormalize_cam_params_to_trans :109-128.
//
//   bev_pack_kernel     front-view maps (B,128,128,{4,16}) NHWC -> Conv1d input (B, W=128, (c,h)=2560)
//   bev_maps_kernel     center_map_3d = center_fv (x) center_bv ; cam_maps_3d = coordmap + offsets
//   conv3d_kernel<C>    3x3x3 conv on 1 / 3 channels (NCDHW volumes 64x128x128): LDS slice ring, marches along D
//   bev_nms_kernel + bev_rank_kernel + bev_compact_kernel   MaxPool3d(5) NMS + ordered top-K
//   bev_regress_kernel  per person: cam gather, anchor argmin, feature + depth embedding, 3-layer MLP,
//                       rot6D -> axis-angle, SMPL-A betas, camera -> translation
// The 2-D convs of the head and the Conv1d stack run on the MFMA conv kernel (conv_mfma.hip).
#include "common.h"
#include "rot6d.h"
#include <vector>

namespace romp {

constexpr int BM = 128, BD = 64, BVOX = BD * BM * BM;   // map size, depth levels, voxels per image
constexpr int BEV_CAP = 32768;                           // candidate capacity per image in the 3-D parse (a 5^3 NMS leaves fewer maxima)

// out[b][w][c*128 + h]: c<4 from maps_fv (B,128,128,4), c>=4 from img_feats (B,128,128,16)   (model.py:190)
__global__ void bev_pack_kernel(const float* __restrict__ fv, int fv_cs, const float* __restrict__ feats, int f_cs,
                                float* __restrict__ out, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int h = r % BM; r /= BM;
        const int c = r % 20; r /= 20;
        const int w = r % BM;
        const int b = r / BM;
        const size_t pix = ((size_t)b * BM + h) * BM + w;
        out[i] = c < 4 ? fv[pix * fv_cs + c] : feats[pix * f_cs + (c - 4)];
    }
}

struct Anchors { float a[BD]; };

// center3d[b][d][h][w] = center_fv[b,h,w] * center_bv[b,d,w]                       (model.py:195-196)
// cam3d[b][c][d][h][w] = coord(c; d,h,w) + cam_off[b,c,h,w] (+ cam_off_bv[b,d,w] on c == 2)  (:209-212)
// fv: (B,128,128,4) NHWC [center, off0, off1, off2]; bv: (B, W=128, 128) [w][0..63 center_bv | 64..127 offset_bv]
__global__ void bev_maps_kernel(const float* __restrict__ fv, int fv_cs, const float* __restrict__ bv, int bv_cs,
                                Anchors an, float* __restrict__ center3d, float* __restrict__ cam3d, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int w = r % BM; r /= BM;
        const int h = r % BM; r /= BM;
        const int d = r % BD;
        const int b = r / BD;
        const float* f = fv + (((size_t)b * BM + h) * BM + w) * fv_cs;
        const float* v = bv + ((size_t)b * BM + w) * bv_cs;
        center3d[i] = f[0] * v[d];
        float* cm = cam3d + (size_t)b * 3 * BVOX + ((size_t)d * BM + h) * BM + w;
        cm[0] = an.a[d] + f[1];
        cm[BVOX] = ((float)h / BM * 2.f - 1.f) + f[2];
        cm[2 * (size_t)BVOX] = (((float)w / BM * 2.f - 1.f) + f[3]) + v[BD + d];
    }
}

struct Conv3dParams {
    const float* in; const float* res; float* out;
    float w[3 * 3 * 27]; float scale[3]; float shift[3];
    int relu;
};

// 3x3x3 convolution + folded BN (+ residual) (+ ReLU) on NCDHW volumes (D=64, H=128, W=128), C in {1,3}
// channels in and out (BasicBlock_3D, bev/model.py:52-75).  Arithmetic intensity 2*27*C FLOP per 4-byte
// voxel read: VALU/HBM balanced, far too thin for the matrix pipe.  A workgroup owns a (T3_D x T3_H x 128)
// output brick and marches along D with a ring of three haloed input slices in LDS (one new slice per
// step, all C channels); a thread produces 4 consecutive voxels along W for every output channel from
// float4 + float2 row-segment reads (conflict-free), weights come in as scalar operands from the kernel
// arguments.  Loads and stores are full 512-byte rows.
constexpr int T3_H = 8, T3_D = 16, T3_RW = BM + 8;            // row pitch in floats: 4 left pad (halo at [3]) + 128 + 4

template <int C>
__global__ __launch_bounds__(256) void conv3d_kernel(Conv3dParams p, int B) {
    constexpr int SLICE = C * (T3_H + 2) * T3_RW;               // floats per ring slot
    __shared__ __attribute__((aligned(16))) float ring[3 * SLICE];
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int hb = t % (BM / T3_H); t /= (BM / T3_H);
    const int db = t % (BD / T3_D);
    const int b = t / (BD / T3_D);
    const int h0 = hb * T3_H, d0 = db * T3_D;
    const float* inb = p.in + (size_t)b * C * BVOX;
    // zero the pads once (columns -1 and 128 of every row stay zero: the brick spans the full W)
    for (int i = tid; i < 3 * SLICE; i += 256) ring[i] = 0.f;
    __syncthreads();
    auto load_slice = [&](int d, int slot) {                     // slice d (may be outside: zeros), rows h0-1 .. h0+T3_H
        float* dst = ring + slot * SLICE;
        constexpr int NE = C * (T3_H + 2) * (BM / 4), NL = (NE + 255) / 256;
        float4 raw[NL];                                          // branch-free, all loads of the slice in flight together
        bool ok[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = min(tid + k * 256, NE - 1);
            const int q = i % (BM / 4), r = i / (BM / 4);
            const int row = r % (T3_H + 2), ci = r / (T3_H + 2);
            const int h = h0 - 1 + row;
            ok[k] = (unsigned)d < (unsigned)BD && (unsigned)h < (unsigned)BM;
            raw[k] = *reinterpret_cast<const float4*>(inb + (ok[k] ? (size_t)ci * BVOX + ((size_t)d * BM + h) * BM + q * 4 : 0));
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = tid + k * 256;
            if (i < NE) {
                const int q = i % (BM / 4), r = i / (BM / 4);
                const int row = r % (T3_H + 2), ci = r / (T3_H + 2);
                *reinterpret_cast<float4*>(dst + (ci * (T3_H + 2) + row) * T3_RW + 4 + q * 4) = ok[k] ? raw[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    load_slice(d0 - 1, 0);
    load_slice(d0, 1);
    const int q = tid & 31, hr = tid >> 5;                       // 4 voxels w = 4q..4q+3 of row h0 + hr
    for (int dd = 0; dd < T3_D; ++dd) {
        load_slice(d0 + dd + 1, (dd + 2) % 3);
        __syncthreads();
        float acc[C][4];
#pragma unroll
        for (int co = 0; co < C; ++co)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[co][j] = 0.f;
#pragma unroll 1
        for (int ci = 0; ci < C; ++ci)
#pragma unroll 1
            for (int dz = 0; dz < 3; ++dz) {
                const float* sl = ring + ((dd + dz) % 3) * SLICE + (ci * (T3_H + 2) + hr) * T3_RW + q * 4;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float* row = sl + dy * T3_RW;
                    const float xm = row[3];                                         // w = 4q - 1
                    const float4 x4 = *reinterpret_cast<const float4*>(row + 4);     // w = 4q .. 4q+3
                    const float xp = row[8];                                         // w = 4q + 4
                    const float x[6] = {xm, x4.x, x4.y, x4.z, x4.w, xp};
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int co = 0; co < C; ++co) {
                            const float wv = p.w[(co * C + ci) * 27 + (dz * 3 + dy) * 3 + dx];
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[co][j] = fmaf(x[j + dx], wv, acc[co][j]);
                        }
                }
            }
        const int d = d0 + dd, h = h0 + hr;
#pragma unroll
        for (int co = 0; co < C; ++co) {
            const size_t o = (size_t)b * C * BVOX + (size_t)co * BVOX + ((size_t)d * BM + h) * BM + q * 4;
            float4 v;
            v.x = fmaf(acc[co][0], p.scale[co], p.shift[co]);
            v.y = fmaf(acc[co][1], p.scale[co], p.shift[co]);
            v.z = fmaf(acc[co][2], p.scale[co], p.shift[co]);
            v.w = fmaf(acc[co][3], p.scale[co], p.shift[co]);
            if (p.res) {
                const float4 r = *reinterpret_cast<const float4*>(p.res + o);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(p.out + o) = v;
        }
        __syncthreads();                                         // everyone done with slot dd%3 before it is reloaded
    }
}

int launch_bev_pack(const float* fv, int fv_cs, const float* feats, int f_cs, float* out, int B, hipStream_t st) {
    const size_t total = (size_t)B * BM * 20 * BM;
    hipLaunchKernelGGL(bev_pack_kernel, dim3(2048), dim3(256), 0, st, fv, fv_cs, feats, f_cs, out, total);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

int launch_bev_maps(const float* fv, int fv_cs, const float* bv, int bv_cs, const float* anchors_host, float* center3d,
                    float* cam3d, int B, hipStream_t st) {
    Anchors an;
    for (int i = 0; i < BD; ++i) an.a[i] = anchors_host[i];
    const size_t total = (size_t)B * BVOX;
    hipLaunchKernelGGL(bev_maps_kernel, dim3(4096), dim3(256), 0, st, fv, fv_cs, bv, bv_cs, an, center3d, cam3d, total);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

int launch_conv3d(int C, const float* w_host, const float* scale_host, const float* shift_host, int relu, const float* in,
                  const float* res, float* out, int B, hipStream_t st) {
    ROMP_REQUIRE(C == 1 || C == 3, "conv3d: %d channels unsupported", C);
    Conv3dParams p;
    p.in = in; p.res = res; p.out = out; p.relu = relu;
    for (int i = 0; i < C * C * 27; ++i) p.w[i] = w_host[i];
    for (int i = 0; i < C; ++i) { p.scale[i] = scale_host[i]; p.shift[i] = shift_host[i]; }
    const int grid = B * (BD / T3_D) * (BM / T3_H);
    if (C == 1) hipLaunchKernelGGL(conv3d_kernel<1>, dim3(grid), dim3(256), 0, st, p, B);
    else hipLaunchKernelGGL(conv3d_kernel<3>, dim3(grid), dim3(256), 0, st, p, B);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

// ---- 3-D center parse ---------------------------------------------------------------------------
// ws per image: [1] candidate counter, [1] kept count, [BEV_CAP] flat index, [BEV_CAP] score bits,
//               [max_person] top flat, [max_person] top score bits
__device__ __forceinline__ int ws_stride(int max_person) { return 2 + 2 * BEV_CAP + 2 * max_person; }

__global__ __launch_bounds__(256) void bev_nms_kernel(const float* __restrict__ c3d, float thresh, int max_person,
                                                       int32_t* __restrict__ ws, int B) {
    const size_t total = (size_t)B * BVOX;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float v = c3d[i];
        if (!(v > thresh)) continue;                       // thresh > 0: suppressed voxels (det*0) never pass
        size_t r = i;
        const int w = r % BM; r /= BM;
        const int h = r % BM; r /= BM;
        const int d = r % BD;
        const int b = r / BD;
        const float* vol = c3d + (size_t)b * BVOX;
        bool is_max = true;                                // MaxPool3d(5,1,2): keep where max == value
        for (int dz = -2; dz <= 2 && is_max; ++dz) {
            const int zz = d + dz;
            if ((unsigned)zz >= (unsigned)BD) continue;
            for (int dy = -2; dy <= 2 && is_max; ++dy) {
                const int yy = h + dy;
                if ((unsigned)yy >= (unsigned)BM) continue;
                for (int dx = -2; dx <= 2; ++dx) {
                    const int xx = w + dx;
                    if ((unsigned)xx >= (unsigned)BM) continue;
                    if (vol[((size_t)zz * BM + yy) * BM + xx] > v) { is_max = false; break; }
                }
            }
        }
        if (!is_max) continue;
        int32_t* wsb = ws + (size_t)b * ws_stride(max_person);
        const int slot = atomicAdd(wsb, 1);
        if (slot < BEV_CAP) {
            wsb[2 + slot] = (int)(i - (size_t)b * BVOX);
            wsb[2 + BEV_CAP + slot] = __float_as_int(v);
        }
    }
}

__global__ __launch_bounds__(256) void bev_rank_kernel(int max_person, int32_t* __restrict__ ws) {
    int32_t* wsb = ws + (size_t)blockIdx.x * ws_stride(max_person);
    const int n = min(wsb[0], BEV_CAP);
    const int32_t* flat = wsb + 2;
    const int32_t* sc = wsb + 2 + BEV_CAP;
    int32_t* top = wsb + 2 + 2 * BEV_CAP;
    for (int c = threadIdx.x; c < n; c += 256) {
        const float s = __int_as_float(sc[c]);
        const int id = flat[c];
        int rank = 0;
        for (int k = 0; k < n; ++k) {
            const float so = __int_as_float(sc[k]);
            rank += (so > s) || (so == s && flat[k] < id);
        }
        if (rank < max_person) { top[rank] = id; top[max_person + rank] = sc[c]; }
    }
    if (threadIdx.x == 0) wsb[1] = n < max_person ? n : max_person;
}

__global__ void bev_compact_kernel(int B, int max_person, const int32_t* __restrict__ ws, int32_t* batch_ids,
                                   int32_t* czyx, float* confs) {
    const int r = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int st = ws_stride(max_person);
    if (r >= ws[(size_t)b * st + 1]) return;
    int off = 0;
    for (int k = lane; k < b; k += 64) off += ws[(size_t)k * st + 1];
    for (int d = 32; d > 0; d >>= 1) off += __shfl_xor(off, d);
    if (lane == 0) {
        const int row = off + r;
        const int32_t* top = ws + (size_t)b * st + 2 + 2 * BEV_CAP;
        const int flat = top[r];
        batch_ids[row] = b;
        czyx[row * 3 + 0] = flat / (BM * BM);
        czyx[row * 3 + 1] = (flat / BM) % BM;
        czyx[row * 3 + 2] = flat % BM;
        confs[row] = __int_as_float(top[max_person + r]);
    }
}

// ---- per-person regression ------------------------------------------------------------------------
struct MlpWeights { const float *emb, *w1t, *b1, *w2t, *b2, *w3t, *b3; };

__global__ __launch_bounds__(256) void bev_regress_kernel(
    const float* __restrict__ cam3d, const float* __restrict__ feat, int feat_cs, const int32_t* __restrict__ batch_ids,
    const int32_t* __restrict__ czyx, Anchors an, MlpWeights m, float* params_pred, int32_t* cam_czyx, float* cam_out,
    float* thetas, float* betas, float* cam_trans) {
    __shared__ float s_x[128], s_h1[512], s_h2[512], s_p[148];
    __shared__ int s_c[3];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int b = batch_ids[row], z = czyx[row * 3], y = czyx[row * 3 + 1], x = czyx[row * 3 + 2];
    if (tid < 3) s_p[tid] = cam3d[((size_t)b * 3 + tid) * BVOX + ((size_t)z * BM + y) * BM + x];   // model.py:242
    __syncthreads();
    if (tid == 0) {
        // convert_cam_params_to_centermap_coords + denormalize_center (model.py:89-102)
        int k = 0;
        float best = fabsf(s_p[0] - an.a[0]);
        for (int i = 1; i < BD; ++i) {
            const float dd = fabsf(s_p[0] - an.a[i]);
            if (dd < best) { best = dd; k = i; }
        }
        float cz = ((float)k / 128.f * 2.f - 1.f + 1.f) / 2.f * BM;
        float cy = (s_p[1] + 1.f) / 2.f * BM, cx = (s_p[2] + 1.f) / 2.f * BM;
        cz = fminf(fmaxf(cz, 1.f), BM - 1.f); cy = fminf(fmaxf(cy, 1.f), BM - 1.f); cx = fminf(fmaxf(cx, 1.f), BM - 1.f);
        s_c[0] = (int)cz; s_c[1] = (int)cy; s_c[2] = (int)cx;
        cam_czyx[row * 3] = s_c[0]; cam_czyx[row * 3 + 1] = s_c[1]; cam_czyx[row * 3 + 2] = s_c[2];
    }
    __syncthreads();
    if (tid < 128)                                          // feature + depth embedding (model.py:217-223)
        s_x[tid] = feat[(((size_t)b * BM + s_c[1]) * BM + s_c[2]) * feat_cs + tid] + m.emb[s_c[0] * 128 + tid];
    __syncthreads();
    for (int o = tid; o < 512; o += 256) {                   // Linear(128,512) + ReLU
        float a = m.b1[o];
        for (int i = 0; i < 128; ++i) a = fmaf(s_x[i], m.w1t[i * 512 + o], a);
        s_h1[o] = fmaxf(a, 0.f);
    }
    __syncthreads();
    for (int o = tid; o < 512; o += 256) {                   // Linear(512,512) + ReLU
        float a = m.b2[o];
        for (int i = 0; i < 512; ++i) a = fmaf(s_h1[i], m.w2t[i * 512 + o], a);
        s_h2[o] = fmaxf(a, 0.f);
    }
    __syncthreads();
    if (tid < 143) {                                        // Linear(512,143)
        float a = m.b3[tid];
        for (int i = 0; i < 512; ++i) a = fmaf(s_h2[i], m.w3t[i * 143 + tid], a);
        s_p[3 + tid] = a;
    }
    __syncthreads();
    for (int c = tid; c < 146; c += 256) params_pred[(size_t)row * 146 + c] = s_p[c];
    // pack_params_dict (post_parser.py:240-253): cam 3 | global_orient 6 | body_pose 126 | betas 11
    if (tid < 3) cam_out[row * 3 + tid] = s_p[tid];
    if (tid < 11) betas[row * 11 + tid] = s_p[135 + tid];
    if (tid < 22) {
        float aa[3];
        rot6d_to_aa_dev(s_p + 3 + tid * 6, aa);
        thetas[row * 72 + tid * 3] = aa[0]; thetas[row * 72 + tid * 3 + 1] = aa[1]; thetas[row * 72 + tid * 3 + 2] = aa[2];
    } else if (tid < 28) {
        thetas[row * 72 + 66 + (tid - 22)] = 0.f;
    }
    if (tid == 0) {                                         // denormalize_cam_params_to_trans (post_parser.py:114-128)
        const float tan_fov = 0.57735026918962573f;
        const float depth = 1.f / (s_p[0] * tan_fov + 1e-3f);
        cam_trans[row * 3 + 0] = s_p[2] * depth * tan_fov;
        cam_trans[row * 3 + 1] = s_p[1] * depth * tan_fov;
        cam_trans[row * 3 + 2] = depth;
    }
}

}  // namespace romp

using namespace romp;

extern "C" {

int romp_bev_workspace_ints(int B, int max_person) { return B * (2 + 2 * BEV_CAP + 2 * max_person); }

int romp_bev_parse(const float* center_maps_3d, int B, float conf_thresh, int max_person, int32_t* count_host,
                   int32_t* batch_ids, int32_t* czyx, float* confs, int32_t* workspace, void* stream) {
    ROMP_REQUIRE(center_maps_3d && count_host && batch_ids && czyx && confs && workspace && B > 0, "romp_bev_parse: bad arguments");
    ROMP_REQUIRE(max_person >= 1 && max_person <= 1024, "romp_bev_parse: max_person %d out of range", max_person);
    ROMP_REQUIRE(conf_thresh > 0.f, "romp_bev_parse: conf_thresh must be > 0 (the reference's NMS zeroes non-maxima)");
    hipStream_t st = (hipStream_t)stream;
    const int stride = 2 + 2 * BEV_CAP + 2 * max_person;
    ROMP_HIP_CHECK(hipMemsetAsync(workspace, 0, (size_t)B * stride * sizeof(int32_t), st));
    hipLaunchKernelGGL(bev_nms_kernel, dim3(8192), dim3(256), 0, st, center_maps_3d, conf_thresh, max_person, workspace, B);
    hipLaunchKernelGGL(bev_rank_kernel, dim3(B), dim3(256), 0, st, max_person, workspace);
    hipLaunchKernelGGL(bev_compact_kernel, dim3(max_person, B), dim3(64), 0, st, B, max_person, workspace, batch_ids, czyx, confs);
    ROMP_HIP_CHECK(hipGetLastError());
    std::vector<int32_t> head((size_t)B * 2);
    ROMP_HIP_CHECK(hipMemcpy2DAsync(head.data(), 2 * sizeof(int32_t), workspace, (size_t)stride * sizeof(int32_t),
                                    2 * sizeof(int32_t), B, hipMemcpyDeviceToHost, st));
    ROMP_HIP_CHECK(hipStreamSynchronize(st));
    int total = 0;
    for (int b = 0; b < B; ++b) {
        if (head[(size_t)b * 2] > BEV_CAP) {
            set_error("romp_bev_parse: image %d has %d NMS maxima above the threshold (capacity %d)", b, head[(size_t)b * 2], BEV_CAP);
            return ROMP_ECAPACITY;
        }
        total += head[(size_t)b * 2 + 1];
    }
    *count_host = total;
    return ROMP_OK;
}

int romp_bev_regress(const float* cam_maps_3d, const float* fv_features, int feat_cstride, int N, const int32_t* batch_ids,
                     const int32_t* czyx, const float* anchors_host, const float* emb, const float* w1t, const float* b1,
                     const float* w2t, const float* b2, const float* w3t, const float* b3, float* params_pred,
                     int32_t* cam_czyx, float* cam, float* thetas, float* betas, float* cam_trans, void* stream) {
    ROMP_REQUIRE(cam_maps_3d && fv_features && batch_ids && czyx && anchors_host && emb && w1t && b1 && w2t && b2 && w3t && b3 &&
                 params_pred && cam_czyx && cam && thetas && betas && cam_trans && N >= 0, "romp_bev_regress: bad arguments");
    if (N == 0) return ROMP_OK;
    Anchors an;
    for (int i = 0; i < BD; ++i) an.a[i] = anchors_host[i];
    MlpWeights m{emb, w1t, b1, w2t, b2, w3t, b3};
    hipLaunchKernelGGL(bev_regress_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, cam_maps_3d, fv_features, feat_cstride,
                       batch_ids, czyx, an, m, params_pred, cam_czyx, cam, thetas, betas, cam_trans);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

}  // extern "C"
