// parse.hip -- center-map parsing and parameter packing (seam #2) + projection (seam #4).
//
// Reference: CenterMap.parse_centermap / nms / gather_feature (post_parser.py:27-64),
// parameter_sampling + pack_params_dict + parsing_outputs (post_parser.py:66-79,128-146),
// params_maps[:,0] = 1.1**params_maps[:,0] (main.py:113),
// rot6D_to_angular -> rot6d_to_rotmat -> rotation_matrix_to_quaternion -> quaternion_to_angle_axis
// (utils.py:471-491, 535-682), batch_orth_proj (utils.py:309-315),
// convert_proejection_from_input_to_orgimg (post_parser.py:81-88), convert_cam_to_3d_trans
// (utils.py:303-307).
//
// The reference runs ~12 small kernels + a host sync for the parse, a full (B,145,64,64) transpose
// copy to sample N rows, and ~60 elementwise kernels for the rotation conversion.  Here: one
// workgroup per image does the 5x5 max-NMS, threshold and ordered top-K out of LDS; one wave per
// detection gathers its 580-byte NHWC parameter row (contiguous -- no transpose) and converts the
// 22 rotations.  Everything is latency-bound; bytes moved = 16 KB + N_b*580 B per image.
#include "common.h"
#include "rot6d.h"
#include <vector>

#pragma clang fp contract(off)   // keep the reference's rounding in the ill-conditioned branches

namespace romp {

constexpr int MAP = 64, NPIX = MAP * MAP;

// ws layout per image: [max_person] flat index, [max_person] score bits, [1] count, [1] total candidates
constexpr int PARSE_THREADS = 1024;     // one workgroup per image; 4 pixels per thread (256 threads: 20 us on the single-image critical path)
__global__ __launch_bounds__(PARSE_THREADS) void parse_nms_topk_kernel(const float* __restrict__ center, float thresh,
                                                              int max_person, int32_t* __restrict__ ws) {
    __shared__ float s_map[NPIX];
    __shared__ float s_score[NPIX];
    __shared__ int s_idx[NPIX];
    __shared__ int s_count;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* cm = center + (size_t)b * NPIX;
    for (int p = tid; p < NPIX; p += PARSE_THREADS) s_map[p] = cm[p];
    if (tid == 0) s_count = 0;
    __syncthreads();
    for (int p = tid; p < NPIX; p += PARSE_THREADS) {
        const int y = p / MAP, x = p % MAP;
        const float v = s_map[p];
        float m = v;                                   // MaxPool2d(5,1,2): implicit -inf padding
        for (int dy = -2; dy <= 2; ++dy) {
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)MAP) continue;
            for (int dx = -2; dx <= 2; ++dx) {
                const int xx = x + dx;
                if ((unsigned)xx >= (unsigned)MAP) continue;
                m = fmaxf(m, s_map[yy * MAP + xx]);
            }
        }
        const float score = (m == v) ? v : v * 0.0f;   // det * (maxm == det).float()
        if (score > thresh) {
            const int slot = atomicAdd(&s_count, 1);
            s_score[slot] = score;
            s_idx[slot] = p;
        }
    }
    __syncthreads();
    const int n = s_count;
    int32_t* w = ws + (size_t)b * (2 * max_person + 2);
    for (int c = tid; c < n; c += PARSE_THREADS) {
        const float sc = s_score[c];
        const int id = s_idx[c];
        int rank = 0;
        for (int k = 0; k < n; ++k) {
            const float so = s_score[k];
            rank += (so > sc) || (so == sc && s_idx[k] < id);
        }
        if (rank < max_person) {
            w[rank] = id;
            w[max_person + rank] = __float_as_int(sc);
        }
    }
    if (tid == 0) {
        w[2 * max_person] = n < max_person ? n : max_person;
        w[2 * max_person + 1] = n;
    }
}

__global__ __launch_bounds__(64) void parse_pack_kernel(
    const float* __restrict__ center, const float* __restrict__ params, int B, int max_person,
    int32_t* __restrict__ ws, int32_t* batch_ids, int32_t* flat_inds, float* scores, float* params_pred,
    float* cam, float* thetas, float* betas, int32_t* center_preds, const int32_t* watch) {
    const int r = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int stride = 2 * max_person + 2;
    // romp_parse_watch: the watched word (the net's saturation counter, bumped by agent-scope atomics of kernels that may still be
    // running on another stream) rides behind the counts; nobody else touches that workspace slot
    if (watch && r == 0 && b == 0 && lane == 0)
        ws[(size_t)B * stride] = __hip_atomic_load(watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int cnt = ws[(size_t)b * stride + 2 * max_person];
    if (r >= cnt) return;
    int off = 0;
    for (int k = lane; k < b; k += 64) off += ws[(size_t)k * stride + 2 * max_person];
    for (int d = 32; d > 0; d >>= 1) off += __shfl_xor(off, d);
    const int row = off + r;
    const int flat = ws[(size_t)b * stride + r];
    const float sc = __int_as_float(ws[(size_t)b * stride + max_person + r]);
    __shared__ float s_p[148];
    const float* src = params + ((size_t)b * NPIX + flat) * 145;
    for (int c = lane; c < 145; c += 64) {
        float v = src[c];
        if (c == 0) v = powf(1.1f, v);                 // main.py:113
        s_p[c] = v;
        params_pred[(size_t)row * 145 + c] = v;
    }
    __syncthreads();
    if (lane < 3) cam[row * 3 + lane] = s_p[lane];
    if (lane < 10) betas[row * 10 + lane] = s_p[135 + lane];
    if (lane < 22) {
        float aa[3];
        rot6d_to_aa_dev(s_p + 3 + lane * 6, aa);       // joint 0 = global_orient, 1..21 = body_pose
        thetas[row * 72 + lane * 3 + 0] = aa[0];
        thetas[row * 72 + lane * 3 + 1] = aa[1];
        thetas[row * 72 + lane * 3 + 2] = aa[2];
    } else if (lane < 28) {
        thetas[row * 72 + 66 + (lane - 22)] = 0.f;     // two hand joints padded with zeros (post_parser.py:76)
    }
    if (lane == 0) {
        batch_ids[row] = b;
        flat_inds[row] = flat;
        scores[row] = sc;                              // == center_confs (post_parser.py:145)
        center_preds[row * 2 + 0] = (flat % MAP) * 512 / 64;
        center_preds[row * 2 + 1] = (flat / MAP) * 512 / 64;
    }
}

__global__ void rot6d_kernel(const float* __restrict__ x6, int n, float* __restrict__ aa) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x[6], o[3];
    for (int k = 0; k < 6; ++k) x[k] = x6[(size_t)i * 6 + k];
    rot6d_to_aa_dev(x, o);
    for (int k = 0; k < 3; ++k) aa[(size_t)i * 3 + k] = o[k];
}

__global__ void project_kernel(const float* __restrict__ joints, int N, int J, const float* __restrict__ cam,
                               float pad_size, float left, float top, float* pj2d, float* pj2d_org,
                               float* cam_trans) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N * J) {
        const int n = i / J;
        const float s = cam[n * 3], tx = cam[n * 3 + 1], ty = cam[n * 3 + 2];
        const float px = joints[(size_t)i * 3] * s + tx, py = joints[(size_t)i * 3 + 1] * s + ty;
        pj2d[(size_t)i * 2] = px; pj2d[(size_t)i * 2 + 1] = py;
        pj2d_org[(size_t)i * 2] = (px + 1.f) * pad_size / 2.f - left;
        pj2d_org[(size_t)i * 2 + 1] = (py + 1.f) * pad_size / 2.f - top;
    }
    if (i < N) {
        const float s = cam[i * 3], tx = cam[i * 3 + 1], ty = cam[i * 3 + 2];
        cam_trans[i * 3 + 0] = (tx / s) * 2.f;
        cam_trans[i * 3 + 1] = (ty / s) * 2.f;
        cam_trans[i * 3 + 2] = (1.f / s) * 2.f;
    }
}

// convert_cam_to_3d_trans (utils.py:303-307): (s, tx, ty) -> (tx / s, ty / s, 1 / s) * weight
__global__ void cam_to_trans_kernel(const float* __restrict__ cam, int N, float weight, float* __restrict__ trans) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float s = cam[i * 3], tx = cam[i * 3 + 1], ty = cam[i * 3 + 2];
    trans[i * 3 + 0] = (tx / s) * weight;
    trans[i * 3 + 1] = (ty / s) * weight;
    trans[i * 3 + 2] = (1.f / s) * weight;
}

// Camera translation that best explains the orthographic projection under the perspective camera: the reference's
// estimate_translation with OpenCV absent (utils.py:391-434 -> estimate_translation_np :347-389, unit weights) over the
// first K joints.  A joint counts when its pixel ROW coordinate is > -2 (`joints_conf = joints_2d[:, :, -1] > -2.`, :405-406,
// reads the last coordinate of a 2-column array) and its depth is not the -2 sentinel; fewer than 4 -> INVALID_TRANS (-1).
// Rows (f, 0, cx - u | (u - cx) Z - f X) and (0, f, cy - v | (v - cy) Z - f Y); the 3x3 normal equations in double.
// One 64-lane workgroup per person.
__global__ __launch_bounds__(64) void translation_lsq_kernel(const float* __restrict__ joints, int J, int K, const float* __restrict__ pj2d,
                                                               double f, double cx, double cy, float* __restrict__ trans) {
    const int n = blockIdx.x, lane = threadIdx.x;
    double a02 = 0, a12 = 0, a22 = 0, b0 = 0, b1 = 0, b2 = 0;
    int cnt = 0;
    for (int k = lane; k < K; k += 64) {
        const float* X = joints + ((size_t)n * J + k) * 3;
        const float uf = (pj2d[((size_t)n * J + k) * 2] + 1.f) * (float)cx, vf = (pj2d[((size_t)n * J + k) * 2 + 1] + 1.f) * (float)cy;   // float32, like (pj + 1) * 256
        if (!(vf > -2.f) || X[2] == -2.f) continue;
        ++cnt;
        const double u = uf, v = vf;
        const double qx = cx - u, qy = cy - v;
        const double rx = (u - cx) * (double)X[2] - f * (double)X[0], ry = (v - cy) * (double)X[2] - f * (double)X[1];
        a02 += f * qx; a12 += f * qy; a22 += qx * qx + qy * qy;
        b0 += f * rx; b1 += f * ry; b2 += qx * rx + qy * ry;
    }
    for (int d = 32; d > 0; d >>= 1) {
        a02 += __shfl_xor(a02, d); a12 += __shfl_xor(a12, d); a22 += __shfl_xor(a22, d);
        b0 += __shfl_xor(b0, d); b1 += __shfl_xor(b1, d); b2 += __shfl_xor(b2, d);
        cnt += __shfl_xor(cnt, d);
    }
    if (lane == 0) {
        // A = [[d, 0, a02], [0, d, a12], [a02, a12, a22]] with d = count * f^2: eliminate the first two unknowns
        const double d = (double)cnt * f * f;
        const double piv = a22 - (a02 * a02 + a12 * a12) / d;
        float t0 = -1.f, t1 = -1.f, t2 = -1.f;
        if (cnt >= 4 && fabs(piv) > 1e-12 * fabs(a22)) {
            const double z = (b2 - (a02 * b0 + a12 * b1) / d) / piv;
            t0 = (float)((b0 - a02 * z) / d); t1 = (float)((b1 - a12 * z) / d); t2 = (float)z;
        }
        trans[n * 3] = t0; trans[n * 3 + 1] = t1; trans[n * 3 + 2] = t2;
    }
}

// batch_orth_proj(mode='3d', keep_dim=True) + convert_proejection_from_input_to_orgimg on (N,V,3) vertices
// (post_parser.py:81-88,108,113; utils.py:309-315): x,y projected, z kept; then all three to original-image pixels.
#pragma clang fp contract(off)   // mul, then add, like the reference's two tensor ops: the renderer's z-test sees these bits
__global__ void project_verts_kernel(const float* __restrict__ verts, size_t total, int V, const float* __restrict__ cam,
                                     float pad_size, float left, float top, float* __restrict__ camed, float* __restrict__ org) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t n = i / V;
    const float s = cam[n * 3], tx = cam[n * 3 + 1], ty = cam[n * 3 + 2];
    const float px = verts[i * 3] * s + tx, py = verts[i * 3 + 1] * s + ty, pz = verts[i * 3 + 2];
    if (camed) { camed[i * 3] = px; camed[i * 3 + 1] = py; camed[i * 3 + 2] = pz; }
    org[i * 3] = (px + 1.f) * pad_size / 2.f - left;
    org[i * 3 + 1] = (py + 1.f) * pad_size / 2.f - top;
    org[i * 3 + 2] = (pz + 1.f) * pad_size / 2.f;
}

// bev/post_parser.py:68-107,144-151: ((v + t).xy / ((v + t).z + 1e-6)) * 443.4 / 256, z = v.z, then input -> original image
__global__ void bev_project_verts_kernel(const float* __restrict__ verts, size_t total, int V, const float* __restrict__ trans,
                                         float pad_size, float left, float top, float* __restrict__ org) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t n = i / V;
    const float vz = verts[i * 3 + 2];
    const float x = verts[i * 3] + trans[n * 3], y = verts[i * 3 + 1] + trans[n * 3 + 1], z = vz + trans[n * 3 + 2] + 1e-6f;
    const float px = x / z * 443.4f / 256.f, py = y / z * 443.4f / 256.f;
    org[i * 3] = (px + 1.f) * pad_size / 2.f - left;
    org[i * 3 + 1] = (py + 1.f) * pad_size / 2.f - top;
    org[i * 3 + 2] = (vz + 1.f) * pad_size / 2.f;
}

}  // namespace romp

using namespace romp;

extern "C" {

int romp_bev_project_verts(const float* verts, int N, int V, const float* cam_trans, const float* pad_info_host, float* verts_camed_org,
                           void* stream) {
    ROMP_REQUIRE(verts && cam_trans && pad_info_host && verts_camed_org && N > 0 && V > 0, "romp_bev_project_verts: bad arguments");
    const float top = pad_info_host[0], left = pad_info_host[2], h = pad_info_host[4], w = pad_info_host[5];
    const size_t total = (size_t)N * V;
    hipLaunchKernelGGL(bev_project_verts_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, verts, total,
                       V, cam_trans, h > w ? h : w, left, top, verts_camed_org);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

int romp_project_verts(const float* verts, int N, int V, const float* cam, const float* pad_info_host, float* verts_camed,
                       float* verts_camed_org, void* stream) {
    ROMP_REQUIRE(verts && cam && pad_info_host && verts_camed_org && N > 0 && V > 0, "romp_project_verts: bad arguments");
    const float top = pad_info_host[0], left = pad_info_host[2], h = pad_info_host[4], w = pad_info_host[5];
    const size_t total = (size_t)N * V;
    hipLaunchKernelGGL(project_verts_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, verts, total, V,
                       cam, h > w ? h : w, left, top, verts_camed, verts_camed_org);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

int romp_parse_watch(const float* center_maps, const float* params_maps, int B, float conf_thresh, int max_person,
                     int32_t* count_host, int32_t* batch_ids, int32_t* flat_inds, float* scores, float* params_pred,
                     float* cam, float* thetas, float* betas, int32_t* center_preds, int32_t* workspace, void* stream,
                     const int32_t* watch, int32_t* watch_host) {
    ROMP_REQUIRE(center_maps && params_maps && workspace && B > 0, "romp_parse: bad arguments");
    ROMP_REQUIRE(max_person >= 1 && max_person <= 1024, "romp_parse: max_person %d out of range", max_person);
    ROMP_REQUIRE(batch_ids && flat_inds && scores && params_pred && cam && thetas && betas && center_preds,
                 "romp_parse: null output");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(parse_nms_topk_kernel, dim3(B), dim3(PARSE_THREADS), 0, st, center_maps, conf_thresh, max_person, workspace);
    ROMP_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(parse_pack_kernel, dim3(max_person, B), dim3(64), 0, st, center_maps, params_maps, B, max_person,
                       workspace, batch_ids, flat_inds, scores, params_pred, cam, thetas, betas, center_preds, watch);
    ROMP_HIP_CHECK(hipGetLastError());
    if (!count_host) return ROMP_OK;     // asynchronous form: image b's count stays in workspace[b * (2 * max_person + 2) + 2 * max_person]
    const size_t n_cnt = (size_t)B * (2 * max_person + 2);
    std::vector<int32_t> cnt(n_cnt + 1);                                     // (+ the watched word, when there is one: the same copy)
    ROMP_HIP_CHECK(hipMemcpyAsync(cnt.data(), workspace, (n_cnt + (watch ? 1 : 0)) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ROMP_HIP_CHECK(hipStreamSynchronize(st));
    int total = 0;
    for (int b = 0; b < B; ++b) total += cnt[(size_t)b * (2 * max_person + 2) + 2 * max_person];
    *count_host = total;
    if (watch && watch_host) *watch_host = cnt[n_cnt];
    return ROMP_OK;
}

int romp_parse(const float* center_maps, const float* params_maps, int B, float conf_thresh, int max_person,
               int32_t* count_host, int32_t* batch_ids, int32_t* flat_inds, float* scores, float* params_pred,
               float* cam, float* thetas, float* betas, int32_t* center_preds, int32_t* workspace, void* stream) {
    return romp_parse_watch(center_maps, params_maps, B, conf_thresh, max_person, count_host, batch_ids, flat_inds, scores, params_pred,
                            cam, thetas, betas, center_preds, workspace, stream, nullptr, nullptr);
}

int romp_rot6d_to_aa(const float* x6, int n, float* aa, void* stream) {
    ROMP_REQUIRE(x6 && aa && n >= 0, "romp_rot6d_to_aa: bad arguments");
    if (n == 0) return ROMP_OK;
    hipLaunchKernelGGL(rot6d_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, x6, n, aa);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

int romp_cam_to_trans(const float* cam, int N, float weight, float* trans, void* stream) {
    ROMP_REQUIRE(cam && trans && N >= 0, "romp_cam_to_trans: bad arguments");
    if (N == 0) return ROMP_OK;
    hipLaunchKernelGGL(cam_to_trans_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, cam, N, weight, trans);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

int romp_estimate_translation(const float* joints, int N, int J, int K, const float* pj2d, float focal_length, float img_size,
                              float* trans, void* stream) {
    ROMP_REQUIRE(joints && pj2d && trans && N >= 0 && J > 0 && K >= 2 && K <= J, "romp_estimate_translation: bad arguments");
    if (N == 0) return ROMP_OK;
    hipLaunchKernelGGL(translation_lsq_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, joints, J, K, pj2d, (double)focal_length,
                       (double)img_size / 2.0, (double)img_size / 2.0, trans);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

int romp_project(const float* joints, int N, int J, const float* cam, const float* pad_info_host, float* pj2d,
                 float* pj2d_org, float* cam_trans, void* stream) {
    ROMP_REQUIRE(joints && cam && pad_info_host && pj2d && pj2d_org && cam_trans && N >= 0 && J > 0,
                 "romp_project: bad arguments");
    if (N == 0) return ROMP_OK;
    const float top = pad_info_host[0], left = pad_info_host[2], h = pad_info_host[4], w = pad_info_host[5];
    const float pad_size = h > w ? h : w;
    const int total = N * J;
    hipLaunchKernelGGL(project_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, joints, N, J, cam,
                       pad_size, left, top, pj2d, pj2d_org, cam_trans);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

}  // extern "C"
