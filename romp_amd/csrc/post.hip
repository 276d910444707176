// post.hip -- the callers either side of the network (SURVEY.md §8f "next" rows 1 and 2):
//
//  * preprocess_kernel: img_preprocess (simple_romp/romp/utils.py:16-30) on device -- BGR->RGB,
//    centred zero pad to a square, cv::resize(INTER_CUBIC) to 512x512 in OpenCV's own arithmetic (pixel
//    centres, a = -0.75, 11-bit fixed-point coefficients, int32 horizontal pass, (v + 2^21) >> 22,
//    saturate, replicated border) -> float32 (B,512,512,3).  The caller uploads uint8 frames (0.25 B/px/ch)
//    instead of the float tensor (4x the bytes).  Bit-exact against the test suite's CPU
//    restatement of OpenCV's published scalar algorithm (cv2 itself is not installed here; its SIMD
//    builds round the vertical pass in float32, which can differ by one grey level on rare pixels).
//
//  * bev_post_kernel: BEV's per-image post-processing (simple_romp/bev/post_parser.py):
//    denormalize_cam_params_to_trans :114-128, perspective_projection :68-107 (+ to-original-image
//    :129-136), suppressing_redundant_prediction_via_projection :167-198, remove_outlier :200-222.
//    One workgroup per image; persons of an image are contiguous rows.  Output: projections and a
//    keep mask (the caller drops the rows).
#include "common.h"

namespace romp {

// cv::resize(INTER_CUBIC) tables for one destination coordinate, exactly as OpenCV builds them (resize.cpp: the sampling
// position in double, the cubic weights in float with A = -0.75, then 11-bit fixed point with round-half-even).  `fp contract(off)`
// keeps hipcc from fusing a*b+c into an FMA, which OpenCV's scalar code does not do.
__device__ __forceinline__ void cv_cubic_tab(int d, int src, int dst, int& s0, int (&coef)[4]) {
#pragma clang fp contract(off)
    const double scale = 1.0 / ((double)dst / (double)src);
    const double fd = ((double)d + 0.5) * scale - 0.5;
    const float f = (float)fd;
    const int fl = (int)floorf(f);
    const float x = f - (float)fl;
    const float A = -0.75f;
    const float xp = x + 1.f, xm = 1.f - x;
    float c[4];
    c[0] = ((A * xp - 5.f * A) * xp + 8.f * A) * xp - 4.f * A;
    c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    c[2] = ((A + 2.f) * xm - (A + 3.f)) * xm * xm + 1.f;
    c[3] = 1.f - c[0] - c[1] - c[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) coef[k] = min(max((int)rintf(c[k] * 2048.f), -32768), 32767);
    s0 = fl - 1;
}

// One thread per output pixel of one frame (blockIdx.y): the 4x4 taps of the PADDED square image (zero padding around the
// frame, replicated border of the padded image), horizontal pass to int32, vertical pass, (v + 2^21) >> 22, saturate.
__global__ void preprocess_kernel(const unsigned char* __restrict__ src_all, int H, int W, int side, int top, int left,
                                  float* __restrict__ dst_all, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * S) return;
    const unsigned char* src = src_all + (size_t)blockIdx.y * H * W * 3;
    float* dst = dst_all + (size_t)blockIdx.y * S * S * 3;
    const int ox = i % S, oy = i / S;
    int sx, sy, ca[4], cb[4];
    cv_cubic_tab(ox, side, S, sx, ca);
    cv_cubic_tab(oy, side, S, sy, cb);
    int acc[3] = {0, 0, 0};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int py = min(max(sy + a, 0), side - 1) - top;             // replicated border of the PADDED image
        int row[3] = {0, 0, 0};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int px = min(max(sx + b, 0), side - 1) - left;
            // branch-free (taps in the zero padding read pixel 0 with weight 0): all 48 byte loads of a thread in flight
            const bool ok = (unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W;
            const unsigned char* p = src + (ok ? ((size_t)py * W + px) * 3 : 0);
            const int wgt = ok ? ca[b] : 0;
            row[0] += wgt * (int)p[2]; row[1] += wgt * (int)p[1]; row[2] += wgt * (int)p[0];       // BGR -> RGB
        }
        acc[0] += row[0] * cb[a]; acc[1] += row[1] * cb[a]; acc[2] += row[2] * cb[a];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[(size_t)i * 3 + c] = (float)min(max((acc[c] + (1 << 21)) >> 22, 0), 255);
}

constexpr int PJ = 71, PMAX = 64;

__global__ __launch_bounds__(256) void bev_post_kernel(const float* __restrict__ joints, const float* __restrict__ cam,
                                                        const int* __restrict__ offsets, const float* __restrict__ pad_info,
                                                        float nms_thresh, float rel_thresh, float scale_thresh,
                                                        float* pj2d, float* pj2d_org, float* cam_trans, int* keep) {
    __shared__ float s_pj[PMAX * PJ * 2];
    __shared__ float s_scale[PMAX], s_tr[PMAX][3], s_mean[PMAX];
    __shared__ int s_removed[PMAX], s_alive[PMAX], s_n_alive;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int off = offsets[b], n = min(offsets[b + 1] - off, PMAX);
    if (n <= 0) return;
    const float* pi = pad_info + b * 6;                       // top, bottom, left, right, h, w
    const float top = pi[0], left = pi[2], pad_size = fmaxf(pi[4], pi[5]);
    const float tan_fov = 0.57735026918962573f;
    if (tid < n) {                                            // denormalize_cam_params_to_trans (:114-128)
        const float* c = cam + (size_t)(off + tid) * 3;
        const float depth = 1.f / (c[0] * tan_fov + 1e-3f);
        s_tr[tid][0] = c[2] * depth * tan_fov; s_tr[tid][1] = c[1] * depth * tan_fov; s_tr[tid][2] = depth;
        for (int k = 0; k < 3; ++k) cam_trans[(size_t)(off + tid) * 3 + k] = s_tr[tid][k];
        s_scale[tid] = c[0] * 2.f;
        s_removed[tid] = 0;
    }
    __syncthreads();
    for (int i = tid; i < n * PJ; i += 256) {                  // perspective_projection (:68-107), then to org image
        const int p = i / PJ;
        const float* j = joints + (size_t)(off * PJ + i) * 3;
        const float z = (j[2] + s_tr[p][2]) + 1e-6f;
        float x = (j[0] + s_tr[p][0]) / z * 443.4f, y = (j[1] + s_tr[p][1]) / z * 443.4f;
        x /= 256.f; y /= 256.f;                                // normalize: /= img_size/2
        pj2d[(size_t)(off * PJ + i) * 2] = x; pj2d[(size_t)(off * PJ + i) * 2 + 1] = y;
        x = (x + 1.f) * pad_size / 2.f - left; y = (y + 1.f) * pad_size / 2.f - top;
        s_pj[i * 2] = x; s_pj[i * 2 + 1] = y;
        pj2d_org[(size_t)(off * PJ + i) * 2] = x; pj2d_org[(size_t)(off * PJ + i) * 2 + 1] = y;
    }
    __syncthreads();
    // suppressing_redundant_prediction_via_projection (:167-198): pairs a<b closer than the threshold
    if (n > 1) {
        const float thr = nms_thresh * pad_size / 640.f;       // max(img_shape) == max(h,w) == pad size
        for (int pr = tid; pr < n * n; pr += 256) {
            const int a = pr / n, c = pr % n;
            if (a >= c) continue;
            float d = 0.f;
            for (int k = 0; k < PJ; ++k) {
                const float dx = s_pj[(a * PJ + k) * 2] - s_pj[(c * PJ + k) * 2];
                const float dy = s_pj[(a * PJ + k) * 2 + 1] - s_pj[(c * PJ + k) * 2 + 1];
                d += sqrtf(dx * dx + dy * dy);
            }
            d = d / PJ / fmaxf(s_scale[a], s_scale[c]);
            if (d < thr) atomicExch(&s_removed[s_scale[a] < s_scale[c] ? a : c], 1);   // drop the smaller (farther) one
        }
    }
    __syncthreads();
    if (tid == 0) {
        int m = 0;
        for (int p = 0; p < n; ++p)
            if (!s_removed[p]) s_alive[m++] = p;
        s_n_alive = m;
    }
    __syncthreads();
    // remove_outlier (:200-222) on the survivors: mean distance to the others without the smallest and
    // the largest entry of the row (sorted()[1:-1])
    const int m = s_n_alive;
    if (m >= 3) {
        if (tid < m) {
            const int a = s_alive[tid];
            float sum = 0.f, mn = 3.4e38f, mx = -1.f;
            for (int k = 0; k < m; ++k) {
                const int c = s_alive[k];
                const float dx = s_tr[a][0] - s_tr[c][0], dy = s_tr[a][1] - s_tr[c][1], dz = s_tr[a][2] - s_tr[c][2];
                const float d = sqrtf(dx * dx + dy * dy + dz * dz);
                sum += d; mn = fminf(mn, d); mx = fmaxf(mx, d);
            }
            s_mean[tid] = (sum - mn - mx) / (m - 2);
        }
        __syncthreads();
        if (tid < m) {
            float tot = 0.f;
            for (int k = 0; k < m; ++k) tot += s_mean[k];
            const float rel = s_mean[tid] / ((tot - s_mean[tid]) / (m - 1));
            const int a = s_alive[tid];
            if (rel > rel_thresh && cam[(size_t)(off + a) * 3] < scale_thresh) s_removed[a] = 1;
        }
        __syncthreads();
    }
    if (tid < n) keep[off + tid] = !s_removed[tid];
}

}  // namespace romp

using namespace romp;

extern "C" {

int romp_preprocess(const unsigned char* bgr_u8, int H, int W, float* out_rgb_f32, int out_size, float* pad_info_host,
                    void* stream) {
    ROMP_REQUIRE(bgr_u8 && out_rgb_f32 && H > 0 && W > 0 && out_size > 0, "romp_preprocess: bad arguments");
    const int side = H > W ? H : W;
    const int top = (side - H) / 2, left = (side - W) / 2;
    if (pad_info_host) {
        pad_info_host[0] = (float)top; pad_info_host[1] = (float)(top + H); pad_info_host[2] = (float)left;
        pad_info_host[3] = (float)(left + W); pad_info_host[4] = (float)H; pad_info_host[5] = (float)W;
    }
    const int total = out_size * out_size;
    hipLaunchKernelGGL(preprocess_kernel, dim3((total + 255) / 256, 1), dim3(256), 0, (hipStream_t)stream, bgr_u8, H, W, side, top,
                       left, out_rgb_f32, out_size);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

int romp_preprocess_batch(const unsigned char* bgr_u8, int B, int H, int W, float* out_rgb_f32, int out_size, float* pad_info_host,
                          void* stream) {
    ROMP_REQUIRE(bgr_u8 && out_rgb_f32 && B > 0 && B < 65536 && H > 0 && W > 0 && out_size > 0, "romp_preprocess_batch: bad arguments");
    const int side = H > W ? H : W;
    const int top = (side - H) / 2, left = (side - W) / 2;
    if (pad_info_host) {
        pad_info_host[0] = (float)top; pad_info_host[1] = (float)(top + H); pad_info_host[2] = (float)left;
        pad_info_host[3] = (float)(left + W); pad_info_host[4] = (float)H; pad_info_host[5] = (float)W;
    }
    const int total = out_size * out_size;
    hipLaunchKernelGGL(preprocess_kernel, dim3((total + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, bgr_u8, H, W, side, top,
                       left, out_rgb_f32, out_size);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

int romp_bev_postprocess(const float* joints, const float* cam, const int32_t* offsets, int B, const float* pad_info,
                         float nms_thresh, float relative_scale_thresh, float* pj2d, float* pj2d_org, float* cam_trans,
                         int32_t* keep, void* stream) {
    ROMP_REQUIRE(joints && cam && offsets && pad_info && pj2d && pj2d_org && cam_trans && keep && B > 0,
                 "romp_bev_postprocess: bad arguments");
    hipLaunchKernelGGL(bev_post_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, joints, cam, offsets, pad_info, nms_thresh,
                       relative_scale_thresh, 0.25f, pj2d, pj2d_org, cam_trans, keep);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

}  // extern "C"
