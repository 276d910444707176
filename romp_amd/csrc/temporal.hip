// temporal.hip -- OneEuro smoothing of the per-person estimates between video frames.
//
// Reference: simple_romp/romp/utils.py  LowPassFilter :203-215, OneEuroFilter :217-245, create_OneEuroFilter
// :257-258 (thetas: mincutoff = smooth_coeff, cam 1.6, betas 0.6, global_rot smooth_coeff; beta 0.7, dcutoff 1,
// 30 Hz), smooth_results :261-269, smooth_global_rot_matrix :188-192 (the root orientation is filtered as a
// 3x3 matrix: batch_rodrigues :493-533 -> filter -> rotation_matrix_to_angle_axis :535-552).
//
// One workgroup per tracked person, one lane per filtered scalar (9 + 69 + n_betas + 3).  The filter state of
// a track lives in a caller-owned device buffer: [initialised, x_raw[D], x_filtered[D], dx_filtered[D]].
#include "common.h"
#include "rot6d.h"

namespace romp {

struct OneEuroCfg {
    float freq, te;                 // 30 Hz, 1/freq
    float alpha_d, one_m_alpha_d;   // compute_alpha(dcutoff) and 1 - it (formed in double like the reference's scalars)
    float two_pi;
    float mincut[4], beta[4];       // per segment: global rot, body pose, betas, cam
};

__device__ __forceinline__ float rodrigues_elem(const float* aa, int e) {      // utils.py:493-533, entry e of the 3x3
    const float ax = aa[0] + 1e-8f, ay = aa[1] + 1e-8f, az = aa[2] + 1e-8f;
    const float angle = sqrtf(ax * ax + ay * ay + az * az);
    const float nx = aa[0] / angle, ny = aa[1] / angle, nz = aa[2] / angle;
    const float half = angle * 0.5f, c = cosf(half), s = sinf(half);
    float w = c, x = s * nx, y = s * ny, z = s * nz;
    const float qn = sqrtf(w * w + x * x + y * y + z * z);
    w /= qn; x /= qn; y /= qn; z /= qn;
    const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z, wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    switch (e) {
        case 0: return w2 + x2 - y2 - z2;
        case 1: return 2 * xy - 2 * wz;
        case 2: return 2 * wy + 2 * xz;
        case 3: return 2 * wz + 2 * xy;
        case 4: return w2 - x2 + y2 - z2;
        case 5: return 2 * yz - 2 * wx;
        case 6: return 2 * xz - 2 * wy;
        case 7: return 2 * wx + 2 * yz;
        default: return w2 - x2 - y2 + z2;
    }
}

__global__ __launch_bounds__(128) void oneeuro_kernel(float* __restrict__ state, const int32_t* __restrict__ slots, int n_betas,
                                                       OneEuroCfg cfg, float* __restrict__ thetas, float* __restrict__ betas,
                                                       float* __restrict__ cam) {
    __shared__ float s_rot[9];
    const int n = blockIdx.x, e = threadIdx.x;
    const int D = 9 + 69 + n_betas + 3;
    float* st = state + (size_t)slots[n] * (1 + 3 * D);
    const bool fresh = st[0] == 0.f;
    if (e < D) {
        int seg;
        float x;
        if (e < 9) { seg = 0; x = rodrigues_elem(thetas + (size_t)n * 72, e); }
        else if (e < 78) { seg = 1; x = thetas[(size_t)n * 72 + 3 + (e - 9)]; }
        else if (e < 78 + n_betas) { seg = 2; x = betas[(size_t)n * n_betas + (e - 78)]; }
        else { seg = 3; x = cam[(size_t)n * 3 + (e - 78 - n_betas)]; }
        float* x_raw = st + 1, *x_f = st + 1 + D, *dx_f = st + 1 + 2 * D;
        float out, edx;
        if (fresh) {                                   // LowPassFilter.process with prev_raw_value None (:209-210)
            out = x; edx = 0.f;
        } else {
            const float dx = (x - x_raw[e]) * cfg.freq;
            edx = cfg.alpha_d * dx + cfg.one_m_alpha_d * dx_f[e];
            const float cutoff = cfg.mincut[seg] + cfg.beta[seg] * fabsf(edx);
            const float tau = 1.0f / (cfg.two_pi * cutoff);
            const float alpha = 1.0f / (1.0f + tau / cfg.te);
            out = alpha * x + (1.0f - alpha) * x_f[e];
        }
        x_raw[e] = x; x_f[e] = out; dx_f[e] = edx;
        if (e < 9) s_rot[e] = out;
        else if (e < 78) thetas[(size_t)n * 72 + 3 + (e - 9)] = out;
        else if (e < 78 + n_betas) betas[(size_t)n * n_betas + (e - 78)] = out;
        else cam[(size_t)n * 3 + (e - 78 - n_betas)] = out;
    }
    __syncthreads();
    if (e == 0) {
        st[0] = 1.f;
        // rotation_matrix_to_angle_axis on the filtered matrix R (row-major s_rot): m(i,j) = R[j][i]
        rmat_t_to_aa_dev(s_rot[0], s_rot[3], s_rot[6], s_rot[1], s_rot[4], s_rot[7], s_rot[2], s_rot[5], s_rot[8],
                         thetas + (size_t)n * 72);
    }
}

}  // namespace romp

using namespace romp;

extern "C" {

int romp_oneeuro_state_floats(int n_betas) { return 1 + 3 * (9 + 69 + n_betas + 3); }

int romp_oneeuro_smooth(float* state, const int32_t* slots, int N, int n_betas, float smooth_coeff, float* thetas, float* betas,
                        float* cam, void* stream) {
    ROMP_REQUIRE(state && slots && thetas && betas && cam && N > 0 && n_betas >= 1 && n_betas <= 16 && smooth_coeff > 0.f,
                 "romp_oneeuro_smooth: bad arguments");
    OneEuroCfg c;
    const double freq = 30.0, te = 1.0 / freq, tau_d = 1.0 / (2 * 3.141592653589793 * 1.0), alpha_d = 1.0 / (1.0 + tau_d / te);
    c.freq = (float)freq; c.te = (float)te; c.alpha_d = (float)alpha_d; c.one_m_alpha_d = (float)(1.0 - alpha_d);
    c.two_pi = (float)(2 * 3.141592653589793);
    const float mc[4] = {smooth_coeff, smooth_coeff, 0.6f, 1.6f};
    for (int k = 0; k < 4; ++k) { c.mincut[k] = mc[k]; c.beta[k] = 0.7f; }
    hipLaunchKernelGGL(oneeuro_kernel, dim3(N), dim3(128), 0, (hipStream_t)stream, state, slots, n_betas, c, thetas, betas, cam);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

}  // extern "C"
