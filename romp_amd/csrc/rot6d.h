// rot6d.h -- 6D rotation -> axis-angle on device, branch-for-branch the reference's chain
// rot6d_to_rotmat -> rotation_matrix_to_quaternion -> quaternion_to_angle_axis (+ NaN -> 0)
// (simple_romp/romp/utils.py:477-491, 606-682, 554-604, 551).  Shared by parse.hip, bev.hip and
// temporal.hip (rotation_matrix_to_angle_axis alone, utils.py:535-552).
#pragma once
#include <hip/hip_runtime.h>

namespace romp {

// rmat_t = R^T of the reference (utils.py:636): m(i,j) = R[j][i].
__device__ __forceinline__ void rmat_t_to_aa_dev(float m00, float m01, float m02, float m10, float m11, float m12,
                                                 float m20, float m21, float m22, float* aa) {
#pragma clang fp contract(off)   // keep the reference's rounding in the ill-conditioned branches
    const bool d2 = m22 < 1e-6f, d01 = m00 > m11, d0n1 = m00 < -m11;
    float q0, q1, q2, q3, t;
    if (d2 && d01) {
        t = 1 + m00 - m11 - m22;
        q0 = m12 - m21; q1 = t; q2 = m01 + m10; q3 = m20 + m02;
    } else if (d2 && !d01) {
        t = 1 - m00 + m11 - m22;
        q0 = m20 - m02; q1 = m01 + m10; q2 = t; q3 = m12 + m21;
    } else if (!d2 && d0n1) {
        t = 1 - m00 - m11 + m22;
        q0 = m01 - m10; q1 = m20 + m02; q2 = m12 + m21; q3 = t;
    } else {
        t = 1 + m00 + m11 + m22;
        q0 = t; q1 = m12 - m21; q2 = m20 - m02; q3 = m01 - m10;
    }
    const float st = sqrtf(t);
    q0 = q0 / st * 0.5f; q1 = q1 / st * 0.5f; q2 = q2 / st * 0.5f; q3 = q3 / st * 0.5f;
    // quaternion_to_angle_axis (utils.py:554-604)
    const float s2 = q1 * q1 + q2 * q2 + q3 * q3;
    const float s = sqrtf(s2);
    const float two_theta = 2.0f * (q0 < 0.0f ? atan2f(-s, -q0) : atan2f(s, q0));
    const float k = s2 > 0.0f ? two_theta / s : 2.0f;
    float rx = q1 * k, ry = q2 * k, rz = q3 * k;
    aa[0] = (rx != rx) ? 0.f : rx;                     // aa[isnan(aa)] = 0 (utils.py:551)
    aa[1] = (ry != ry) ? 0.f : ry;
    aa[2] = (rz != rz) ? 0.f : rz;
}

__device__ __forceinline__ void rot6d_to_aa_dev(const float* x, float* aa) {
#pragma clang fp contract(off)   // keep the reference's rounding in the ill-conditioned branches
    // rot6d_to_rotmat (utils.py:477-491): x.view(3,2): a1 = x[0],x[2],x[4]; a2 = x[1],x[3],x[5]
    const float a1x = x[0], a1y = x[2], a1z = x[4], a2x = x[1], a2y = x[3], a2z = x[5];
    float n1 = sqrtf(a1x * a1x + a1y * a1y + a1z * a1z);
    n1 = fmaxf(n1, 1e-6f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float dot = b1x * a2x + b1y * a2y + b1z * a2z;
    const float ux = a2x - dot * b1x, uy = a2y - dot * b1y, uz = a2z - dot * b1z;
    float n2 = sqrtf(ux * ux + uy * uy + uz * uz);
    n2 = fmaxf(n2, 1e-6f);
    const float b2x = ux / n2, b2y = uy / n2, b2z = uz / n2;
    const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
    // R[r][c]: columns b1,b2,b3.  rmat_t = R^T, m(i,j) = R[j][i]  (utils.py:636)
    rmat_t_to_aa_dev(b1x, b1y, b1z, b2x, b2y, b2z, b3x, b3y, b3z, aa);   // m(0,j) = R[j][0] = b1[j], ...
}

}  // namespace romp
