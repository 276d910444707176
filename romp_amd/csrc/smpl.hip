// smpl.hip -- SMPL forward (seam #3): shape + pose blend shapes, Rodrigues, kinematic chain,
// linear-blend skinning of 6890 vertices x 24 joints, 71-joint regression, optional root alignment.
//
// Reference: SMPL.forward smpl.py:62-108, lbs :111-188, batch_rodrigues :191-222,
// batch_rigid_transform :236-290, VertexJointSelector.forward :24-35.
//
// The reference materialises v_shaped, pose_offsets, v_posed, the expanded weights and a per-vertex
// 4x4 transform tensor T (N,6890,4,4 = 28 MB at N=64) -- ~10x the algorithmic traffic.  Here:
//   kernel A (one wave per person)  Rodrigues x24, rest joints from the PRE-REGRESSED template
//            J = J_template + J_shapedirs.beta  (algebraically J_regressor @ (v_template + S.beta),
//            smpl.py:153-156, without the 6890-long reduction), 24-step parent chain, A matrices;
//   kernel B (one lane per vertex, PB persons per workgroup)  shape blend + 207-term pose blend +
//            24-joint skinning entirely in registers; posedirs (17 MB) is the only large stream and
//            each element is used for PB persons; nothing but the final vertex is written;
//   kernel C (one workgroup per person)  21 vertex picks + 26 regressed joints, wavefront reductions.
// All float32; bound = HBM (constants 20 MB once + 82.7 KB written per person).
#include "common.h"
#include <vector>

namespace romp {

constexpr int NV = 6890, NJ = 24, NPF = 207, NJOUT = 71, NREG = 26, NPICK = 21;
constexpr int PB = 8;    // persons per workgroup in the skinning kernel

struct Parents { int p[NJ]; };

// J_template[j][k] = sum_v Jreg[j][v] v_template[v][k];  J_shapedirs[j][k][l] = sum_v Jreg[j][v] S[v][k][l]
__global__ __launch_bounds__(256) void smpl_prep_kernel(const float* __restrict__ Jreg, const float* __restrict__ vt,
                                                         const float* __restrict__ sd, int nb, float* __restrict__ Jt,
                                                         float* __restrict__ Js) {
    const int j = blockIdx.x, e = blockIdx.y;           // e in [0, 3*(nb+1)): k = e/(nb+1), l = e%(nb+1) (l==nb -> template)
    const int k = e / (nb + 1), l = e % (nb + 1);
    double acc = 0.0;
    for (int v = threadIdx.x; v < NV; v += 256) {
        const float w = Jreg[(size_t)j * NV + v];
        const float x = (l == nb) ? vt[v * 3 + k] : sd[((size_t)v * 3 + k) * nb + l];
        acc += (double)w * (double)x;
    }
    __shared__ double s[256];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (l == nb) Jt[j * 3 + k] = (float)s[0];
        else Js[(j * 3 + k) * nb + l] = (float)s[0];
    }
}

// ---- kernel A: per-person pose prologue ---------------------------------------------------------
__global__ __launch_bounds__(64) void smpl_pose_kernel(const float* __restrict__ betas, int nb,
                                                        const float* __restrict__ thetas,
                                                        const float* __restrict__ Jt, const float* __restrict__ Js,
                                                        Parents par, float* __restrict__ pose_feat,
                                                        float* __restrict__ Amat, float* __restrict__ joints) {
    __shared__ float sR[NJ][9], sJ[NJ][3], sG[NJ][12];
    const int n = blockIdx.x, lane = threadIdx.x;
    if (lane < NJ) {
        const float* r = thetas + (size_t)n * 72 + lane * 3;
        const float rx0 = r[0], ry0 = r[1], rz0 = r[2];
        const float ex = rx0 + 1e-8f, ey = ry0 + 1e-8f, ez = rz0 + 1e-8f;     // smpl.py:206
        const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
        const float rx = rx0 / angle, ry = ry0 / angle, rz = rz0 / angle;
        float sn, cs;
        sincosf(angle, &sn, &cs);
        const float oc = 1.f - cs;
        // K = [[0,-rz,ry],[rz,0,-rx],[-ry,rx,0]];  R = I + sin K + (1-cos) K K   (smpl.py:217-221)
        const float kk00 = -(rz * rz) - ry * ry, kk11 = -(rz * rz) - rx * rx, kk22 = -(ry * ry) - rx * rx;
        const float kk01 = ry * rx, kk02 = rz * rx, kk12 = rz * ry;
        float R[9];
        R[0] = 1.f + oc * kk00;      R[1] = sn * -rz + oc * kk01; R[2] = sn * ry + oc * kk02;
        R[3] = sn * rz + oc * kk01;  R[4] = 1.f + oc * kk11;      R[5] = sn * -rx + oc * kk12;
        R[6] = sn * -ry + oc * kk02; R[7] = sn * rx + oc * kk12;  R[8] = 1.f + oc * kk22;
#pragma unroll
        for (int e = 0; e < 9; ++e) sR[lane][e] = R[e];
        if (lane >= 1) {
#pragma unroll
            for (int e = 0; e < 9; ++e)
                pose_feat[(size_t)n * NPF + (lane - 1) * 9 + e] = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float a = Jt[lane * 3 + k];
            for (int l = 0; l < nb; ++l) a = fmaf(betas[(size_t)n * nb + l], Js[(lane * 3 + k) * nb + l], a);
            sJ[lane][k] = a;
        }
    }
    __syncthreads();
    // kinematic chain (smpl.py:269-275): G[i] = G[parent] * [R_i | J_i - J_parent]
    const int rr = lane / 4, cc = lane % 4;            // lanes 0..11 own one element of the 3x4 result
    if (lane < 12) sG[0][lane] = (cc < 3) ? sR[0][rr * 3 + cc] : sJ[0][rr];
    __syncthreads();
    for (int i = 1; i < NJ; ++i) {
        const int p = par.p[i];
        if (lane < 12) {
            float v;
            if (cc < 3) {
                v = sG[p][rr * 4 + 0] * sR[i][0 * 3 + cc];
                v = fmaf(sG[p][rr * 4 + 1], sR[i][1 * 3 + cc], v);
                v = fmaf(sG[p][rr * 4 + 2], sR[i][2 * 3 + cc], v);
            } else {
                v = sG[p][rr * 4 + 0] * (sJ[i][0] - sJ[p][0]);
                v = fmaf(sG[p][rr * 4 + 1], sJ[i][1] - sJ[p][1], v);
                v = fmaf(sG[p][rr * 4 + 2], sJ[i][2] - sJ[p][2], v);
                v += sG[p][rr * 4 + 3];
            }
            sG[i][lane] = v;
        }
        __syncthreads();
    }
    // posed joints + relative transforms A = G - pad(G [J;0])   (smpl.py:280-288)
    for (int idx = lane; idx < NJ * 12; idx += 64) {
        const int j = idx / 12, e = idx % 12, r = e / 4, c = e % 4;
        float v = sG[j][e];
        if (c == 3) {
            joints[((size_t)n * NJOUT + j) * 3 + r] = v;
            const float t = sG[j][r * 4 + 0] * sJ[j][0] + sG[j][r * 4 + 1] * sJ[j][1] + sG[j][r * 4 + 2] * sJ[j][2];
            v -= t;
        }
        Amat[(size_t)n * NJ * 12 + idx] = v;
    }
}

// ---- kernel B: per-vertex blend shapes + skinning -----------------------------------------------
// Workgroup = 4 waves on the SAME 64 vertices and PB persons: each wave accumulates a quarter of the 207
// pose-blend terms (the only long dependent loop: strided posedirs loads), the partial sums meet in LDS, then
// wave w finishes persons 2w, 2w+1 (shape blend, skinning).  4x the waves in flight of the one-wave version
// and a 4x shorter load chain: 92 -> 33 us at N = 64.
constexpr int SKIN_WAVES = 4, KQ = (NPF + SKIN_WAVES - 1) / SKIN_WAVES;   // 52 terms per wave
template <int NB>
__global__ __launch_bounds__(256) void smpl_skin_kernel(const float* __restrict__ betas, const float* __restrict__ pose_feat,
                                                         const float* __restrict__ Amat, const float* __restrict__ vt,
                                                         const float* __restrict__ sd, const float* __restrict__ pd,
                                                         const float* __restrict__ lbsw, int N, float* __restrict__ verts) {
    __shared__ __attribute__((aligned(16))) float s_pf[NPF][PB];
    __shared__ __attribute__((aligned(16))) float s_A[PB][NJ][12];
    __shared__ float s_beta[PB][NB];
    __shared__ float s_po[SKIN_WAVES][PB * 3][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p0 = blockIdx.y * PB;
    const int np = min(PB, N - p0);
    for (int idx = tid; idx < NPF * PB; idx += 256) {
        const int k = idx / PB, p = idx % PB;
        s_pf[k][p] = p < np ? pose_feat[(size_t)(p0 + p) * NPF + k] : 0.f;
    }
    for (int idx = tid; idx < PB * NJ * 12; idx += 256) {
        const int p = idx / (NJ * 12);
        (&s_A[0][0][0])[idx] = p < np ? Amat[(size_t)p0 * NJ * 12 + idx] : 0.f;
    }
    for (int idx = tid; idx < PB * NB; idx += 256) {
        const int p = idx / NB;
        (&s_beta[0][0])[idx] = p < np ? betas[(size_t)p0 * NB + idx] : 0.f;
    }
    __syncthreads();
    const int v = min(blockIdx.x * 64 + lane, NV - 1);            // tail lanes recompute the last vertex (no divergent barrier)
    // pose blend shapes, this wave's quarter of  pose_feature @ posedirs   (smpl.py:167-170)
    float po[PB][3];
#pragma unroll
    for (int p = 0; p < PB; ++p) po[p][0] = po[p][1] = po[p][2] = 0.f;
    const float* pdv = pd + (size_t)v * 3;
    const int k0 = wave * KQ, k1 = min(NPF, k0 + KQ);
#pragma unroll 4
    for (int k = k0; k < k1; ++k) {
        const float d0 = pdv[(size_t)k * (NV * 3) + 0], d1 = pdv[(size_t)k * (NV * 3) + 1], d2 = pdv[(size_t)k * (NV * 3) + 2];
#pragma unroll
        for (int p4 = 0; p4 < PB; p4 += 4) {
            const float4 f = *reinterpret_cast<const float4*>(&s_pf[k][p4]);
            po[p4 + 0][0] = fmaf(f.x, d0, po[p4 + 0][0]); po[p4 + 0][1] = fmaf(f.x, d1, po[p4 + 0][1]); po[p4 + 0][2] = fmaf(f.x, d2, po[p4 + 0][2]);
            po[p4 + 1][0] = fmaf(f.y, d0, po[p4 + 1][0]); po[p4 + 1][1] = fmaf(f.y, d1, po[p4 + 1][1]); po[p4 + 1][2] = fmaf(f.y, d2, po[p4 + 1][2]);
            po[p4 + 2][0] = fmaf(f.z, d0, po[p4 + 2][0]); po[p4 + 2][1] = fmaf(f.z, d1, po[p4 + 2][1]); po[p4 + 2][2] = fmaf(f.z, d2, po[p4 + 2][2]);
            po[p4 + 3][0] = fmaf(f.w, d0, po[p4 + 3][0]); po[p4 + 3][1] = fmaf(f.w, d1, po[p4 + 3][1]); po[p4 + 3][2] = fmaf(f.w, d2, po[p4 + 3][2]);
        }
    }
#pragma unroll
    for (int p = 0; p < PB; ++p)
#pragma unroll
        for (int k = 0; k < 3; ++k) s_po[wave][p * 3 + k][lane] = po[p][k];
    __syncthreads();
    if (blockIdx.x * 64 + lane >= NV) return;
    // this wave's persons: shape blend, sum of the pose-blend quarters, skinning
    const float t0 = vt[v * 3 + 0], t1 = vt[v * 3 + 1], t2 = vt[v * 3 + 2];
    float sdv[3][NB];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int l = 0; l < NB; ++l) sdv[k][l] = sd[((size_t)v * 3 + k) * NB + l];
    float w[NJ];
#pragma unroll
    for (int j4 = 0; j4 < NJ; j4 += 4) {
        const float4 t = *reinterpret_cast<const float4*>(lbsw + (size_t)v * NJ + j4);
        w[j4] = t.x; w[j4 + 1] = t.y; w[j4 + 2] = t.z; w[j4 + 3] = t.w;
    }
#pragma unroll
    for (int pp = 0; pp < PB / SKIN_WAVES; ++pp) {
        const int p = wave * (PB / SKIN_WAVES) + pp;
        if (p >= np) continue;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;            // einsum('bl,mkl->bmk') then + v_template (smpl.py:153)
#pragma unroll
        for (int l = 0; l < NB; ++l) {
            const float bb = s_beta[p][l];
            a0 = fmaf(bb, sdv[0][l], a0); a1 = fmaf(bb, sdv[1][l], a1); a2 = fmaf(bb, sdv[2][l], a2);
        }
        float q[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            q[k] = (s_po[0][p * 3 + k][lane] + s_po[1][p * 3 + k][lane]) + (s_po[2][p * 3 + k][lane] + s_po[3][p * 3 + k][lane]);
        const float x = q[0] + (t0 + a0), y = q[1] + (t1 + a1), z = q[2] + (t2 + a2);
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {              // T = W @ A   (smpl.py:179)
            const float4 a0v = *reinterpret_cast<const float4*>(&s_A[p][j][0]);
            const float4 a1v = *reinterpret_cast<const float4*>(&s_A[p][j][4]);
            const float4 a2v = *reinterpret_cast<const float4*>(&s_A[p][j][8]);
            T[0] = fmaf(w[j], a0v.x, T[0]); T[1] = fmaf(w[j], a0v.y, T[1]); T[2] = fmaf(w[j], a0v.z, T[2]); T[3] = fmaf(w[j], a0v.w, T[3]);
            T[4] = fmaf(w[j], a1v.x, T[4]); T[5] = fmaf(w[j], a1v.y, T[5]); T[6] = fmaf(w[j], a1v.z, T[6]); T[7] = fmaf(w[j], a1v.w, T[7]);
            T[8] = fmaf(w[j], a2v.x, T[8]); T[9] = fmaf(w[j], a2v.y, T[9]); T[10] = fmaf(w[j], a2v.z, T[10]); T[11] = fmaf(w[j], a2v.w, T[11]);
        }
        float* o = verts + ((size_t)(p0 + p) * NV + v) * 3;   // v_homo = T @ [v_posed,1]  (smpl.py:185)
        o[0] = fmaf(T[0], x, fmaf(T[1], y, fmaf(T[2], z, T[3])));
        o[1] = fmaf(T[4], x, fmaf(T[5], y, fmaf(T[6], z, T[7])));
        o[2] = fmaf(T[8], x, fmaf(T[9], y, fmaf(T[10], z, T[11])));
    }
}

// ---- kernel C: joint regression ---------------------------------------------------------------
// One workgroup per (person, quarter of the 26 regressed joints): at N = 64 a workgroup per person left three quarters
// of the CUs idle and made this the longest SMPL kernel.
constexpr int JSPLIT = 4, RPS = (NREG + JSPLIT - 1) / JSPLIT;          // 7 regressors per workgroup
__global__ __launch_bounds__(256) void smpl_joints_kernel(const float* __restrict__ verts, const float* __restrict__ reg,
                                                           const int* __restrict__ pick, float* __restrict__ joints) {
    __shared__ float s_part[4][RPS * 3];
    const int n = blockIdx.x, r0 = blockIdx.y * RPS, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nr = min(RPS, NREG - r0);
    const float* vn = verts + (size_t)n * NV * 3;
    float acc[RPS][3];
#pragma unroll
    for (int r = 0; r < RPS; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 0.f;
    for (int v = tid; v < NV; v += 256) {
        const float x = vn[v * 3], y = vn[v * 3 + 1], z = vn[v * 3 + 2];
#pragma unroll
        for (int r = 0; r < RPS; ++r) {
            const float w = reg[(size_t)min(r0 + r, NREG - 1) * NV + v];
            acc[r][0] = fmaf(w, x, acc[r][0]); acc[r][1] = fmaf(w, y, acc[r][1]); acc[r][2] = fmaf(w, z, acc[r][2]);
        }
    }
#pragma unroll
    for (int r = 0; r < RPS; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float a = acc[r][k];
            for (int d = 32; d > 0; d >>= 1) a += __shfl_xor(a, d);
            if (lane == 0) s_part[wave][r * 3 + k] = a;
        }
    __syncthreads();
    float* jn = joints + (size_t)n * NJOUT * 3;
    if (tid < nr * 3)                                        // joints 45..70: extra9 then h36m17 (smpl.py:26-29)
        jn[(NJ + NPICK + r0) * 3 + tid] = (s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid]);
    if (blockIdx.y == 0 && tid < NPICK * 3)                  // joints 24..44: vertex picks (smpl.py:25)
        jn[NJ * 3 + tid] = vn[pick[tid / 3] * 3 + tid % 3];
}

// root alignment (smpl.py:102-106): root = joints[45:47].mean(0); joints -= root (vertices: smpl_root_sub_kernel)
__global__ __launch_bounds__(256) void smpl_root_joints_kernel(float* __restrict__ joints, float* __restrict__ root) {
    __shared__ float s_root[3];
    const int n = blockIdx.x, tid = threadIdx.x;
    float* jn = joints + (size_t)n * NJOUT * 3;
    if (tid < 3) {
        const float r0 = (jn[45 * 3 + tid] + jn[46 * 3 + tid]) / 2.f;
        s_root[tid] = r0;
        root[n * 3 + tid] = r0;
    }
    __syncthreads();
    if (tid < NJOUT * 3) jn[tid] -= s_root[tid % 3];
}

__global__ void smpl_root_sub_kernel(float* __restrict__ verts, const float* __restrict__ root, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / (NV * 3);
        verts[i] -= root[n * 3 + (i % 3)];
    }
}

}  // namespace romp

using namespace romp;

struct smpl_ctx {
    int nb = 10;
    int cap = 0;
    Parents par;
    float *vt = nullptr, *sd = nullptr, *pd = nullptr, *lbsw = nullptr, *reg = nullptr, *Jt = nullptr, *Js = nullptr;
    int* pick = nullptr;
    float *pose_feat = nullptr, *Amat = nullptr, *root = nullptr;
};

static int smpl_reserve(smpl_ctx* c, int N) {
    if (N <= c->cap) return ROMP_OK;
    if (c->pose_feat) hipFree(c->pose_feat);
    if (c->Amat) hipFree(c->Amat);
    if (c->root) hipFree(c->root);
    c->pose_feat = c->Amat = c->root = nullptr;
    c->cap = 0;
    ROMP_HIP_CHECK(hipMalloc((void**)&c->pose_feat, (size_t)N * NPF * 4));
    ROMP_HIP_CHECK(hipMalloc((void**)&c->Amat, (size_t)N * NJ * 12 * 4));
    ROMP_HIP_CHECK(hipMalloc((void**)&c->root, (size_t)N * 3 * 4));
    c->cap = N;
    return ROMP_OK;
}

extern "C" {

int smpl_ctx_create(smpl_ctx** out, const float* v_template, const float* shapedirs, int n_betas,
                    const float* posedirs, const float* J_regressor, const float* lbs_weights,
                    const int64_t* parents_host, const float* J_regressor_extra9, const float* J_regressor_h36m17,
                    const int64_t* extra_idx_host, int max_persons, void* stream) {
    ROMP_REQUIRE(out && v_template && shapedirs && posedirs && J_regressor && lbs_weights && parents_host &&
                 J_regressor_extra9 && J_regressor_h36m17 && extra_idx_host, "smpl_ctx_create: null argument");
    ROMP_REQUIRE(n_betas == 10 || n_betas == 11, "smpl_ctx_create: n_betas %d (10 = SMPL, 11 = SMPL-A)", n_betas);
    hipStream_t st = (hipStream_t)stream;
    smpl_ctx* c = new smpl_ctx();
    c->nb = n_betas;
    for (int j = 0; j < NJ; ++j) {
        c->par.p[j] = (int)parents_host[j];
        if (j > 0 && (c->par.p[j] < 0 || c->par.p[j] >= j)) {
            set_error("smpl_ctx_create: parents[%d]=%d is not an earlier joint", j, c->par.p[j]);
            delete c;
            return ROMP_EINVAL;
        }
    }
    int pick_h[NPICK];
    for (int i = 0; i < NPICK; ++i) {
        pick_h[i] = (int)extra_idx_host[i];
        if (pick_h[i] < 0 || pick_h[i] >= NV) { set_error("smpl_ctx_create: extra index out of range"); delete c; return ROMP_EINVAL; }
    }
#define SMPL_ALLOC_COPY(dst, src, count)                                                           \
    do {                                                                                           \
        if (hipMalloc((void**)&(dst), (size_t)(count) * 4) != hipSuccess ||                        \
            hipMemcpyAsync((dst), (src), (size_t)(count) * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) { \
            set_error("smpl_ctx_create: alloc/copy of " #dst " failed");                           \
            smpl_ctx_destroy(c);                                                                   \
            return ROMP_EHIP;                                                                      \
        }                                                                                          \
    } while (0)
    SMPL_ALLOC_COPY(c->vt, v_template, NV * 3);
    SMPL_ALLOC_COPY(c->sd, shapedirs, (size_t)NV * 3 * n_betas);
    SMPL_ALLOC_COPY(c->pd, posedirs, (size_t)NPF * NV * 3);
    SMPL_ALLOC_COPY(c->lbsw, lbs_weights, (size_t)NV * NJ);
#undef SMPL_ALLOC_COPY
    if (hipMalloc((void**)&c->reg, (size_t)NREG * NV * 4) != hipSuccess || hipMalloc((void**)&c->Jt, NJ * 3 * 4) != hipSuccess ||
        hipMalloc((void**)&c->Js, (size_t)NJ * 3 * n_betas * 4) != hipSuccess || hipMalloc((void**)&c->pick, NPICK * 4) != hipSuccess) {
        set_error("smpl_ctx_create: hipMalloc failed");
        smpl_ctx_destroy(c);
        return ROMP_ENOMEM;
    }
    hipMemcpyAsync(c->reg, J_regressor_extra9, (size_t)9 * NV * 4, hipMemcpyDeviceToDevice, st);
    hipMemcpyAsync(c->reg + (size_t)9 * NV, J_regressor_h36m17, (size_t)17 * NV * 4, hipMemcpyDeviceToDevice, st);
    hipMemcpyAsync(c->pick, pick_h, NPICK * 4, hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(smpl_prep_kernel, dim3(NJ, 3 * (n_betas + 1)), dim3(256), 0, st, J_regressor, c->vt, c->sd, n_betas,
                       c->Jt, c->Js);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {
        set_error("smpl_ctx_create: constant preparation failed");
        smpl_ctx_destroy(c);
        return ROMP_EHIP;
    }
    int rc = smpl_reserve(c, max_persons > 0 ? max_persons : 64);
    if (rc) { smpl_ctx_destroy(c); return rc; }
    *out = c;
    return ROMP_OK;
}

int smpl_forward(smpl_ctx* c, const float* betas, int n_betas, const float* thetas, int N, int root_align,
                 float* verts, float* joints, void* stream) {
    ROMP_REQUIRE(c && betas && thetas && verts && joints && N >= 0, "smpl_forward: bad arguments");
    ROMP_REQUIRE(n_betas == c->nb, "smpl_forward: n_betas %d but context was built with %d", n_betas, c->nb);
    if (N == 0) return ROMP_OK;
    int rc = smpl_reserve(c, N);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(smpl_pose_kernel, dim3(N), dim3(64), 0, st, betas, c->nb, thetas, c->Jt, c->Js, c->par,
                       c->pose_feat, c->Amat, joints);
    ROMP_HIP_CHECK(hipGetLastError());
    dim3 grid((NV + 63) / 64, (N + PB - 1) / PB);
    if (c->nb == 10)
        hipLaunchKernelGGL(smpl_skin_kernel<10>, grid, dim3(256), 0, st, betas, c->pose_feat, c->Amat, c->vt, c->sd, c->pd,
                           c->lbsw, N, verts);
    else
        hipLaunchKernelGGL(smpl_skin_kernel<11>, grid, dim3(256), 0, st, betas, c->pose_feat, c->Amat, c->vt, c->sd, c->pd,
                           c->lbsw, N, verts);
    ROMP_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(smpl_joints_kernel, dim3(N, JSPLIT), dim3(256), 0, st, verts, c->reg, c->pick, joints);
    ROMP_HIP_CHECK(hipGetLastError());
    if (root_align) {
        hipLaunchKernelGGL(smpl_root_joints_kernel, dim3(N), dim3(256), 0, st, joints, c->root);
        const size_t total = (size_t)N * NV * 3;
        size_t blocks = (total + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(smpl_root_sub_kernel, dim3((unsigned)blocks), dim3(256), 0, st, verts, c->root, total);
        ROMP_HIP_CHECK(hipGetLastError());
    }
    return ROMP_OK;
}

void smpl_ctx_destroy(smpl_ctx* c) {
    if (!c) return;
    float* ptrs[] = {c->vt, c->sd, c->pd, c->lbsw, c->reg, c->Jt, c->Js, c->pose_feat, c->Amat, c->root};
    for (float* p : ptrs)
        if (p) hipFree(p);
    if (c->pick) hipFree(c->pick);
    delete c;
}

}  // extern "C"
