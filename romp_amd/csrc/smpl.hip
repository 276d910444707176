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
//            smpl.py:153-156, without the 6890-long reduction), parent chain walked LEVEL by level of the
//            kinematic tree (8 dependent steps for SMPL instead of 23), A matrices;
//   kernel B (one lane per vertex, 64 vertices x PB = 16 persons per workgroup)  shape blend + 207-term pose blend +
//            24-joint skinning in registers (packed-f32 FMAs); posedirs (17 MB) is the only large stream: the
//            person groups of one vertex tile are placed on the SAME XCD back to back so that the tile is fetched
//            from HBM once and re-read from that XCD's L2; the finished vertices also give this tile's share of the
//            26 regressed joints (partial sums, one row per tile);
//   kernel C (one workgroup per person)  sums the 108 per-tile partials in a fixed order, picks 21 vertices,
//            optional root alignment of the joints.
// All float32; bound = HBM (constants 20 MB once + 82.7 KB written per person).
#include "common.h"
#include <vector>

namespace romp {

constexpr int NV = 6890, NJ = 24, NPF = 207, NJOUT = 71, NREG = 26, NPICK = 21;
constexpr int PB = 16;                                 // persons per workgroup in the skinning kernel
constexpr int NT = (NV + 63) / 64;                    // 108 vertex tiles
constexpr int MAXLV = NJ;

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32

struct Parents { int p[NJ]; };
// joints ordered by depth in the kinematic tree: level l = order[start[l] .. start[l + 1])
struct Levels { int n; int start[MAXLV + 1]; int order[NJ]; };

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
// pose_feat is written in the layout the skinning kernel stages: [person group][207][PB]
template <int NB>
__global__ __launch_bounds__(64) void smpl_pose_kernel(const float* __restrict__ betas,
                                                        const float* __restrict__ thetas,
                                                        const float* __restrict__ Jt, const float* __restrict__ Js,
                                                        const int* __restrict__ sched, float* __restrict__ pose_feat,
                                                        float* __restrict__ Amat, float* __restrict__ joints) {
    __shared__ float sR[NJ][9], sJ[NJ][3], sG[NJ][12];
    const int n = blockIdx.x, lane = threadIdx.x;
    if (lane < NJ) {
        const float* r = thetas + (size_t)n * 72 + lane * 3;
        const float rx0 = r[0], ry0 = r[1], rz0 = r[2];
        float jt[3], js[3][NB], bt[NB];                      // every load of the prologue in flight before the first use
#pragma unroll
        for (int l = 0; l < NB; ++l) bt[l] = betas[(size_t)n * NB + l];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            jt[k] = Jt[lane * 3 + k];
#pragma unroll
            for (int l = 0; l < NB; ++l) js[k][l] = Js[(lane * 3 + k) * NB + l];
        }
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
            float* pf = pose_feat + (size_t)(n / PB) * NPF * PB + (n % PB);
#pragma unroll
            for (int e = 0; e < 9; ++e) pf[((lane - 1) * 9 + e) * PB] = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float a = jt[k];
#pragma unroll
            for (int l = 0; l < NB; ++l) a = fmaf(bt[l], js[k][l], a);
            sJ[lane][k] = a;
        }
    }
    __syncthreads();
    // kinematic chain (smpl.py:269-275): G[i] = G[parent] * [R_i | J_i - J_parent], one tree level per step; five groups of
    // 12 lanes each own one joint of the level (one element of its 3x4 transform per lane)
    const int grp = lane / 12, el = lane % 12, rr = el / 4, cc = el % 4;
    if (lane < 12) sG[0][lane] = (cc < 3) ? sR[0][rr * 3 + cc] : sJ[0][rr];
    __syncthreads();
    const int n_lv = sched[0];
    for (int lv = 1; lv < n_lv; ++lv) {
        const int q1 = sched[1 + lv + 1];
        for (int q = sched[1 + lv] + grp; q < q1 && grp < 5; q += 5) {
            const int i = sched[1 + (MAXLV + 1) + q], p = sched[1 + (MAXLV + 1) + NJ + i];
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
            sG[i][el] = v;
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

// ---- kernel B: per-vertex blend shapes + skinning + this tile's share of the joint regression ----------------------
// Workgroup = 4 waves on the SAME 64 vertices and PB persons.  Each wave accumulates a quarter of the 207 pose-blend terms
// (the only long dependent loop: strided posedirs loads), the partial sums meet in LDS, then wave w finishes persons
// 4w .. 4w+3 (shape blend, skinning) and the workgroup reduces reg[r][tile] . vertex for the 26 regressors.
// Workgroup id -> (tile, person group): ids that differ by 8 run on the same XCD, so the person groups of a tile get
// consecutive slots of ONE XCD (posedirs tile: HBM once, then L2).
constexpr int SKIN_WAVES = 4, KQ = (NPF + SKIN_WAVES - 1) / SKIN_WAVES;   // 52 terms per wave
constexpr int PPW = PB / SKIN_WAVES;                                      // persons finished per wave
constexpr int U_FLOATS = SKIN_WAVES * PB * 3 * 64;                        // blend partials, then the vertex tile
constexpr int KC = 13, NCH = KQ / KC;                                     // posedirs terms per register chunk, chunks per wave
static_assert(KQ == KC * NCH, "chunking of the pose-blend quarter");

__device__ __forceinline__ void pd_load(float (&d)[KC][3], const float* __restrict__ pdv, int kbase) {
#pragma unroll
    for (int i = 0; i < KC; ++i) {
        const size_t off = (size_t)min(kbase + i, NPF - 1) * (NV * 3);
        d[i][0] = pdv[off]; d[i][1] = pdv[off + 1]; d[i][2] = pdv[off + 2];
    }
}

__device__ __forceinline__ void pd_blend(f2 (&po)[PB / 2][3], const float (&d)[KC][3], const float (*s_pf)[PB], int kbase) {
#pragma unroll
    for (int i = 0; i < KC; ++i) {
        const int k = kbase + i;
        const bool in = k < NPF;                                  // the last quarter is one term short
        const float d0 = in ? d[i][0] : 0.f, d1 = in ? d[i][1] : 0.f, d2 = in ? d[i][2] : 0.f;
        const f2 dd0{d0, d0}, dd1{d1, d1}, dd2{d2, d2};
        const float* row = s_pf[min(k, NPF - 1)];
#pragma unroll
        for (int p4 = 0; p4 < PB; p4 += 4) {
            const float4 f = *reinterpret_cast<const float4*>(row + p4);
            const f2 fa{f.x, f.y}, fb{f.z, f.w};
            po[p4 / 2][0] = pk_fma(fa, dd0, po[p4 / 2][0]); po[p4 / 2][1] = pk_fma(fa, dd1, po[p4 / 2][1]); po[p4 / 2][2] = pk_fma(fa, dd2, po[p4 / 2][2]);
            po[p4 / 2 + 1][0] = pk_fma(fb, dd0, po[p4 / 2 + 1][0]); po[p4 / 2 + 1][1] = pk_fma(fb, dd1, po[p4 / 2 + 1][1]); po[p4 / 2 + 1][2] = pk_fma(fb, dd2, po[p4 / 2 + 1][2]);
        }
    }
}

// Measured alternatives (N = 64, whole SMPL call): pose features and the relative transforms read through the scalar cache
// as SGPR operands instead of LDS (no staging, one barrier less): 44 us against 37 us -- the s_load round trips cannot be
// prefetched deep enough with ~100 SGPRs.
constexpr int SVS = 196, RS = 68;                                         // person / regressor strides of the LDS tiles (16-byte rows)
constexpr int RPAIRS = NREG / 2;                                          // joint regressors in pairs, one (person, pair) per thread
static_assert(NREG % 2 == 0 && PB * RPAIRS <= 256, "joint partials: one pass");
template <int NB>
__global__ __launch_bounds__(256) void smpl_skin_kernel(const float* __restrict__ betas, const float* __restrict__ pose_feat,
                                                         const float* __restrict__ Amat, const float* __restrict__ vt,
                                                         const float* __restrict__ sd, const float* __restrict__ pd,
                                                         const float* __restrict__ lbsw, const float* __restrict__ reg, int N, int groups,
                                                         float* __restrict__ verts, float* __restrict__ jpart, int dbg) {
    __shared__ __attribute__((aligned(16))) float s_u[U_FLOATS];          // pose features, then blend partials, then the vertex tile
    __shared__ __attribute__((aligned(16))) float s_A[5 * 256 * 4];       // PB x 24 x 12 (+ staging slack); later: the regressor tile [NREG][RS]
    __shared__ float s_beta[PB][NB];
    static_assert(NPF * PB <= U_FLOATS && 4 * 256 * 4 <= U_FLOATS && PB * SVS <= U_FLOATS && NREG * RS <= PB * NJ * 12 && NT % 4 == 0, "LDS aliasing");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = (slot / groups) * 8 + xcd, grp = slot % groups;
    if (tile >= NT) return;
    const int p0 = grp * PB;
    const int np = min(PB, N - p0);
    const bool live = tile * 64 + lane < NV;
    const int v = min(tile * 64 + lane, NV - 1);                  // tail lanes recompute the last vertex
    const float* pdv = pd + (size_t)v * 3;
    const int k0 = wave * KQ;
    // the first posedirs chunk and the staging loads leave together; nothing waits before everything is in flight
    float d[2][KC][3];
    pd_load(d[0], pdv, k0);
    constexpr int PF4 = NPF * PB / 4, A4 = PB * NJ * 12 / 4;     // 828 and 1152 16-byte units, contiguous per person group
    static_assert(PF4 <= 4 * 256 && A4 <= 5 * 256 && PB * NB <= 256, "staging loops");
    const float4* src_pf = reinterpret_cast<const float4*>(pose_feat + (size_t)grp * NPF * PB);
    const float4* src_A = reinterpret_cast<const float4*>(Amat + (size_t)p0 * NJ * 12);
    float4 tpf[4], tA[5];
#pragma unroll
    for (int i = 0; i < 4; ++i) tpf[i] = src_pf[min(tid + 256 * i, PF4 - 1)];
#pragma unroll
    for (int i = 0; i < 5; ++i) tA[i] = src_A[min(tid + 256 * i, A4 - 1)];
    const float tb = (tid < PB * NB && tid / NB < np) ? betas[(size_t)p0 * NB + tid] : 0.f;
    // unconditional stores (the units past the end land in slack): a predicated store would pull its load into the branch
#pragma unroll
    for (int i = 0; i < 4; ++i) reinterpret_cast<float4*>(s_u)[tid + 256 * i] = tpf[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) reinterpret_cast<float4*>(s_A)[tid + 256 * i] = tA[i];
    if (tid < PB * NB) (&s_beta[0][0])[tid] = tb;
    __syncthreads();
    // pose blend shapes, this wave's quarter of  pose_feature @ posedirs   (smpl.py:167-170); persons in packed pairs,
    // posedirs in register chunks of 13 terms, the next chunk loading while this one is used
    const float(*s_pf)[PB] = reinterpret_cast<const float(*)[PB]>(s_u);   // [NPF][PB]
    f2 po[PB / 2][3];
#pragma unroll
    for (int p = 0; p < PB / 2; ++p) po[p][0] = po[p][1] = po[p][2] = f2{0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH && !(dbg & 2)) pd_load(d[(c + 1) & 1], pdv, k0 + (c + 1) * KC);
        if (!(dbg & 8)) pd_blend(po, d[c & 1], s_pf, k0 + c * KC);
    }
    // per-vertex constants of the finishing phase and the regressor tile: in flight across the exchange of the partials
    const float t0 = vt[v * 3 + 0], t1 = vt[v * 3 + 1], t2 = vt[v * 3 + 2];
    float sdv[3][NB];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int l = 0; l < NB; ++l) sdv[k][l] = sd[((size_t)v * 3 + k) * NB + l];
    f2 w[NJ];
#pragma unroll
    for (int j4 = 0; j4 < NJ; j4 += 4) {
        const float4 t = *reinterpret_cast<const float4*>(lbsw + (size_t)v * NJ + j4);
        w[j4] = f2{t.x, t.x}; w[j4 + 1] = f2{t.y, t.y}; w[j4 + 2] = f2{t.z, t.z}; w[j4 + 3] = f2{t.w, t.w};
    }
    float treg[7];
    static_assert(NREG * 64 <= 7 * 256, "regressor tile staging");
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int idx = min(tid + 256 * i, NREG * 64 - 1), vv = tile * 64 + (idx & 63);
        const float x = reg[(size_t)(idx >> 6) * NV + min(vv, NV - 1)];
        treg[i] = vv < NV ? x : 0.f;                              // zero weight for the recomputed tail lanes
    }
    __syncthreads();                                              // every wave is done with the pose features: s_u becomes the partials
    float(*s_po)[PB * 3][64] = reinterpret_cast<float(*)[PB * 3][64]>(s_u);
#pragma unroll
    for (int p = 0; p < PB / 2; ++p)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            s_po[wave][(2 * p) * 3 + k][lane] = po[p][k].x;
            s_po[wave][(2 * p + 1) * 3 + k][lane] = po[p][k].y;
        }
    __syncthreads();
    // this wave's persons: shape blend, sum of the pose-blend quarters, skinning
    float out[PPW][3];
#pragma unroll
    for (int pp = 0; pp < PPW; ++pp) {
        const int p = wave * PPW + pp;
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
        f2 T[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) T[e] = f2{0.f, 0.f};
        const float* Ap = s_A + p * NJ * 12;
        if (!(dbg & 4))
#pragma unroll
        for (int j = 0; j < NJ; ++j) {              // T = W @ A   (smpl.py:179)
            const float4 a0v = *reinterpret_cast<const float4*>(Ap + j * 12);
            const float4 a1v = *reinterpret_cast<const float4*>(Ap + j * 12 + 4);
            const float4 a2v = *reinterpret_cast<const float4*>(Ap + j * 12 + 8);
            T[0] = pk_fma(w[j], f2{a0v.x, a0v.y}, T[0]); T[1] = pk_fma(w[j], f2{a0v.z, a0v.w}, T[1]);
            T[2] = pk_fma(w[j], f2{a1v.x, a1v.y}, T[2]); T[3] = pk_fma(w[j], f2{a1v.z, a1v.w}, T[3]);
            T[4] = pk_fma(w[j], f2{a2v.x, a2v.y}, T[4]); T[5] = pk_fma(w[j], f2{a2v.z, a2v.w}, T[5]);
        }
        out[pp][0] = fmaf(T[0].x, x, fmaf(T[0].y, y, fmaf(T[1].x, z, T[1].y)));   // v_homo = T @ [v_posed,1]  (smpl.py:185)
        out[pp][1] = fmaf(T[2].x, x, fmaf(T[2].y, y, fmaf(T[3].x, z, T[3].y)));
        out[pp][2] = fmaf(T[4].x, x, fmaf(T[4].y, y, fmaf(T[5].x, z, T[5].y)));
        if (live && p < np) {
            float* o = verts + ((size_t)(p0 + p) * NV + v) * 3;
            o[0] = out[pp][0]; o[1] = out[pp][1]; o[2] = out[pp][2];
        }
    }
    __syncthreads();                                              // partials and A are consumed: vertex tile + regressor tile
#pragma unroll
    for (int pp = 0; pp < PPW; ++pp)
#pragma unroll
        for (int k = 0; k < 3; ++k) s_u[(wave * PPW + pp) * SVS + k * 64 + lane] = out[pp][k];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int idx = tid + 256 * i;
        if (idx < NREG * 64) s_A[(idx >> 6) * RS + (idx & 63)] = treg[i];
    }
    __syncthreads();
    // joints 45..70 (extra9 then h36m17, smpl.py:26-29): this tile's 64 terms of  regressor @ vertices; one thread per
    // (person, regressor pair), 16-byte LDS reads along the vertices
    if (tid < PB * RPAIRS && !(dbg & 1)) {
        const int p = tid % PB, rp = tid / PB;
        const float* sv = s_u + p * SVS;
        const float* sr = s_A + rp * 2 * RS;
        f2 acc[3] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};  // [component] over the regressor pair
#pragma unroll 4
        for (int i = 0; i < 64; i += 4) {
            const float4 x4 = *reinterpret_cast<const float4*>(sv + i), y4 = *reinterpret_cast<const float4*>(sv + 64 + i),
                         z4 = *reinterpret_cast<const float4*>(sv + 128 + i);
            const float4 r0 = *reinterpret_cast<const float4*>(sr + i), r1 = *reinterpret_cast<const float4*>(sr + RS + i);
#define JSTEP(c)                                                             \
            acc[0] = pk_fma(f2{r0.c, r1.c}, f2{x4.c, x4.c}, acc[0]);        \
            acc[1] = pk_fma(f2{r0.c, r1.c}, f2{y4.c, y4.c}, acc[1]);        \
            acc[2] = pk_fma(f2{r0.c, r1.c}, f2{z4.c, z4.c}, acc[2]);
            JSTEP(x) JSTEP(y) JSTEP(z) JSTEP(w)
#undef JSTEP
        }
        if (p < np) {
            float* o = jpart + (((size_t)(p0 + p) * NT + tile) * NREG + rp * 2) * 3;
            o[0] = acc[0].x; o[1] = acc[1].x; o[2] = acc[2].x;
            o[3] = acc[0].y; o[4] = acc[1].y; o[5] = acc[2].y;
        }
    }
}

// ---- kernel C: joints 24..70 + optional root alignment ------------------------------------------
// One workgroup per person: the 108 per-tile partial sums are added in tile order (deterministic), the 21 picked
// vertices copied; with root alignment (smpl.py:102-106) root = joints[45:47].mean(0) is subtracted from all 71 joints
// here and from the vertices by smpl_root_sub_kernel.
__global__ __launch_bounds__(256) void smpl_joints_kernel(const float* __restrict__ verts, const float* __restrict__ jpart,
                                                           const int* __restrict__ pick, int root_align, float* __restrict__ joints,
                                                           float* __restrict__ root) {
    __shared__ float s_j[NJOUT * 3];
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* vn = verts + (size_t)n * NV * 3;
    float* jn = joints + (size_t)n * NJOUT * 3;
    __shared__ float s_part[3][NREG * 3];
    static_assert(NT % 3 == 0 && 3 * NREG * 3 <= 256 && NREG * 3 + NPICK * 3 + NJ * 3 <= 256, "thread roles");
    if (tid < 3 * NREG * 3) {                                // 78 outputs x 3 thirds of the tiles, every load in flight at once
        const int e = tid % (NREG * 3), part = tid / (NREG * 3);
        const float* src = jpart + ((size_t)n * NT + part * (NT / 3)) * NREG * 3 + e;
        float t[NT / 3];
#pragma unroll
        for (int u = 0; u < NT / 3; ++u) t[u] = src[(size_t)u * NREG * 3];
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NT / 3; ++u) acc[u & 3] += t[u];
        s_part[part][e] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    __syncthreads();
    if (tid < NREG * 3) {
        s_j[(NJ + NPICK) * 3 + tid] = (s_part[0][tid] + s_part[1][tid]) + s_part[2][tid];
    } else if (tid < NREG * 3 + NPICK * 3) {
        const int e = tid - NREG * 3;
        s_j[NJ * 3 + e] = vn[pick[e / 3] * 3 + e % 3];       // joints 24..44: vertex picks (smpl.py:25)
    } else if (tid < NREG * 3 + NPICK * 3 + NJ * 3) {
        const int e = tid - NREG * 3 - NPICK * 3;
        s_j[e] = jn[e];                                      // joints 0..23 come from the pose kernel
    }
    __syncthreads();
    if (tid < NJOUT * 3) {
        float v = s_j[tid];
        if (root_align) {
            const float r0 = (s_j[45 * 3 + tid % 3] + s_j[46 * 3 + tid % 3]) / 2.f;
            v -= r0;
            if (tid < 3) root[n * 3 + tid] = r0;
        }
        if (root_align || tid >= NJ * 3) jn[tid] = v;
    }
}

__global__ __launch_bounds__(256) void smpl_root_sub_kernel(float* __restrict__ verts, const float* __restrict__ root) {
    const int n = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;       // one float2 per lane
    static_assert(NV * 3 % 2 == 0, "person stride is a whole number of float2");
    if (i >= NV * 3 / 2) return;
    const float r0 = root[n * 3], r1 = root[n * 3 + 1], r2 = root[n * 3 + 2];
    float2* p = reinterpret_cast<float2*>(verts + (size_t)n * NV * 3) + i;
    float2 t = *p;
    const int c = (i * 2) % 3;                                          // component of t.x
    t.x -= c == 0 ? r0 : (c == 1 ? r1 : r2);
    t.y -= c == 0 ? r1 : (c == 1 ? r2 : r0);
    *p = t;
}

}  // namespace romp

using namespace romp;

struct smpl_ctx {
    int nb = 10;
    int cap = 0;
    Parents par;
    int dbg = 0;                   // ROMP_SMPL_DBG: phase knock-outs for timing experiments (results are wrong when set)
    int* sched = nullptr;          // [n_levels, start[MAXLV + 1], order[NJ], parent[NJ]] for the pose kernel
    float* jpart = nullptr;        // (cap, NT, NREG, 3) per-tile joint partial sums
    float *vt = nullptr, *sd = nullptr, *pd = nullptr, *lbsw = nullptr, *reg = nullptr, *Jt = nullptr, *Js = nullptr;
    int* pick = nullptr;
    float *pose_feat = nullptr, *Amat = nullptr, *root = nullptr;
};

static int smpl_reserve(smpl_ctx* c, int N) {
    if (N <= c->cap) return ROMP_OK;
    if (c->pose_feat) hipFree(c->pose_feat);
    if (c->Amat) hipFree(c->Amat);
    if (c->root) hipFree(c->root);
    if (c->jpart) hipFree(c->jpart);
    c->pose_feat = c->Amat = c->root = c->jpart = nullptr;
    c->cap = 0;
    const size_t cap = (size_t)(N + PB - 1) / PB * PB;        // whole person groups: the skinning kernel stages PB persons at a time
    ROMP_HIP_CHECK(hipMalloc((void**)&c->pose_feat, cap * NPF * 4));
    ROMP_HIP_CHECK(hipMalloc((void**)&c->Amat, cap * NJ * 12 * 4));
    ROMP_HIP_CHECK(hipMalloc((void**)&c->root, cap * 3 * 4));
    ROMP_HIP_CHECK(hipMalloc((void**)&c->jpart, cap * NT * NREG * 3 * 4));
    ROMP_HIP_CHECK(hipMemset(c->pose_feat, 0, cap * NPF * 4));   // rows of a partial last group stay finite
    ROMP_HIP_CHECK(hipMemset(c->Amat, 0, cap * NJ * 12 * 4));
    c->cap = (int)cap;
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
    if (const char* e = getenv("ROMP_SMPL_DBG")) c->dbg = atoi(e);
    for (int j = 0; j < NJ; ++j) {
        c->par.p[j] = (int)parents_host[j];
        if (j > 0 && (c->par.p[j] < 0 || c->par.p[j] >= j)) {
            set_error("smpl_ctx_create: parents[%d]=%d is not an earlier joint", j, c->par.p[j]);
            delete c;
            return ROMP_EINVAL;
        }
    }
    // kinematic tree by depth (parents precede children, so one pass gives the depths)
    int sched_h[1 + (MAXLV + 1) + 2 * NJ] = {0}, depth[NJ] = {0}, n_lv = 1;
    for (int j = 1; j < NJ; ++j) {
        depth[j] = depth[c->par.p[j]] + 1;
        if (depth[j] + 1 > n_lv) n_lv = depth[j] + 1;
    }
    sched_h[0] = n_lv;
    for (int lv = 0, q = 0; lv < n_lv; ++lv) {
        sched_h[1 + lv] = q;
        for (int j = 0; j < NJ; ++j)
            if (depth[j] == lv) sched_h[1 + (MAXLV + 1) + q++] = j;
        sched_h[1 + lv + 1] = q;
    }
    for (int j = 0; j < NJ; ++j) sched_h[1 + (MAXLV + 1) + NJ + j] = j ? c->par.p[j] : 0;
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
        hipMalloc((void**)&c->Js, (size_t)NJ * 3 * n_betas * 4) != hipSuccess || hipMalloc((void**)&c->pick, NPICK * 4) != hipSuccess ||
        hipMalloc((void**)&c->sched, sizeof(sched_h)) != hipSuccess) {
        set_error("smpl_ctx_create: hipMalloc failed");
        smpl_ctx_destroy(c);
        return ROMP_ENOMEM;
    }
    hipMemcpyAsync(c->reg, J_regressor_extra9, (size_t)9 * NV * 4, hipMemcpyDeviceToDevice, st);
    hipMemcpyAsync(c->reg + (size_t)9 * NV, J_regressor_h36m17, (size_t)17 * NV * 4, hipMemcpyDeviceToDevice, st);
    hipMemcpyAsync(c->pick, pick_h, NPICK * 4, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(c->sched, sched_h, sizeof(sched_h), hipMemcpyHostToDevice, st);
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
    if (c->nb == 10)
        hipLaunchKernelGGL(smpl_pose_kernel<10>, dim3(N), dim3(64), 0, st, betas, thetas, c->Jt, c->Js, c->sched, c->pose_feat, c->Amat, joints);
    else
        hipLaunchKernelGGL(smpl_pose_kernel<11>, dim3(N), dim3(64), 0, st, betas, thetas, c->Jt, c->Js, c->sched, c->pose_feat, c->Amat, joints);
    ROMP_HIP_CHECK(hipGetLastError());
    const int groups = (N + PB - 1) / PB;
    const dim3 grid(8 * ((NT + 7) / 8) * groups);
    if (c->nb == 10)
        hipLaunchKernelGGL(smpl_skin_kernel<10>, grid, dim3(256), 0, st, betas, c->pose_feat, c->Amat, c->vt, c->sd, c->pd,
                           c->lbsw, c->reg, N, groups, verts, c->jpart, c->dbg);
    else
        hipLaunchKernelGGL(smpl_skin_kernel<11>, grid, dim3(256), 0, st, betas, c->pose_feat, c->Amat, c->vt, c->sd, c->pd,
                           c->lbsw, c->reg, N, groups, verts, c->jpart, c->dbg);
    ROMP_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(smpl_joints_kernel, dim3(N), dim3(256), 0, st, verts, c->jpart, c->pick, root_align, joints, c->root);
    ROMP_HIP_CHECK(hipGetLastError());
    if (root_align) {
        hipLaunchKernelGGL(smpl_root_sub_kernel, dim3((NV * 3 / 2 + 255) / 256, N), dim3(256), 0, st, verts, c->root);
        ROMP_HIP_CHECK(hipGetLastError());
    }
    return ROMP_OK;
}

void smpl_ctx_destroy(smpl_ctx* c) {
    if (!c) return;
    float* ptrs[] = {c->vt, c->sd, c->pd, c->lbsw, c->reg, c->Jt, c->Js, c->pose_feat, c->Amat, c->root, c->jpart};
    for (float* p : ptrs)
        if (p) hipFree(p);
    if (c->pick) hipFree(c->pick);
    if (c->sched) hipFree(c->sched);
    delete c;
}

}  // extern "C"
