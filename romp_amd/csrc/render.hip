// render.hip -- Sim3DR mesh renderer on the device (the only native component next to
// the hot path; `--render_mesh` pays ~10x the network time for it on the host in the reference).
//
// Reference: simple_romp/vis_human/sim3drender/lib/rasterize_kernel.cpp  _get_normal :171-229,
// get_point_weight :56-85, _rasterize :233-300; simple_romp/vis_human/sim3drender/renderer.py
// Sim3DR.render :64-118 (lighting), __call__ :120-133.
//
// The reference is sequential (triangles in index order, strict `>` z-test, colour written on every
// accepted fragment).  With alpha == 1 (the only value its Python passes) the final colour of a pixel is
// the colour of the fragment with the greatest depth, ties going to the LOWEST triangle index -- an
// order-free statement, so the device runs two passes: (1) one thread per triangle scans its bounding box
// and atomicMax-es a 64-bit key {orderable depth, ~triangle} per pixel; (2) one thread per pixel recomputes
// the winner's barycentric weights and writes the truncated uint8 colour.  Every float expression keeps
// the reference's operation order with contraction off, so the image is BIT-IDENTICAL to the C++ one.
// Vertex normals: the reference accumulates face normals onto vertices in triangle order; here one thread
// per vertex walks its (triangle, corner) incidence list in ascending order -- the same additions in the
// same order.  Lighting: one workgroup per mesh (min / max reductions of norm_vertices in LDS).
#include "common.h"

#pragma clang fp contract(off)

namespace romp {

__global__ void sim3dr_normal_kernel(const float* __restrict__ v, const int32_t* __restrict__ tri,
                                     const int32_t* __restrict__ adj_off, const int32_t* __restrict__ adj_ent, int nver,
                                     float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nver) return;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    for (int e = adj_off[i]; e < adj_off[i + 1]; ++e) {
        const int t = adj_ent[e] / 3;
        const int a = tri[3 * t], b = tri[3 * t + 1], c = tri[3 * t + 2];
        const float v1x = v[3 * b] - v[3 * a], v1y = v[3 * b + 1] - v[3 * a + 1], v1z = v[3 * b + 2] - v[3 * a + 2];
        const float v2x = v[3 * c] - v[3 * a], v2y = v[3 * c + 1] - v[3 * a + 1], v2z = v[3 * c + 2] - v[3 * a + 2];
        nx += v1y * v2z - v1z * v2y;
        ny += v1z * v2x - v1x * v2z;
        nz += v1x * v2y - v1y * v2x;
    }
    float det = sqrtf(nx * nx + ny * ny + nz * nz);
    if (det <= 0.f) det = 1e-6f;
    out[3 * i] = nx / det; out[3 * i + 1] = ny / det; out[3 * i + 2] = nz / det;
}

struct LightCfg {
    float ambient[3];          // intensity_ambient * color, already rounded to float32 as numpy does (renderer.py:83)
    float i_dir, i_spec;       // 0: term switched off
    float color_dir[3], light_pos[3], view_pos[3];
};

__device__ __forceinline__ float clip01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

// renderer.py:19-24 (norm_vertices) + :77-110
__global__ __launch_bounds__(1024) void sim3dr_light_kernel(const float* __restrict__ v, const float* __restrict__ nrm, int nver,
                                                             LightCfg cfg, float* __restrict__ light) {
    __shared__ float red[3][1024];
    __shared__ float s_min[3], s_max1, s_max3[3];
    const int tid = threadIdx.x;
    float m[3] = {3.4e38f, 3.4e38f, 3.4e38f};
    for (int i = tid; i < nver; i += 1024)
        for (int k = 0; k < 3; ++k) m[k] = fminf(m[k], v[3 * i + k]);
    for (int k = 0; k < 3; ++k) red[k][tid] = m[k];
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) for (int k = 0; k < 3; ++k) red[k][tid] = fminf(red[k][tid], red[k][tid + s]);
        __syncthreads();
    }
    if (tid < 3) s_min[tid] = red[tid][0];
    __syncthreads();
    float mx = -3.4e38f;                                          // vertices.max() after the shift
    for (int i = tid; i < nver; i += 1024)
        for (int k = 0; k < 3; ++k) mx = fmaxf(mx, v[3 * i + k] - s_min[k]);
    red[0][tid] = mx;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) red[0][tid] = fmaxf(red[0][tid], red[0][tid + s]);
        __syncthreads();
    }
    if (tid == 0) s_max1 = red[0][0];
    __syncthreads();
    float m3[3] = {-3.4e38f, -3.4e38f, -3.4e38f};                 // vertices.max(0) after /max, *2
    for (int i = tid; i < nver; i += 1024)
        for (int k = 0; k < 3; ++k) m3[k] = fmaxf(m3[k], ((v[3 * i + k] - s_min[k]) / s_max1) * 2.f);
    __syncthreads();
    for (int k = 0; k < 3; ++k) red[k][tid] = m3[k];
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) for (int k = 0; k < 3; ++k) red[k][tid] = fmaxf(red[k][tid], red[k][tid + s]);
        __syncthreads();
    }
    if (tid < 3) s_max3[tid] = red[tid][0] / 2.f;
    __syncthreads();
    for (int i = tid; i < nver; i += 1024) {
        float vn[3], n[3], l[3];
        for (int k = 0; k < 3; ++k) {
            vn[k] = ((v[3 * i + k] - s_min[k]) / s_max1) * 2.f - s_max3[k];
            n[k] = nrm[3 * i + k];
            l[k] = cfg.ambient[k];
        }
        if (cfg.i_dir > 0.f) {
            float d[3];
            for (int k = 0; k < 3; ++k) d[k] = cfg.light_pos[k] - vn[k];
            const float dl = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            for (int k = 0; k < 3; ++k) d[k] = d[k] / dl;
            const float cs = n[0] * d[0] + n[1] * d[1] + n[2] * d[2];
            const float cc = clip01(cs);
            for (int k = 0; k < 3; ++k) l[k] += cfg.i_dir * (cfg.color_dir[k] * cc);
            if (cfg.i_spec > 0.f) {
                float e[3];
                for (int k = 0; k < 3; ++k) e[k] = cfg.view_pos[k] - vn[k];
                const float el = sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
                float spe = 0.f;
                for (int k = 0; k < 3; ++k) {
                    const float r = (2.f * cs) * n[k] - d[k];
                    const float t = (e[k] / el) * r;
                    spe = k == 0 ? t : spe + t;
                }
                spe = cs != 0.f ? clip01(spe) : 0.f;
                for (int k = 0; k < 3; ++k) l[k] += (cfg.i_spec * cfg.color_dir[k]) * clip01(spe);
            }
        }
        for (int k = 0; k < 3; ++k) light[3 * i + k] = clip01(l[k]);
    }
}

// rasterize_kernel.cpp:56-85
__device__ __forceinline__ void point_weight(float px, float py, float p0x, float p0y, float p1x, float p1y, float p2x, float p2y,
                                             float& w0, float& w1, float& w2) {
    const float v0x = p2x - p0x, v0y = p2y - p0y, v1x = p1x - p0x, v1y = p1y - p0y, v2x = px - p0x, v2y = py - p0y;
    const float d00 = v0x * v0x + v0y * v0y, d01 = v0x * v1x + v0y * v1y, d02 = v0x * v2x + v0y * v2y;
    const float d11 = v1x * v1x + v1y * v1y, d12 = v1x * v2x + v1y * v2y;
    const float den = d00 * d11 - d01 * d01;
    const float inv = den == 0.f ? 0.f : 1.f / den;
    const float u = (d11 * d02 - d01 * d12) * inv;
    const float vv = (d00 * d12 - d01 * d02) * inv;
    w0 = 1.f - u - vv; w1 = vv; w2 = u;
}

__device__ __forceinline__ unsigned orderable(float d) {
    const unsigned u = __float_as_uint(d);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void sim3dr_raster_kernel(const float* __restrict__ v, const int32_t* __restrict__ tri, int ntri, int h, int w,
                                     unsigned long long* __restrict__ keys) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntri) return;
    const int a = tri[3 * t], b = tri[3 * t + 1], c = tri[3 * t + 2];
    const float p0x = v[3 * a], p0y = v[3 * a + 1], z0 = v[3 * a + 2];
    const float p1x = v[3 * b], p1y = v[3 * b + 1], z1 = v[3 * b + 2];
    const float p2x = v[3 * c], p2y = v[3 * c + 1], z2 = v[3 * c + 2];
    const int x_min = max((int)ceilf(fminf(p0x, fminf(p1x, p2x))), 0);
    const int x_max = min((int)floorf(fmaxf(p0x, fmaxf(p1x, p2x))), w - 1);
    const int y_min = max((int)ceilf(fminf(p0y, fminf(p1y, p2y))), 0);
    const int y_max = min((int)floorf(fmaxf(p0y, fmaxf(p1y, p2y))), h - 1);
    if (x_max < x_min || y_max < y_min) return;
    const unsigned long long low = 0xFFFFFFFFull - (unsigned)t;          // ties in depth: lowest triangle index wins
    for (int y = y_min; y <= y_max; ++y)
        for (int x = x_min; x <= x_max; ++x) {
            float w0, w1, w2;
            point_weight((float)x, (float)y, p0x, p0y, p1x, p1y, p2x, p2y, w0, w1, w2);
            if (w2 >= 0.f && w1 >= 0.f && w0 > 0.f) {
                const float d = w0 * z0 + w1 * z1 + w2 * z2;
                if (d > -1e8f) atomicMax(&keys[(size_t)y * w + x], ((unsigned long long)orderable(d) << 32) | low);
            }
        }
}

__global__ void sim3dr_resolve_kernel(const float* __restrict__ v, const int32_t* __restrict__ tri, const float* __restrict__ col,
                                      const unsigned long long* __restrict__ keys, int h, int w, int c, int reverse,
                                      unsigned char* __restrict__ image) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h * w) return;
    const unsigned long long key = keys[i];
    if (key == 0ull) return;
    const int t = (int)(0xFFFFFFFFull - (key & 0xFFFFFFFFull));
    const int x = i % w, y = i / w;
    const int a = tri[3 * t], b = tri[3 * t + 1], cc = tri[3 * t + 2];
    float w0, w1, w2;
    point_weight((float)x, (float)y, v[3 * a], v[3 * a + 1], v[3 * b], v[3 * b + 1], v[3 * cc], v[3 * cc + 1], w0, w1, w2);
    const int row = reverse ? (h - 1 - y) : y;
    const float alpha = 1.f;
    for (int k = 0; k < c; ++k) {
        const float pc = w0 * col[c * a + k] + w1 * col[c * b + k] + w2 * col[c * cc + k];
        unsigned char* px = image + ((size_t)row * w + x) * c + k;
        *px = (unsigned char)((1 - alpha) * (*px) + alpha * 255 * pc);      // rasterize_kernel.cpp:287-288
    }
}

}  // namespace romp

using namespace romp;

extern "C" {

int romp_sim3dr_normals(const float* verts, const int32_t* tris, const int32_t* adj_off, const int32_t* adj_ent, int nver,
                        float* normals, void* stream) {
    ROMP_REQUIRE(verts && tris && adj_off && adj_ent && normals && nver > 0, "romp_sim3dr_normals: bad arguments");
    hipLaunchKernelGGL(sim3dr_normal_kernel, dim3((nver + 255) / 256), dim3(256), 0, (hipStream_t)stream, verts, tris, adj_off,
                       adj_ent, nver, normals);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

int romp_sim3dr_light(const float* verts, const float* normals, int nver, const float* cfg_host, float* light, void* stream) {
    ROMP_REQUIRE(verts && normals && cfg_host && light && nver > 0, "romp_sim3dr_light: bad arguments");
    LightCfg c;
    for (int k = 0; k < 3; ++k) {
        c.ambient[k] = cfg_host[k]; c.color_dir[k] = cfg_host[5 + k]; c.light_pos[k] = cfg_host[8 + k]; c.view_pos[k] = cfg_host[11 + k];
    }
    c.i_dir = cfg_host[3]; c.i_spec = cfg_host[4];
    hipLaunchKernelGGL(sim3dr_light_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, verts, normals, nver, c, light);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

int romp_sim3dr_rasterize(unsigned char* image, const float* verts, const int32_t* tris, const float* colors, int ntri, int h,
                          int w, int c, int reverse, unsigned long long* keys, void* stream) {
    ROMP_REQUIRE(image && verts && tris && colors && keys && ntri > 0 && h > 0 && w > 0 && c > 0 && c <= 4,
                 "romp_sim3dr_rasterize: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    ROMP_HIP_CHECK(hipMemsetAsync(keys, 0, (size_t)h * w * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(sim3dr_raster_kernel, dim3((ntri + 63) / 64), dim3(64), 0, st, verts, tris, ntri, h, w, keys);
    hipLaunchKernelGGL(sim3dr_resolve_kernel, dim3((h * w + 255) / 256), dim3(256), 0, st, verts, tris, colors, keys, h, w, c,
                       reverse, image);
    ROMP_HIP_CHECK(hipGetLastError());
    return ROMP_OK;
}

}  // extern "C"
