// probe_r2.hip -- two hardware questions the f16x2 conv kernels depend on (run on the MI355X box):
//  1. do v_cvt_f16_f32 and v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL values (the low piece h2 of a small activation)?
//  2. does the ds_read_b128 bank model of MI355X_MICROARCH.md (4 x 16-lane service groups) predict the cost of the conv
//     kernels' pixel-fragment reads -- i.e. is the (pixel pad 16, row pad 0) layout at TW=16 really 2-way conflicted and
//     the (pad, row pad) pair chosen by conv_split.h:best_pads conflict-free?
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/probe_r2.hip -o scripts/micro/_bin/probe_r2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void subnormal_kernel(float* out) {
    const int lane = threadIdx.x;
    // A[i][k]: row i = lane&31, k = 8*(lane>>5)+e ; B[k][j]: col j = lane&31.  A = subnormal value at k==0 only, B = 1024 at k==0.
    const float tiny = 3.0e-6f;                         // fp16 subnormal (min normal 6.1e-5), representable as 50 * 2^-24
    const _Float16 ht = (_Float16)tiny;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)0.f; b[e] = (_Float16)0.f; }
    if (lane < 32) { a[0] = ht; b[0] = (_Float16)1024.f; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) {
        out[0] = (float)ht;                              // conversion result (0 if the cvt flushes)
        out[1] = acc[0];                                 // D[0][0] = ht * 1024 (0 if the MFMA flushes subnormal inputs)
        out[2] = tiny;
    }
}

// every lane reads `reads` fragments at base + its pixel offset; returns cycles
__global__ void lds_read_kernel(int psb, int rowb, int tw, int iters, long long* cycles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    for (int i = threadIdx.x; i < 40960; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)i;
    __syncthreads();
    const int off = (li / tw) * rowb + (li % tw) * psb + lh * 16;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(smem + off + (k & 3) * psb + (k >> 2) * rowb);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
}

int main() {
    float* d;
    hipMalloc(&d, 16 * sizeof(float));
    hipLaunchKernelGGL(subnormal_kernel, dim3(1), dim3(64), 0, 0, d);
    float h[3];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("subnormal probe: tiny %.6e  cvt_f16 -> %.6e  mfma(tiny*1024) -> %.6e (expect %.6e)\n", h[2], h[0], h[1], h[0] * 1024.f);
    printf("  cvt keeps subnormals: %s   mfma keeps subnormal inputs: %s\n", h[0] != 0.f ? "YES" : "NO", h[1] != 0.f ? "YES" : "NO");

    long long* dc;
    float* sink;
    hipMalloc(&dc, 8 * sizeof(long long));
    hipMalloc(&sink, 8 * 256 * sizeof(float));
    hipFuncSetAttribute(reinterpret_cast<const void*>(lds_read_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    struct Case { const char* name; int np, ck, tw, ppad, rpad; };
    const Case cases[] = {
        {"bf16x3 CK16 TW16 pad16 row0 (r01 layout)", 3, 16, 16, 16, 0},
        {"bf16x3 CK16 TW16 pad0  row16 (best_pads)", 3, 16, 16, 0, 16},
        {"f16x2  CK16 TW16 pad16 row0", 2, 16, 16, 16, 0},
        {"f16x2  CK16 TW16 pad16 row96 (best_pads)", 2, 16, 16, 16, 96},
        {"f16x2  CK16 TW16 pad0  row0 (dense)", 2, 16, 16, 0, 0},
        {"bf16x3 CK16 TW32 pad16 row0", 3, 16, 32, 16, 0},
        {"f16x2  CK16 TW32 pad16 row0", 2, 16, 32, 16, 0},
        {"f16x2  CK32 TW16 pad16 row224 (best_pads)", 2, 32, 16, 16, 224},
        {"f16x2  CK32 TW16 pad16 row0", 2, 32, 16, 16, 0},
    };
    for (const Case& c : cases) {
        const int psb = c.np * c.ck * 2 + c.ppad, hc = c.tw + 2, rowb = hc * psb + c.rpad;
        for (int waves = 4; waves <= 8; waves += 4) {
            hipLaunchKernelGGL(lds_read_kernel, dim3(1), dim3(64 * waves), 163840, 0, psb, rowb, c.tw, 2000, dc, sink);
            hipDeviceSynchronize();
            long long cyc;
            hipMemcpy(&cyc, dc, sizeof(cyc), hipMemcpyDeviceToHost);
            printf("%-46s waves %d: %.2f cycles per wave-level ds_read_b128 (CU total), %.2f per read per wave\n", c.name, waves,
                   (double)cyc / (2000.0 * 16 * waves), (double)cyc / (2000.0 * 16));
        }
    }
    return 0;
}
