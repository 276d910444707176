// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate (a) registers only, (b) with LDS fragment reads.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ int g_random = 0;
template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float s[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) {
        unsigned h = (i + blockIdx.x * 8192) * 2654435761u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        s[i] = g_random ? ((int)(h & 0xffffff) - 0x800000) * (1.0f / 0x800000) : i * 1e-6f;
    }
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float4 x = make_float4(1.f, 2.f, 3.f, 4.f), w = make_float4(0.5f, 0.25f, 0.125f, 1.f);
    const int off = (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            x = *reinterpret_cast<const float4*>(s + off + (it & 15) * 256);
            w = *reinterpret_cast<const float4*>(s + 4096 + off + (it & 15) * 256);
        }
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, x.x, acc[a], 0, 0, 0);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, x.y, acc[a], 0, 0, 0);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, x.z, acc[a], 0, 0, 0);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, x.w, acc[a], 0, 0, 0);
        }
    }
    float t = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) t += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int NACC, bool LDS>
void run(const char* name, int blocks_per_cu) {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 4000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(grid), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 /*waves*/ * iters * NACC * 4 * 4096.0;
    printf("%-28s blocks/CU %d: %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
  for (int rnd = 0; rnd < 2; ++rnd) {
    hipMemcpyToSymbol(HIP_SYMBOL(g_random), &rnd, sizeof(int));
    printf("---- LDS data: %s\n", rnd ? "random [-1,1)" : "smooth tiny values");
    run<4, false>("regs only, 4 acc", 1);
    run<4, false>("regs only, 4 acc", 2);
    run<2, false>("regs only, 2 acc", 1);
    run<2, false>("regs only, 2 acc", 2);
    run<1, false>("regs only, 1 acc", 2);
    run<2, true>("LDS frags, 2 acc", 1);
    run<2, true>("LDS frags, 2 acc", 2);
    run<2, true>("LDS frags, 2 acc", 3);
    run<4, true>("LDS frags, 4 acc", 2);
  }
    return 0;
}
