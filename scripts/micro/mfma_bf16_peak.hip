// Microbenchmark: sustained v_mfma_f32_32x32x16_bf16 rate on random data, in the shapes the bf16x3
// conv kernels use: (a) registers only, (b) the bx3 step -- MT*3 + NT*3 ds_read_b128 fragment reads
// feeding 6*MT*NT MFMAs -- at 1 and 2 workgroups per CU.  Gives the practical ceiling (clock under
// matrix load included) that roofline fractions of the conv kernels should be read against.
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_bf16_peak.hip -o /tmp/mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MT, int NT, bool LDS>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char s[];      // 64 KB: 4096 units of 16 B
    for (int i = threadIdx.x; i < 16384; i += 256) {
        unsigned h = (i + blockIdx.x * 16384) * 2654435761u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        const float a = ((int)(h & 0xffff) - 0x8000) * (1.0f / 0x8000), b = ((int)(h >> 16) - 0x8000) * (1.0f / 0x8000);
        const __bf16 ba = (__bf16)a, bb = (__bf16)b;
        reinterpret_cast<unsigned*>(s)[i] = (unsigned)__builtin_bit_cast(unsigned short, ba) | ((unsigned)__builtin_bit_cast(unsigned short, bb) << 16);
    }
    __syncthreads();
    f32x16 acc[MT][NT];
    for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    bf16x8 xf[MT][3], wf[NT][3];
    const int lane = threadIdx.x & 63;
    for (int m = 0; m < MT; ++m) for (int pc = 0; pc < 3; ++pc) xf[m][pc] = *reinterpret_cast<const bf16x8*>(s + ((m * 3 + pc) * 64 + lane) * 16);
    for (int n = 0; n < NT; ++n) for (int pc = 0; pc < 3; ++pc) wf[n][pc] = *reinterpret_cast<const bf16x8*>(s + ((16 + n * 3 + pc) * 64 + lane) * 16);
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            const char* base = s + (it & 1) * 32768 + lane * 16;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) xf[m][pc] = *reinterpret_cast<const bf16x8*>(base + (m * 3 + pc) * 1024);
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) wf[n][pc] = *reinterpret_cast<const bf16x8*>(base + (12 + n * 3 + pc) * 1024);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][2], xf[m][0], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][1], xf[m][1], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][0], xf[m][2], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][1], xf[m][0], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][0], xf[m][1], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][0], xf[m][0], acc[m][n], 0, 0, 0);
            }
    }
    float t = 0.f;
    for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) t += acc[m][n][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int MT, int NT, bool LDS>
void run(const char* name, int blocks_per_cu) {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 3000, grid = 256 * blocks_per_cu;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MT, NT, LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MT, NT, LDS>), dim3(grid), dim3(256), 65536, 0, out, 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MT, NT, LDS>), dim3(grid), dim3(256), 65536, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)grid * 4 * iters * MT * NT * 6 * (2.0 * 32 * 32 * 16);
    printf("%-34s WG/CU %d: %.3f ms  %7.1f TFLOP/s bf16 issued = %6.1f TFLOP/s f32-equivalent\n", name, blocks_per_cu, best,
           flops / best / 1e9, flops / best / 1e9 / 6);
    hipFree(out);
}

int main() {
    run<2, 2, false>("regs only mt2 nt2", 1);
    run<2, 2, false>("regs only mt2 nt2", 2);
    run<1, 2, false>("regs only mt1 nt2", 2);
    run<2, 2, true>("LDS frags mt2 nt2 (12 rd/24 mfma)", 1);
    run<2, 2, true>("LDS frags mt2 nt2 (12 rd/24 mfma)", 2);
    run<1, 2, true>("LDS frags mt1 nt2 (9 rd/12 mfma)", 2);
    run<2, 1, true>("LDS frags mt2 nt1 (9 rd/12 mfma)", 2);
    run<1, 1, true>("LDS frags mt1 nt1 (6 rd/6 mfma)", 2);
    return 0;
}
