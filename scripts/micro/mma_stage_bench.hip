// Microbenchmark of mma_stage (the MFMA + LDS-fragment loop of conv_mfma.hip) in isolation.
#include "../../romp_amd/csrc/conv_mfma.hip"
#include <stdio.h>
using namespace romp;

template <int KS, int S, int MT, int NT, int TW, int CK>
__global__ __launch_bounds__(256) void stage_only(float* out, int iters) {
    using C = ConvCfg<KS, S, MT, NT, TW, CK>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + C::HR * C::HC * C::PS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < C::LDS_BYTES / 4; i += 256) {
        unsigned h = (i + blockIdx.x * 8192) * 2654435761u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        smem[i] = ((int)(h & 0xffffff) - 0x800000) * (1.0f / 0x800000);
    }
    __syncthreads();
    int xoff[MT];
    for (int m = 0; m < MT; ++m) {
        const int mb = wave * MT + m;
        const int row = mb * C::RPB + li / TW, col = li % TW;
        xoff[m] = ((row * S) * C::HC + col * S) * C::PS + lh * 4;
    }
    const int woff = (lh * C::NW + li) * 4;
    f32x16 acc[MT][NT];
    for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    for (int it = 0; it < iters; ++it) mma_stage<KS, S, MT, NT, TW, CK>(sA, sB, xoff, woff, acc);
    float t = 0.f;
    for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) t += acc[m][n][r];
    out[blockIdx.x * 256 + tid] = t;
}

template <int KS, int S, int MT, int NT, int TW, int CK>
void run(int blocks_per_cu) {
    using C = ConvCfg<KS, S, MT, NT, TW, CK>;
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    auto fn = stage_only<KS, S, MT, NT, TW, CK>;
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    const int iters = 40, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), C::LDS_BYTES, 0, out, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), C::LDS_BYTES, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * (double)(C::TAPS * (CK / 8) * 4 * MT * NT) * 4096.0;
    printf("k%d s%d mt%d nt%d tw%d ck%d  blocks/CU %d (lds %d KB): %.3f ms  %.1f TFLOP/s  err=%s\n", KS, S, MT, NT, TW, CK,
           blocks_per_cu, C::LDS_BYTES / 1024, ms, flops / ms / 1e9, hipGetErrorString(hipGetLastError()));
    hipFree(out);
}

int main() {
    run<3, 1, 2, 2, 16, 16>(1); run<3, 1, 2, 2, 16, 16>(2);
    run<3, 1, 2, 1, 16, 16>(1); run<3, 1, 2, 1, 16, 16>(2); run<3, 1, 2, 1, 16, 16>(3);
    run<3, 1, 2, 2, 32, 16>(1); run<3, 1, 2, 2, 32, 16>(2);
    run<3, 1, 2, 1, 32, 16>(1); run<3, 1, 2, 1, 32, 16>(2); run<3, 1, 2, 1, 32, 16>(3);
    run<3, 1, 1, 1, 32, 16>(2); run<3, 1, 1, 1, 32, 16>(4);
    run<1, 1, 2, 2, 32, 32>(2);
    return 0;
}
