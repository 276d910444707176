// ingest_bench.hip -- how many bytes per cycle can ONE CU pull in, and what does it depend on?
// Every conv kernel variant of round 2 ends up moving ~10-14 B/clk/CU through its global-load / LDS-DMA path regardless of
// whether the bytes come from HBM or from L2 (weights shared by all CUs).  This probe measures the ceiling directly:
// grid = one workgroup per CU (256 WGs), W waves each; every wave streams `bytes_per_wave` with `INFL` 16-byte loads per lane in
// flight, either into registers (global_load_dwordx4) or straight into LDS (global_load_lds_dwordx4);
// source = (a) one 144 KiB block shared by every CU (L2-resident, the "weights" case), (b) a distinct 256 KiB block per CU,
// re-read (L2 / MALL), (c) a distinct 8 MiB block per CU, streamed once (HBM).
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/ingest_bench.hip -o scripts/micro/_bin/ingest_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int INFL, bool DMA>
__global__ __launch_bounds__(1024) void ingest_kernel(const uint4* __restrict__ src, size_t cu_stride_u4, int block_u4, int iters,
                                                      float* sink, long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint4* base = src + (size_t)blockIdx.x * cu_stride_u4;
    uint4 acc = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const long long t0 = clock64();
    // the workgroup sweeps its block `iters` times; a wave's k-th instruction covers units (k * nw + wave) * 64 + lane
    int idx = wave * 64 + lane;
    const int step = nw * 64;
    for (int it = 0; it < iters; ++it) {
        for (int u = idx; u < block_u4; u += step * INFL) {
            if (DMA) {
#pragma unroll
                for (int k = 0; k < INFL; ++k) {
                    const int uu = (u + k * step) % block_u4;
                    __builtin_amdgcn_global_load_lds((glb_void_t*)(base + uu), (lds_void_t*)(smem + ((wave * INFL + k) & 63) * 1024), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                uint4 v[INFL];
#pragma unroll
                for (int k = 0; k < INFL; ++k) v[k] = base[(u + k * step) % block_u4];
#pragma unroll
                for (int k = 0; k < INFL; ++k) { acc.x ^= v[k].x; acc.y += v[k].y; acc.z ^= v[k].z; acc.w += v[k].w; }
            }
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (!DMA) sink[blockIdx.x * blockDim.x + threadIdx.x] = (float)(acc.x + acc.y + acc.z + acc.w);
    else if (threadIdx.x == 0) sink[blockIdx.x] = reinterpret_cast<float*>(smem)[lane];
}

template <int INFL, bool DMA>
static void run(const char* what, const uint4* src, size_t cu_stride_u4, int block_u4, int iters, int waves, float* sink, long long* dcyc) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(ingest_kernel<INFL, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((ingest_kernel<INFL, DMA>), dim3(256), dim3(64 * waves), 65536, 0, src, cu_stride_u4, block_u4, iters, sink, dcyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(256);
    hipMemcpy(c.data(), dcyc, 256 * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (long long v : c) mean += (double)v / 256.0;
    const double bytes_cu = (double)block_u4 * 16.0 * iters;
    printf("%-34s %s waves %2d infl %2d: %7.1f us  %6.2f B/clk/CU (clock64)  %7.1f GB/s/CU  %6.2f TB/s chip\n", what, DMA ? "dma " : "regs", waves, INFL,
           ms * 1e3, bytes_cu / mean, bytes_cu / (ms * 1e-3) / 1e9, bytes_cu * 256 / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t total = (size_t)256 * 8 * 1024 * 1024;           // 2 GiB
    uint4* buf;
    hipMalloc(&buf, total);
    hipMemset(buf, 1, total);
    float* sink; long long* dcyc;
    hipMalloc(&sink, 256 * 1024 * sizeof(float));
    hipMalloc(&dcyc, 256 * sizeof(long long));
    struct Src { const char* name; size_t stride_u4; int block_u4; int iters; };
    const Src srcs[] = {
        {"shared 144 KiB (L2 hit, weights)", 0, 144 * 1024 / 16, 32},
        {"per-CU 256 KiB re-read (L2/MALL)", 8 * 1024 * 1024 / 16, 256 * 1024 / 16, 16},
        {"per-CU 8 MiB streamed (HBM)", 8 * 1024 * 1024 / 16, 8 * 1024 * 1024 / 16, 1},
    };
    for (const Src& s : srcs) {
        for (int waves : {4, 8, 16}) {
            run<4, false>(s.name, buf, s.stride_u4, s.block_u4, s.iters, waves, sink, dcyc);
            run<8, false>(s.name, buf, s.stride_u4, s.block_u4, s.iters, waves, sink, dcyc);
            run<4, true>(s.name, buf, s.stride_u4, s.block_u4, s.iters, waves, sink, dcyc);
            run<8, true>(s.name, buf, s.stride_u4, s.block_u4, s.iters, waves, sink, dcyc);
            run<16, true>(s.name, buf, s.stride_u4, s.block_u4, s.iters, waves, sink, dcyc);
        }
    }
    return 0;
}
