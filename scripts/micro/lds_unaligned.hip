// lds_unaligned.hip -- may ds_read_b128 / ds_read_b64 take an address that is only 4-byte aligned on this stack (gfx950, ROCm 7.2)?
// A sliding-window gather from a split fp16 halo (csrc/stem7p.hip: a pixel's taps start 12 bytes after its neighbour's) wants 8
// consecutive halves per lane per MFMA step: one ds_read_b128 if unaligned DS access is enabled (SH_MEM_CONFIG.ALIGNMENT_MODE =
// unaligned -- the KFD's setting for gfx9), four ds_read_b32 otherwise.  Checks the VALUES for every 4-byte offset and times both
// forms (reads per clock of one wave, conflict-free strides).
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_unaligned.hip -o scripts/micro/_bin/lds_unaligned
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(unsigned* out, long long* cyc, int off_words, int stride_words, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned s[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = 0x1000u + i;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)s + (threadIdx.x * stride_words + off_words) * 4;
    uint4 v, acc = make_uint4(0, 0, 0, 0);
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    long long t1 = clock64();
    uint4 w;
    for (int i = 0; i < iters; ++i) {
        asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:8\n\tds_read_b32 %3, %4 offset:12\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(w.x), "=v"(w.y), "=v"(w.z), "=v"(w.w) : "v"(addr) : "memory");
        acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
    }
    long long t2 = clock64();
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
    out[256 * 4 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    unsigned* out; long long* cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 16);
    unsigned h[256]; long long c[2];
    const int iters = 2000;
    for (int stride = 3; stride <= 4; ++stride)                  // 3 words = the stem's 12-byte pixel stride, 4 = aligned units
        for (int off = 0; off < 4; ++off) {
            probe<<<1, 64>>>(out, cyc, off, stride, iters);
            if (hipDeviceSynchronize() != hipSuccess) { printf("stride %d off %d: FAULT %s\n", stride, off, hipGetErrorString(hipGetLastError())); return 1; }
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int l = 0; l < 64; ++l) for (int k = 0; k < 4; ++k) bad += h[l * 4 + k] != 0x1000u + l * stride + off + k;
            printf("stride %d words, offset %d words: ds_read_b128 values %s (%d wrong); %.1f clk per b128, %.1f clk per 4 x b32 (one wave, dependent)\n",
                   stride, off, bad ? "WRONG" : "ok", bad, (double)c[0] / iters, (double)c[1] / iters);
        }
    return 0;
}
