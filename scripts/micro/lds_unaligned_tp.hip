// lds_unaligned_tp.hip -- THROUGHPUT of 16 bytes per lane out of LDS at a 4-byte-aligned address (stride 12 bytes between lanes: the
// stem's sliding window), as one ds_read_b128, two ds_read_b64 or four ds_read_b32; 8 waves per CU, 8 independent reads in flight each.
// (lds_unaligned.hip showed the values are right in every form; csrc/stem7p.hip with unaligned b128 reads ran 3 x slower than with b32.)
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_unaligned_tp.hip -o scripts/micro/_bin/lds_unaligned_tp
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int FORM>
__global__ __launch_bounds__(512) void tp(unsigned* out, long long* cyc, int off_words, int stride_words, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned s[12288];
    for (int i = threadIdx.x; i < 12288; i += blockDim.x) s[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)s + (wave * 1024 + lane * stride_words + off_words) * 4;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 v[8], acc = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (FORM == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[k]) : "v"(addr), "n"(0) : "memory");
            if (FORM == 1) asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8" : "=&v"(*(unsigned long long*)&v[k]), "=&v"(*((unsigned long long*)&v[k] + 1)) : "v"(addr) : "memory");
            if (FORM == 2) asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:8\n\tds_read_b32 %3, %4 offset:12"
                                        : "=&v"(v[k].x), "=&v"(v[k].y), "=&v"(v[k].z), "=&v"(v[k].w) : "v"(addr) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    unsigned* out; long long* cyc; long long c;
    (void)hipMalloc(&out, 4096 * 4); (void)hipMalloc(&cyc, 16);
    const int iters = 500;
    const char* names[3] = {"1 x b128", "2 x b64", "4 x b32"};
    for (int stride = 3; stride <= 4; ++stride)
        for (int off = 0; off < 2; ++off)
            for (int form = 0; form < 3; ++form) {
                if (form == 0) tp<0><<<1, 512>>>(out, cyc, off, stride, iters);
                if (form == 1) tp<1><<<1, 512>>>(out, cyc, off, stride, iters);
                if (form == 2) tp<2><<<1, 512>>>(out, cyc, off, stride, iters);
                if (hipDeviceSynchronize() != hipSuccess) { printf("FAULT\n"); return 1; }
                (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
                printf("lane stride %2d B, offset %d B, %s: %.1f clk per 16 bytes x 64 lanes per wave (8 waves: %.1f B/clk/CU)\n", stride * 4, off * 4, names[form],
                       (double)c / (iters * 8), 8.0 * 1024.0 * iters * 8 / (double)c);
            }
    return 0;
}
