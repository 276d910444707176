// wino_step_bench.hip -- the Step-B gate of VERDICT r04 item 2: before a Winograd F(2x2,3x3) conv kernel is written, measure what its
// INNER STEP costs on this chip next to the direct f16x2 step it would replace, in the same setting conv_h2r.hip runs in (256-thread
// workgroups, two per CU = two waves per SIMD, pixels in LDS, weight fragments straight from L2 into registers).
//
// Work unit of both steps: 16 input channels x (128 output pixels x 64 output channels) per workgroup.
//   direct   (conv_h2r's sub-stage at P = 2, NS = 2): per wave 64 px x 32 couts: 36 ds_read_b128 pixel fragments, 18 weight
//            fragments (global -> registers), 54 MFMAs 32x32x16 (9 taps x 2 blocks x 3 piece products), 2 accumulator sets.
//   winograd (the cheapest decomposition that fits the register file at two waves per SIMD, see profiles/r05_notes.md): wave w owns
//            the transformed ROW r' = w of the 4x4 tile for 32 tiles (= 128 output pixels) x 64 couts: 16 ds_read_b128 raw H2 units
//            (2 input rows x 4 columns x 2 pieces), decode h1 + h2 -> f32 (64 v_fma_mix), row transform (16 v_pk_add_f32), column
//            transform (16 v_pk_add_f32), re-split of the 32 transformed values into fp16 pieces (16 v_cvt_pk + 32 v_fma_mix +
//            16 v_cvt_pk) = 160 VALU instructions, 16 weight fragments, 24 MFMAs (4 positions x 2 cout blocks x 3), 8 accumulator
//            sets.  (The output transform and its cross-wave reduction are NOT in the step: they belong to the epilogue.)
// Reported: cycles per step and wave (s_memtime is constant-rate, so wall time from HIP events / steps), the MFMA-only floor of each
// step, and the ratio direct / winograd -- 2.25 would be the arithmetic ideal.
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/wino_step_bench.hip -o scripts/micro/_bin/wino_step_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { return a + b; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { return a - b; }

// (h1.lo + h2.lo, h1.hi + h2.hi) as float32: two v_fma_mix_f32
__device__ __forceinline__ f32x2 decode_pair(unsigned h1, unsigned h2) {
    f32x2 r;
    asm("v_fma_mix_f32 %0, %2, 1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_fma_mix_f32 %1, %2, 1.0, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]"
        : "=&v"(r.x), "=&v"(r.y) : "v"(h1), "v"(h2));
    return r;
}
// split (a, b) into packed high pieces and packed low pieces: cvt_pk, 2 fma_mix, cvt_pk
__device__ __forceinline__ void split_pair(f32x2 v, unsigned& hi, unsigned& lo) {
    const f16x2 h = __builtin_convertvector(v, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    float ta, tb;
    asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(ta), "=&v"(tb) : "v"(hi), "v"(v.x), "v"(v.y));
    const f32x2 t = {ta, tb};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(t, f16x2));
}

// a 16-byte register value the compiler knows nothing about (stands in for a load that was knocked out)
__device__ __forceinline__ uint4 opaque() {
    uint4 t = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    asm volatile("" : "+v"(t.x), "+v"(t.y), "+v"(t.z), "+v"(t.w));
    return t;
}
// every bit of a fragment matters to the result (knock-out runs without MFMAs): 4 integer ops
__device__ __forceinline__ float consume(f16x8 v) {
    const uint4 t = __builtin_bit_cast(uint4, v);
    return __builtin_bit_cast(float, ((t.x ^ t.y) ^ (t.z ^ t.w)) & 0x3fffffffu);
}

// MODE 0: direct step, 1: winograd step; WHAT bit 0: MFMAs, bit 1: VALU transform (winograd), bit 2: LDS fragment reads, bit 3: weight loads
template <int MODE, int WHAT>
__global__ __launch_bounds__(256, 2) void step_kernel(const uint4* __restrict__ wts, int w_units, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 32 KB of "pixels"
    for (int i = threadIdx.x; i < 8192; i += 256) {
        unsigned h = (i + blockIdx.x * 8192) * 2654435761u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        // fp16 pairs in a sane range: exponent field 13..16
        reinterpret_cast<unsigned*>(smem)[i] = (h & 0x83ff83ffu) | 0x34003400u | ((h >> 3) & 0x04000400u);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NACC = MODE ? 8 : 2;
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const uint4* wp = wts + lane;
    unsigned woff = (unsigned)((blockIdx.x * 4 + wave) * 1024) & (unsigned)(w_units - 1);
    uint4 wq[2][2];                                              // winograd: the current position's weight fragments
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) wq[cb][pc] = wp[(woff + (cb * 2 + pc) * 64) & (unsigned)(w_units - 1)];
    for (int s = 0; s < steps; ++s) {
        const char* base = smem + ((s & 3) * 4096) + (lane & 31) * 16 + (lane >> 5) * 2048;
        if (MODE == 0) {
            // ---- direct: 9 taps x (2 weight pieces, 2 blocks x 2 pixel pieces, 6 MFMAs)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                f16x8 w[2], x[2][2];
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    uint4 t = (WHAT & 8) ? wp[(woff + (tap * 2 + pc) * 64) & (unsigned)(w_units - 1)] : opaque();
                    w[pc] = __builtin_bit_cast(f16x8, t);
                }
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        uint4 t = (WHAT & 4) ? *reinterpret_cast<const uint4*>(base + ((tap * 4 + g * 2 + pc) * 512) % 16384) : opaque();
                        x[g][pc] = __builtin_bit_cast(f16x8, t);
                    }
                if (WHAT & 1) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1], x[g][0], acc[g], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], x[g][1], acc[g], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], x[g][0], acc[g], 0, 0, 0);
                } else {
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[g][tap] += consume(w[0]) + consume(w[1]) + consume(x[g][0]) + consume(x[g][1]);
                }
            }
            woff = (woff + 18 * 64) & (unsigned)(w_units - 1);
        } else {
            // ---- winograd: raw units of 2 input rows x 4 columns (h1, h2), this lane's channel octet of its tile; column by
            // column: 4 reads, decode + row transform r' = 1 (d1 + d2) -> 8 floats (a real kernel would stream them like this too)
            f32x2 row[4][4];                                 // [column][channel pair]
            unsigned vhi[4][4], vlo[4][4];                   // [position c'][channel pair]: the B fragments (8 halves = 4 dwords) per piece
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint4 raw[2][2];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc)
                        raw[r][pc] = (WHAT & 4) ? *reinterpret_cast<const uint4*>(base + (((r * 4 + c) * 2 + pc) * 512) % 16384) : opaque();
#pragma unroll
                for (int cp = 0; cp < 4; ++cp) {
                    const unsigned a1 = reinterpret_cast<const unsigned*>(&raw[0][0])[cp], a2 = reinterpret_cast<const unsigned*>(&raw[0][1])[cp];
                    const unsigned b1 = reinterpret_cast<const unsigned*>(&raw[1][0])[cp], b2 = reinterpret_cast<const unsigned*>(&raw[1][1])[cp];
                    if (WHAT & 2) row[c][cp] = pk_add(decode_pair(a1, a2), decode_pair(b1, b2));
                    else { row[c][cp].x = __builtin_bit_cast(float, a1 ^ b2); row[c][cp].y = __builtin_bit_cast(float, a2 ^ b1); }
                }
            }
#pragma unroll
            for (int cp = 0; cp < 4; ++cp) {
                if (WHAT & 2) {
                    const f32x2 v0 = pk_sub(row[0][cp], row[2][cp]), v1 = pk_add(row[1][cp], row[2][cp]);
                    const f32x2 v2 = pk_sub(row[2][cp], row[1][cp]), v3 = pk_sub(row[1][cp], row[3][cp]);
                    split_pair(v0, vhi[0][cp], vlo[0][cp]); split_pair(v1, vhi[1][cp], vlo[1][cp]);
                    split_pair(v2, vhi[2][cp], vlo[2][cp]); split_pair(v3, vhi[3][cp], vlo[3][cp]);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        vhi[c][cp] = __builtin_bit_cast(unsigned, row[c][cp].x);
                        vlo[c][cp] = __builtin_bit_cast(unsigned, row[c][cp].y);
                    }
                }
            }
            // weight fragments one POSITION ahead (16 registers in flight; a whole step ahead would not fit beside 128 accumulator
            // registers at two waves per SIMD), pinned in place by scheduling barriers
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                const f16x8 xh = __builtin_bit_cast(f16x8, make_uint4(vhi[pos][0], vhi[pos][1], vhi[pos][2], vhi[pos][3]));
                const f16x8 xl = __builtin_bit_cast(f16x8, make_uint4(vlo[pos][0], vlo[pos][1], vlo[pos][2], vlo[pos][3]));
                uint4 wn[2][2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc)              // position pos + 1 (the next step's first one behind the last)
                        wn[cb][pc] = (WHAT & 8) ? wp[(woff + (((pos + 1) * 2 + cb) * 2 + pc) * 64) & (unsigned)(w_units - 1)] : opaque();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const f16x8 w0 = __builtin_bit_cast(f16x8, wq[cb][0]), w1 = __builtin_bit_cast(f16x8, wq[cb][1]);
                    if (WHAT & 1) {
                        acc[pos * 2 + cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, xh, acc[pos * 2 + cb], 0, 0, 0);
                        acc[pos * 2 + cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, xl, acc[pos * 2 + cb], 0, 0, 0);
                        acc[pos * 2 + cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, xh, acc[pos * 2 + cb], 0, 0, 0);
                    } else {
                        acc[pos * 2 + cb][pos] += consume(w0) + consume(w1) + consume(xh) + consume(xl);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) wq[cb][pc] = wn[cb][pc];
            }
            woff = (woff + 16 * 64) & (unsigned)(w_units - 1);
        }
    }
    float t = 0.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int MODE, int WHAT>
double run(const char* name, const uint4* wts, int w_units, float* out) {
    const int steps = 2048, grid = 512;
    hipFuncSetAttribute(reinterpret_cast<const void*>(step_kernel<MODE, WHAT>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((step_kernel<MODE, WHAT>), dim3(grid), dim3(256), 32768, 0, wts, w_units, out, 16);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((step_kernel<MODE, WHAT>), dim3(grid), dim3(256), 32768, 0, wts, w_units, out, steps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double us_per_step = best * 1e3 / steps;
    const int mfmas = MODE ? 24 : 54;
    const double tf = (WHAT & 1) ? (double)grid * 4 * steps * mfmas * (2.0 * 32 * 32 * 16) / (best * 1e-3) / 1e12 : 0.0;
    printf("%-58s %8.3f ms  %7.3f us/step  %7.1f TFLOP/s f16 issued\n", name, best, us_per_step, tf);
    return us_per_step;
}

int main() {
    const int w_units = 65536;                                   // 1 MB of "weights": L2-resident, shared by every workgroup
    uint4* wts; hipMalloc(&wts, (size_t)w_units * 16);
    hipMemset(wts, 0x3c, (size_t)w_units * 16);
    float* out; hipMalloc(&out, 512 * 256 * 4);
    printf("work unit per step and workgroup: 16 input channels x 128 output pixels x 64 output channels; 512 workgroups, 2 per CU\n");
    const double d_m = run<0, 1>("direct   : MFMAs only (54)", wts, w_units, out);
    const double d_a = run<0, 15>("direct   : MFMAs + 36 LDS reads + 18 weight loads", wts, w_units, out);
    const double w_m = run<1, 1>("winograd : MFMAs only (24)", wts, w_units, out);
    const double w_v = run<1, 2>("winograd : transform VALU only (160 instr)", wts, w_units, out);
    const double w_mv = run<1, 3>("winograd : MFMAs + transform", wts, w_units, out);
    const double w_a = run<1, 15>("winograd : MFMAs + transform + 16 LDS reads + 16 wt loads", wts, w_units, out);
    printf("direct / winograd: MFMAs only %.2f (arithmetic ideal 2.25), with the transform %.2f, full step %.2f\n", d_m / w_m, d_m / w_mv, d_a / w_a);
    printf("transform alone %.3f us/step = %.2f x the winograd step's MFMA time\n", w_v, w_v / w_m);
    return 0;
}
