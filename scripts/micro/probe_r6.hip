// probe_r6.hip -- two hardware questions behind a FREE saturation guard for the f16x2 kernels (run on the MI355X box):
//  1. MODE.FP16_OVFL (hwreg MODE bit 23): with it set, does v_cvt_pk_f16_f32 / v_cvt_f16_f32 of a float32 beyond 65504 give
//     +-65504 (a hardware clamp: the kernels' v_med3 per value would go) instead of +-inf, with true infinities kept?
//  2. TRAPSTS.EXCP (hwreg TRAPSTS bits 8:0): is the OVERFLOW bit (3) sticky and accumulated with traps disabled -- set by such a
//     conversion (with and without FP16_OVFL), readable by s_getreg_b32 at the end of a work item, clearable by s_setreg_b32?
//     Then "did any value handed to an fp16 split leave the range" costs one scalar read per work item instead of a VALU
//     instruction per value (conv_common.h sat_track; the fused BasicBlock kernels' counting builds cost 1.7-2 % of the job).
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/probe_r6.hip -o scripts/micro/_bin/probe_r6
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// hwreg encoding for s_getreg / s_setreg: id | offset << 6 | (width - 1) << 11.  gfx9: MODE = 1, TRAPSTS = 3
#define HWREG(id, off, width) ((id) | ((off) << 6) | (((width) - 1) << 11))

__global__ void probe(const float* in, int n, int ovfl, unsigned* out_bits, unsigned* flags) {
    const int lane = threadIdx.x;
    if (ovfl) __builtin_amdgcn_s_setreg(HWREG(1, 23, 1), 1);             // MODE.FP16_OVFL = 1
    __builtin_amdgcn_s_setreg(HWREG(3, 0, 9), 0);                        // clear TRAPSTS.EXCP
    const unsigned before = __builtin_amdgcn_s_getreg(HWREG(3, 0, 9));
    unsigned mode = __builtin_amdgcn_s_getreg(HWREG(1, 0, 32));
    // phase 1: only in-range values (lane < n/2 of the table are the in-range half)
    float a = in[lane % (n / 2)], b = in[(lane + 1) % (n / 2)];
    f32x2 v = {a, b};
    f16x2 h = __builtin_convertvector(v, f16x2);
    unsigned bits1 = __builtin_bit_cast(unsigned, h);
    unsigned mid;                                                         // (the VALU result is an operand: the read stays behind the conversion)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(mid), "+v"(bits1));
    // phase 2: lane i converts table entry i (the second half leaves the fp16 range)
    float c = in[lane % n], d = -in[lane % n];
    f32x2 w = {c, d};
    f16x2 g = __builtin_convertvector(w, f16x2);                          // v_cvt_pk_f16_f32 (round to nearest even)
    unsigned bits2 = __builtin_bit_cast(unsigned, g);
    _Float16 s = (_Float16)c;                                             // v_cvt_f16_f32
    unsigned sb = __builtin_bit_cast(unsigned short, s);
    unsigned after;
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(after), "+v"(bits2), "+v"(sb));
    __builtin_amdgcn_s_setreg(HWREG(3, 3, 1), 0);                        // clear the overflow bit alone
    const unsigned cleared = __builtin_amdgcn_s_getreg(HWREG(3, 0, 9));
    out_bits[lane * 2] = bits2;
    out_bits[lane * 2 + 1] = sb | (bits1 << 16);
    if (lane == 0) { flags[0] = before; flags[1] = mid; flags[2] = after; flags[3] = cleared; flags[4] = mode; }
}

static float h2f(unsigned short h) {
    unsigned s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
    float v;
    if (e == 0) v = m * 5.9604645e-8f;
    else if (e == 31) v = m ? __builtin_nanf("") : __builtin_inff();
    else v = (1.0f + m / 1024.0f) * __builtin_powif(2.0f, (int)e - 15);
    return s ? -v : v;
}

int main() {
    // first half in range, second half out of range (65520 is the first float that rounds to inf in fp16)
    const float tab[12] = {1.0f, 4094.0f, 65504.0f, 65519.0f, 1e-7f, 3.0e-6f, 65520.0f, 70000.0f, 1e6f, 3e38f, __builtin_inff(), 131008.0f};
    const int n = 12;
    float* din; unsigned *dout, *dfl;
    hipMalloc((void**)&din, sizeof(tab)); hipMalloc((void**)&dout, 64 * 2 * 4); hipMalloc((void**)&dfl, 32);
    hipMemcpy(din, tab, sizeof(tab), hipMemcpyHostToDevice);
    for (int ovfl = 0; ovfl < 2; ++ovfl) {
        hipMemset(dfl, 0xff, 32);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, din, n, ovfl, dout, dfl);
        unsigned out[128], fl[8];
        if (hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return 1; }
        hipMemcpy(fl, dfl, 32, hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d  MODE=0x%08x  TRAPSTS.EXCP: after clear 0x%03x, after in-range conversions 0x%03x, after out-of-range conversions 0x%03x (overflow bit 3 = %u), after clearing bit 3 0x%03x\n",
               ovfl, fl[4], fl[0], fl[1], fl[2], (fl[2] >> 3) & 1, fl[3]);
        for (int i = 0; i < n; ++i)
            printf("   x = %-12g  cvt_pk(+x) = %-10g cvt_pk(-x) = %-10g cvt(x) = %g\n", tab[i], h2f(out[2 * i] & 0xffff), h2f(out[2 * i] >> 16), h2f(out[2 * i + 1] & 0xffff));
    }
    return 0;
}
