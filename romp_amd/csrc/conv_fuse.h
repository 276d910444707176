// conv_fuse.h -- pieces shared by the fused BasicBlock kernels (conv_h2b.hip: 32 channels, conv_h2c.hip: 64 channels)
#pragma once
#include "conv_common.h"

namespace romp {

typedef __attribute__((address_space(3))) void lds_void_f;

// Mixed-precision steps as ONE asm block each (v_fma_mix_f32 takes an fp16 operand as it is; hipcc turns `fma(ext(h), +-1, x)` into a
// convert and an add, and follows every single-instruction asm whose result is used at once with an s_nop).
// fp16x2 of (a - hi.x, b - hi.y): the low pieces of two values whose packed high pieces are `hi`
__device__ __forceinline__ unsigned h2_low_pair(unsigned hi, float a, float b) {
    unsigned lo;
    float ta, tb;
    asm("v_fma_mix_f32 %1, %3, -1.0, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %3, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_cvt_pk_f16_f32 %0, %1, %2"
        : "=v"(lo), "=&v"(ta), "=&v"(tb) : "v"(hi), "v"(a), "v"(b));
    return lo;
}
// min(max(x + half HALF of rh + half HALF of rl, 0), top): the residual's two pieces, ReLU, saturation
template <int HALF>
__device__ __forceinline__ float add_pieces_relu(float x, unsigned rh, unsigned rl, float top) {
    float d;
    if (HALF == 0)
        asm("v_fma_mix_f32 %0, %1, 1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %0, %2, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_med3_f32 %0, %0, 0, %4" : "=&v"(d) : "v"(rh), "v"(rl), "v"(x), "v"(top));
    else
        asm("v_fma_mix_f32 %0, %1, 1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %0, %2, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_med3_f32 %0, %0, 0, %4" : "=&v"(d) : "v"(rh), "v"(rl), "v"(x), "v"(top));
    return d;
}

}  // namespace romp
