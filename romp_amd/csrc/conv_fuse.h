// conv_fuse.h -- pieces shared by the fused BasicBlock kernels (conv_h2b.hip: 32 channels, conv_h2c.h: 64 channels)
#pragma once
#include "conv_common.h"

namespace romp {

typedef __attribute__((address_space(3))) void lds_void_f;

// (h2_low_pair -- the low fp16 pieces of a value pair as one asm block -- lives in conv_common.h: every epilogue uses it)
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
