// conv_h2d.hip -- instantiations of the f16x2 split-precision conv kernels with LDS-DMA weight rows (conv_split.h, NP = 2).
#include "conv_split.h"

namespace romp {

template <int KS, int S, int MT, int NT, int TW, int CK>
__global__ __launch_bounds__(256, 2) void conv_h2d_kernel(ConvParams p) {
    conv_args_now(p);
    if (p.dbg & 32) return;                            // ablation: launch cost only
    conv_splitd_body<2, KS, S, MT, NT, TW, CK>(p);
}

// one-block tiles again with the register budget of four workgroups per CU (115 / 123 VGPRs, no spill): more items in flight per CU
template <int KS, int S, int MT, int NT, int TW, int CK>
__global__ __launch_bounds__(256, 4) void conv_h2do4_kernel(ConvParams p) {
    conv_args_now(p);
    if (p.dbg & 32) return;
    conv_splitd_body<2, KS, S, MT, NT, TW, CK>(p);
}
#define ROMP_CONV_VARIANT_H2DO4(KS, S, MT, NT, TW, CK)                                \
    { KS, S, MT, NT, TW, CK, conv_h2do4_kernel<KS, S, MT, NT, TW, CK>,                \
      SplitCfg<2, KS, S, MT, NT, TW, CK>::LDS_BYTES_DMA, ConvCfg<KS, S, MT, NT, TW, CK>::TH, 0, 0, 4, 0, 1 }
#define ROMP_CONV_VARIANT_H2D(KS, S, MT, NT, TW, CK)                                  \
    { KS, S, MT, NT, TW, CK, conv_h2d_kernel<KS, S, MT, NT, TW, CK>,                  \
      SplitCfg<2, KS, S, MT, NT, TW, CK>::LDS_BYTES_DMA, ConvCfg<KS, S, MT, NT, TW, CK>::TH, 0, 0, 4, 0 }
static ConvVariant kVariantsH2d[] = {
    ROMP_CONV_VARIANT_H2D(3, 1, 2, 2, 16, 16), ROMP_CONV_VARIANT_H2D(3, 1, 2, 2, 32, 16),
    ROMP_CONV_VARIANT_H2D(3, 1, 2, 1, 16, 16), ROMP_CONV_VARIANT_H2D(3, 1, 2, 1, 32, 16),
    ROMP_CONV_VARIANT_H2D(3, 1, 1, 2, 16, 16), ROMP_CONV_VARIANT_H2D(3, 1, 4, 1, 32, 16), ROMP_CONV_VARIANT_H2D(3, 1, 1, 1, 16, 16),
    ROMP_CONV_VARIANT_H2D(3, 1, 1, 2, 32, 16), ROMP_CONV_VARIANT_H2D(3, 1, 1, 1, 32, 16),
    ROMP_CONV_VARIANT_H2D(3, 1, 4, 2, 16, 16), ROMP_CONV_VARIANT_H2D(3, 1, 4, 2, 32, 16),
    ROMP_CONV_VARIANT_H2D(3, 1, 2, 2, 16, 32), ROMP_CONV_VARIANT_H2D(3, 1, 2, 2, 32, 32),
    ROMP_CONV_VARIANT_H2D(3, 1, 4, 1, 16, 16), ROMP_CONV_VARIANT_H2D(3, 1, 4, 1, 16, 32),
    ROMP_CONV_VARIANT_H2D(3, 2, 1, 2, 16, 16),
    ROMP_CONV_VARIANT_H2D(3, 2, 1, 1, 16, 16), ROMP_CONV_VARIANT_H2D(3, 2, 1, 2, 32, 16),
    ROMP_CONV_VARIANT_H2DO4(3, 1, 1, 1, 16, 16), ROMP_CONV_VARIANT_H2DO4(3, 1, 1, 1, 32, 16),
};
ConvVariant* conv_variants_h2d(int* n) { *n = (int)(sizeof(kVariantsH2d) / sizeof(kVariantsH2d[0])); return kVariantsH2d; }

}  // namespace romp
