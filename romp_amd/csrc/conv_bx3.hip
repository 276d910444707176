// conv_bx3.hip -- instantiations of the bf16x3 split-precision conv kernels (conv_split.h, NP = 3).
#include "conv_split.h"

namespace romp {

template <int KS, int S, int MT, int NT, int TW, int CK>
__global__ __launch_bounds__(256, 2) void conv_bx3_kernel(ConvParams p) {
    if (p.dbg & 32) return;                            // ablation: launch cost only
    conv_split_body<3, KS, S, MT, NT, TW, CK>(p);
}
template <int KS, int S, int MT, int NT, int TW, int CK>
__global__ __launch_bounds__(256, 2) void conv_bxd_kernel(ConvParams p) {
    if (p.dbg & 32) return;
    conv_splitd_body<3, KS, S, MT, NT, TW, CK>(p);
}

#define ROMP_CONV_VARIANT_BXD(KS, S, MT, NT, TW, CK)                                  \
    { KS, S, MT, NT, TW, CK, conv_bxd_kernel<KS, S, MT, NT, TW, CK>,                  \
      SplitCfg<3, KS, S, MT, NT, TW, CK>::LDS_BYTES_DMA, ConvCfg<KS, S, MT, NT, TW, CK>::TH, 0, 0, 2, 0 }
#define ROMP_CONV_VARIANT_BX3(KS, S, MT, NT, TW, CK)                                  \
    { KS, S, MT, NT, TW, CK, conv_bx3_kernel<KS, S, MT, NT, TW, CK>,                  \
      SplitCfg<3, KS, S, MT, NT, TW, CK>::LDS_BYTES, ConvCfg<KS, S, MT, NT, TW, CK>::TH, 0, 0, 1, 0 }
static ConvVariant kVariantsBx3[] = {
    ROMP_CONV_VARIANT_BX3(3, 1, 2, 1, 32, 16), ROMP_CONV_VARIANT_BX3(3, 1, 2, 1, 16, 16),
    ROMP_CONV_VARIANT_BX3(3, 1, 2, 2, 32, 16), ROMP_CONV_VARIANT_BX3(3, 1, 2, 2, 16, 16),
    ROMP_CONV_VARIANT_BX3(3, 1, 1, 2, 32, 16), ROMP_CONV_VARIANT_BX3(3, 1, 1, 2, 16, 16),
    ROMP_CONV_VARIANT_BX3(3, 1, 4, 1, 32, 16), ROMP_CONV_VARIANT_BX3(3, 1, 1, 1, 16, 16),
    ROMP_CONV_VARIANT_BX3(3, 2, 1, 2, 16, 16), ROMP_CONV_VARIANT_BX3(3, 2, 1, 1, 16, 16), ROMP_CONV_VARIANT_BX3(3, 2, 1, 2, 32, 16),
    ROMP_CONV_VARIANT_BX3(1, 1, 2, 2, 32, 32), ROMP_CONV_VARIANT_BX3(1, 1, 2, 1, 32, 32), ROMP_CONV_VARIANT_BX3(1, 1, 1, 2, 16, 32),
    ROMP_CONV_VARIANT_BX3(1, 1, 2, 2, 32, 16), ROMP_CONV_VARIANT_BX3(13, 1, 1, 2, 32, 16), ROMP_CONV_VARIANT_BX3(13, 1, 2, 2, 32, 16),
    ROMP_CONV_VARIANT_BX3(2, 1, 1, 2, 16, 16), ROMP_CONV_VARIANT_BX3(2, 1, 2, 2, 16, 16), ROMP_CONV_VARIANT_BX3(1, 1, 1, 2, 32, 32),
    ROMP_CONV_VARIANT_BXD(3, 1, 2, 2, 16, 16), ROMP_CONV_VARIANT_BXD(3, 1, 2, 2, 32, 16),
    ROMP_CONV_VARIANT_BXD(3, 1, 2, 1, 16, 16), ROMP_CONV_VARIANT_BXD(3, 1, 2, 1, 32, 16),
    ROMP_CONV_VARIANT_BXD(3, 1, 1, 2, 16, 16), ROMP_CONV_VARIANT_BXD(3, 1, 4, 1, 32, 16),
};
ConvVariant* conv_variants_bx3(int* n) { *n = (int)(sizeof(kVariantsBx3) / sizeof(kVariantsBx3[0])); return kVariantsBx3; }

}  // namespace romp
