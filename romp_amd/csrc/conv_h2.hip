// conv_h2.hip -- instantiations of the f16x2 split-precision conv kernels with register-staged weights (conv_split.h, NP = 2).
#include "conv_split.h"

namespace romp {

template <int KS, int S, int MT, int NT, int TW, int CK>
__global__ __launch_bounds__(256, 2) void conv_h2_kernel(ConvParams p) {
    conv_args_now(p);
    if (p.dbg & 32) return;                            // ablation: launch cost only
    conv_split_body<2, KS, S, MT, NT, TW, CK>(p);
}

// the smallest tiles again, register budget of four workgroups per CU (128 VGPRs; <3,1,1,1,16,16> spills 7 dwords, the 1x1 none)
template <int KS, int S, int MT, int NT, int TW, int CK>
__global__ __launch_bounds__(256, 4) void conv_h2o4_kernel(ConvParams p) {
    conv_args_now(p);
    if (p.dbg & 32) return;
    conv_split_body<2, KS, S, MT, NT, TW, CK>(p);
}

#define ROMP_CONV_VARIANT_H2O4(KS, S, MT, NT, TW, CK)                                 \
    { KS, S, MT, NT, TW, CK, conv_h2o4_kernel<KS, S, MT, NT, TW, CK>,                 \
      SplitCfg<2, KS, S, MT, NT, TW, CK>::LDS_BYTES, ConvCfg<KS, S, MT, NT, TW, CK>::TH, 0, 0, 3, 0, 1 }
#define ROMP_CONV_VARIANT_H2(KS, S, MT, NT, TW, CK)                                   \
    { KS, S, MT, NT, TW, CK, conv_h2_kernel<KS, S, MT, NT, TW, CK>,                   \
      SplitCfg<2, KS, S, MT, NT, TW, CK>::LDS_BYTES, ConvCfg<KS, S, MT, NT, TW, CK>::TH, 0, 0, 3, 0 }
static ConvVariant kVariantsH2[] = {
    ROMP_CONV_VARIANT_H2(3, 1, 2, 1, 32, 16), ROMP_CONV_VARIANT_H2(3, 1, 2, 1, 16, 16),
    ROMP_CONV_VARIANT_H2(3, 1, 2, 2, 32, 16), ROMP_CONV_VARIANT_H2(3, 1, 2, 2, 16, 16),
    ROMP_CONV_VARIANT_H2(3, 1, 1, 2, 32, 16), ROMP_CONV_VARIANT_H2(3, 1, 1, 2, 16, 16),
    ROMP_CONV_VARIANT_H2(3, 1, 4, 1, 32, 16), ROMP_CONV_VARIANT_H2(3, 1, 1, 1, 16, 16),
    ROMP_CONV_VARIANT_H2(3, 2, 1, 2, 16, 16), ROMP_CONV_VARIANT_H2(3, 2, 1, 1, 16, 16), ROMP_CONV_VARIANT_H2(3, 2, 1, 2, 32, 16),
    ROMP_CONV_VARIANT_H2(1, 1, 2, 2, 32, 32), ROMP_CONV_VARIANT_H2(1, 1, 2, 1, 32, 32), ROMP_CONV_VARIANT_H2(1, 1, 1, 2, 16, 32),
    ROMP_CONV_VARIANT_H2(1, 1, 2, 2, 32, 16), ROMP_CONV_VARIANT_H2(13, 1, 1, 2, 32, 16), ROMP_CONV_VARIANT_H2(13, 1, 2, 2, 32, 16),
    ROMP_CONV_VARIANT_H2(2, 1, 1, 2, 16, 16), ROMP_CONV_VARIANT_H2(2, 1, 2, 2, 16, 16), ROMP_CONV_VARIANT_H2(1, 1, 1, 2, 32, 32),
    ROMP_CONV_VARIANT_H2(1, 1, 1, 1, 16, 32), ROMP_CONV_VARIANT_H2(1, 1, 1, 1, 32, 32), ROMP_CONV_VARIANT_H2(1, 1, 2, 1, 16, 32),
    ROMP_CONV_VARIANT_H2(1, 1, 1, 1, 16, 16), ROMP_CONV_VARIANT_H2(2, 1, 1, 1, 16, 16), ROMP_CONV_VARIANT_H2(1, 2, 1, 1, 16, 32),
    ROMP_CONV_VARIANT_H2(1, 1, 4, 2, 32, 32), ROMP_CONV_VARIANT_H2(1, 1, 4, 1, 32, 32), ROMP_CONV_VARIANT_H2(1, 2, 1, 2, 16, 32),
    // 32 input channels per stage: a 32-channel layer is ONE stage per work item (half the barriers and load-issue passes)
    // (wider tiles at this depth spill: <3,1,1,2,16,32> 196 B, <3,1,2,1,16,32> 232 B of scratch)
    ROMP_CONV_VARIANT_H2(3, 1, 1, 1, 16, 32),
    ROMP_CONV_VARIANT_H2O4(3, 1, 1, 1, 16, 16), ROMP_CONV_VARIANT_H2O4(1, 1, 1, 1, 32, 32), ROMP_CONV_VARIANT_H2O4(1, 1, 1, 1, 16, 32),
};
ConvVariant* conv_variants_h2(int* n) { *n = (int)(sizeof(kVariantsH2) / sizeof(kVariantsH2[0])); return kVariantsH2; }

}  // namespace romp
