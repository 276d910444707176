// common.h -- shared helpers for libromp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/romp_hip.h"

namespace romp {

void set_error(const char* fmt, ...);

#define ROMP_HIP_CHECK(expr)                                                          \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            romp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                            __FILE__, __LINE__);                                      \
            return ROMP_EHIP;                                                         \
        }                                                                             \
    } while (0)

#define ROMP_REQUIRE(cond, ...)                                                       \
    do {                                                                              \
        if (!(cond)) {                                                                \
            romp::set_error(__VA_ARGS__);                                             \
            return ROMP_EINVAL;                                                       \
        }                                                                             \
    } while (0)

// work queues of the persistent conv kernels: 8 counters (one per XCD), each on its own 128-byte line
constexpr int QUEUE_STRIDE = 32, QUEUE_INTS = 8 * QUEUE_STRIDE;

// launchers implemented in the kernel translation units
// variant < 0: heuristic choice; queue: QUEUE_INTS zeroed ints (nullptr: library scratch, memset on `st`)
// wg_cap > 0: at most that many workgroups per CU (room for a co-resident kernel of another stream)
int launch_conv(const romp_op& op, const float* in, const float* res, float* out, int B,
                int mode, int variant, int* queue, hipStream_t st, int wg_cap = 0);
int describe_conv(const romp_op& op, int B, int variant, char* out, int n);
int conv_num_variants();
int conv_family_variants(int math);                              // variants of one kernel family (ConvVariant.math) in this build
int conv_trace_read(unsigned long long* dst_host, int max_words);
int conv_init();
bool conv_variant_valid(const romp_op& op, int variant);
bool conv_variant_tunable(const romp_op& op, int variant);   // ... and offered to romp_net_autotune
void conv_set_sat_counter(int* counter, bool checked_fused);   // conv_mfma.hip: the device counter the following launches report clamped values to
int* conv_sat_counter();
void conv_set_wg_cap(int cap);                                 // fused kernels: workgroups per CU they may take (the net's wg_cap; 0 = all)
int conv_wg_cap();
bool conv_sat_checked();                                       // fused-block kernels: launch the builds that count too
unsigned long long* conv_trace_arm(hipStream_t st);   // conv_mfma.hip: ROMP_CONV_TRACE stamp buffer, zeroed on `st` (nullptr: off)
int launch_seam1x1(const romp_op& opa, const romp_op& opb, const romp_op* opd, const float* m, const float* x, float* t, float* u, int B, hipStream_t st);   // conv_h2x.hip
int launch_bblock32r(const romp_op& op1, const romp_op& op2, const float* x, float* y, int B, int* queue, hipStream_t st);  // conv_h2c.hip
int launch_bblock64(const romp_op& op1, const romp_op& op2, const float* x, float* y, int B, int* queue, hipStream_t st);   // conv_h2c.hip
int launch_bblock32(const romp_op& op1, const romp_op& op2, const float* x, float* y, int B, int* queue, hipStream_t st);
int launch_stem(const romp_op& op, const float* image, float* out, int B, hipStream_t st);
int launch_stem7(const romp_op& op, const float* image, float* out, int B, hipStream_t st);
int launch_stem2(const romp_op& stem, const romp_op& op, const float* image, float* out, int B, hipStream_t st);   // stem2.hip (image == out == nullptr: set-up only)
int launch_maxpool(const romp_op& op, const float* in, float* out, int B, hipStream_t st);
int launch_stem7p(const romp_op& op, const float* image, float* out, int B, hipStream_t st);   // stem7p.hip (image == out == nullptr: set-up only)
struct FuseTerm { const float* ptr; int shift; int cstride; int fmt; };
int launch_fusesum(const FuseTerm* terms, int n_terms, float* out, int B, int H, int W, int C,
                   int out_cstride, int out_coff, int relu, hipStream_t st, int out_fmt = 0, int act_shift = 0);

int launch_fuseup(const romp_op& op, const FuseTerm* terms, float* out, int B, hipStream_t st);      // conv_fup.hip (terms == nullptr: set-up only)

int launch_ksum(const romp_op& op, const float* partial, const float* res, float* out, int B, hipStream_t st);

// BEV head pieces (bev.hip); *_host pointers are dereferenced on the host at launch time
int launch_bev_pack(const float* fv, int fv_cs, const float* feats, int f_cs, float* out, int B, hipStream_t st);
int launch_bev_maps(const float* fv, int fv_cs, const float* bv, int bv_cs, const float* anchors_host, float* center3d,
                    float* cam3d, int B, hipStream_t st);
int launch_conv3d(int C, const float* w_host, const float* scale_host, const float* shift_host, int relu, const float* in,
                  const float* res, float* out, int B, hipStream_t st);

}  // namespace romp
