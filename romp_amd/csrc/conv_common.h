// conv_common.h -- shared pieces of the conv kernel translation units (gfx950 only): launch parameters,
// tile configuration, work-item decoding, the fused epilogue and the variant table entry.
#pragma once
#include "common.h"
#include <stdlib.h>

namespace romp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float* in; const float* w; const float* scale; const float* shift; const float* res;
    float* out;
    const uint4* w3;          // bf16x3-split weights (conv_bx3 / conv_bxd kernels), or nullptr
    const uint4* wh;          // f16x2-split weights (conv_h2 / conv_h2d kernels), or nullptr
    const float* scale_h;     // their epilogue scale (scale / (weight scale * act_scale))
    const float* zero;        // >= 16 bytes of zeros in device memory (what out-of-image lanes of an LDS-DMA fetch)
    float act_scale;          // 2^act_shift, applied to the activations before the fp16 split
    float inv_act_scale;      // 2^-act_shift
    int in_h2, out_h2, res_h2;   // tensor formats: 0 = NHWC float32, 1 = H2 (pre-split fp16 pieces, see h2_pack below)
    int* queue;               // 8 per-XCD work counters, QUEUE_STRIDE ints apart, zeroed before the launch
    int H, W, Ho, Wo;
    int Cout;                 // valid output channels per group (store mask)
    int cin_valid;            // channels physically present in the input (loader mask)
    int cin_pad, cout_pad;    // packed weight dims
    int in_cs, in_co, in_gs;
    int out_cs, out_co, out_gs;
    int res_cs, res_co, res_gs;
    int relu;
    int tiles_x, tiles_y, tiles_total;
    int nslices, ns_total;    // channel slices per group; slices*groups
    int n_queues, per_queue;  // 8 (XCD-aware) or 1
    int vec_io;               // epilogue may use float4 loads/stores
    int w_gs;                 // floats per group in the packed weight
    int pad_h, pad_w;         // rows / columns of zero padding before the first tap
    int out_rs, out_bs;       // output row stride / image stride in floats (dense: Wo*out_cs, Ho*Wo*out_cs)
    unsigned long long* trace;   // nullptr, or TRACE_SLOTS words per wave: (s_memtime << 8 | event code) stamps (env ROMP_CONV_TRACE=1;
                                 // split-precision kernels only; read back with romp_conv_trace_read, scripts/conv_trace.py)
    float* out2; int out2_cs, out2_co;   // (conv_h2x.hip) the second output tensor
    const uint4* wx; const float* scale_x; const float* shift_x;   // (conv_h2x.hip, DS) the folded downsample conv: weights, f16x2 scale, shift
    unsigned res_bytes;       // (conv_h2x.hip, DS) bytes of the tensor at res + res_co (the downsample's input): num_records of its raw buffer
    unsigned in_bytes;        // (fused block kernel) bytes of the input tensor from in + in_co on: num_records of its raw buffer
    int relu_from;            // relu != 0: ReLU on output channels >= relu_from only (romp_op.relu_from; a multiple of 32)
    int* sat;                 // saturation counter of the running net (conv_sat_counter(), may be nullptr): +1 per wave and work item that
                              // clamped a value at +-65504 while splitting it into fp16 pieces (h2_sat): out-of-calibration activations
    int run_len;              // (fused block kernel, strip form) vertically consecutive tiles per run: a work item is a RUN, its tiles hand two
                              // rows of the intermediate on to each other (conv_h2c.h); tiles_y % run_len == 0
    int dbg;                  // ablation switches, env ROMP_CONV_DEBUG (timing experiments only: outputs are wrong).
                              // bits: 1 skip global loads / DMA, 2 skip LDS staging writes, 4 skip epilogue, 8 skip MFMA loop,
                              // 512 (h2r / h2s, correct outputs) the LDS-transposed epilogue instead of the direct one,
                              // 16 skip a stage barrier (f32 kernel), 32 return at once (launch cost), 64 / 128 (bxd) skip
                              // only the pixel loads / only the weight DMA; 64 (h2r): every weight fragment load re-reads the first
                              // tap's 2 KB (same instructions, no weight stream from L2).  scripts/conv_ablate.py, conv_sweep.py and
                              // DESIGN.md §4 use them.
};

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// The arguments a conv kernel's set-up reads -- pointers, geometry, the work split -- asked for in ONE scalar-memory round trip at the
// kernel's first instruction.  hipcc loads a by-value argument where its first use is: the kernels' set-up (early exit, item decode,
// DMA tables, first stage) then runs 4-8 DEPENDENT s_load / s_waitcnt rounds of ~0.3-0.5 us in front of the first memory request
// (round 6: profiles/r06s_trace3.txt, r06s_args_ab.txt).  An empty asm that names the fields as SGPR inputs pins their loads here, in
// one clause under one wait; a field a kernel does not use costs it one more dword of that clause.  -DROMP_NO_ARGS_BATCH: the A/B build.
__device__ __forceinline__ void conv_args_now(const ConvParams& p) {
#ifndef ROMP_NO_ARGS_BATCH
    asm volatile("" :: "s"(p.in), "s"(p.out), "s"(p.res), "s"(p.wh), "s"(p.scale_h), "s"(p.shift), "s"(p.zero), "s"(p.queue), "s"(p.trace),
                 "s"(p.H), "s"(p.W), "s"(p.Ho), "s"(p.Wo), "s"(p.Cout), "s"(p.cin_valid), "s"(p.cin_pad), "s"(p.cout_pad),
                 "s"(p.in_cs), "s"(p.in_co), "s"(p.in_gs), "s"(p.tiles_x), "s"(p.tiles_y), "s"(p.tiles_total), "s"(p.nslices), "s"(p.ns_total),
                 "s"(p.n_queues), "s"(p.per_queue), "s"(p.pad_h), "s"(p.pad_w), "s"(p.dbg), "s"(gridDim.x));
#endif
}

// ---- the H2 activation format -------------------------------------------------------------------------------------
// An activation tensor the f16x2 kernels consume can live in HBM already split: per pixel and channel OCTET o (channels
// 8o..8o+7) one 16-byte unit with the eight HIGH fp16 pieces h1 followed by one with the eight LOW pieces h2, where
// x * 2^act_shift = h1 + h2 (up to 2^-22 relative).  An octet occupies the 32 bytes its eight floats would, so channel
// strides / offsets (multiples of 8) and every address computation are those of the float32 tensor.  Consumers copy the
// units to LDS as they are (no conversion work, and they can be moved by LDS-DMA); producers split once in their
// epilogue; residual adds and fuse sums use h1 + h2 (22 significant bits: 2^-23 relative, f32 rounding class).
// Range: |x * 2^act_shift| beyond fp16's largest finite value SATURATES at +-65504 (v_med3: two VALU per value) instead of
// turning the high piece into inf and the low piece into NaN; which tensors may be H2 at all is decided from measured
// activation ranges (plan.assign_formats, romp_net_range_scan), the clamp is what keeps an unforeseen outlier finite.
constexpr float H2_MAX = 65504.f;
__device__ __forceinline__ float h2_sat(float x) { return __builtin_fminf(__builtin_fmaxf(x, -H2_MAX), H2_MAX); }
// Saturation is OBSERVABLE: every site that clamps keeps a per-lane running max of |value before the clamp| (one
// v_max3_f32 per two values) and reports once per wave and work item -- sat_report -- into the net's device counter
// (romp_net_saturated; RompNet.saturated).  The two register-resident fused-block kernels (conv_h2b / conv_h2c.h), whose side work
// is placed instruction by instruction, count in their checked builds only (romp_net_range_scan and ROMP_CHECK_FINITE=1 run those).
__device__ __forceinline__ void sat_track(float& mx, float a, float b) { mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b))); }
__device__ __forceinline__ void sat_report(int* counter, float mx) {
    if (counter && __builtin_amdgcn_ballot_w64(mx >= H2_MAX) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(counter, 1);
}
// The packed form for POST-ReLU values (non-negative; round 6): the running per-half maximum of the HIGH-piece dwords a kernel has
// just formed -- v_pk_maximum3_f16, new in gfx950: ONE instruction per two dwords = four values (the float32 form above is one per
// two; the fused BasicBlock kernels' counting builds used one per value and cost 1.7-2 % of the job, profiles/r06_guard_cost.txt).
// A half reads 0x7BFF (65504) iff its value was clamped, 0x7E00 if it was a NaN: either is reported.
__device__ __forceinline__ unsigned sat_track_pk(unsigned mx, unsigned h0, unsigned h1) {
    unsigned d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(mx), "v"(h0), "v"(h1));
    return d;
}
__device__ __forceinline__ void sat_report_pk(int* counter, unsigned mx) {
    const bool hit = (mx & 0x7fffu) >= 0x7bffu || ((mx >> 16) & 0x7fffu) >= 0x7bffu;
    if (counter && __builtin_amdgcn_ballot_w64(hit) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(counter, 1);
}
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// fp16x2 of (a - hi.x, b - hi.y): the low pieces of two values whose packed high pieces are `hi`.  ONE asm block (v_fma_mix_f32 takes an
// fp16 operand as it is; hipcc turns `x - (float)h` into a convert and a subtract, and follows every single-instruction asm whose result
// is used at once with an s_nop): 3 instructions per value pair instead of 6.  The difference is exact in float32, so the result
// is the one the plain C form gives.
__device__ __forceinline__ unsigned h2_low_pair(unsigned hi, float a, float b) {
    unsigned lo;
    float ta, tb;
    asm("v_fma_mix_f32 %1, %3, -1.0, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %3, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_cvt_pk_f16_f32 %0, %1, %2"
        : "=v"(lo), "=&v"(ta), "=&v"(tb) : "v"(hi), "v"(a), "v"(b));
    return lo;
}
__device__ __forceinline__ unsigned h2_high_pair(float a, float b) {      // v_cvt_pk_f16_f32: round to nearest even
    typedef float f32x2_p __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_p __attribute__((ext_vector_type(2)));
    const f32x2_p v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_p));
}
__device__ __forceinline__ void h2_pack(float4 v, float act_scale, uint2& hi, uint2& lo, float& mx) {
    const float s[4] = {v.x * act_scale, v.y * act_scale, v.z * act_scale, v.w * act_scale};
    sat_track(mx, s[0], s[1]);
    sat_track(mx, s[2], s[3]);
    const float x[4] = {h2_sat(s[0]), h2_sat(s[1]), h2_sat(s[2]), h2_sat(s[3])};
    // Packed conversions (v_cvt_pk_f16_f32), plain C for the low pieces.  NOT h2_low_pair's asm block here: round 4 measured it in
    // this shared epilogue and the heavily spilling conv_h2_kernel<1,1,4,2,32,32> (784 bytes of scratch per lane) then faulted on
    // the head's 64 -> 142 conv -- in the scalar-store path that never executes the block; the plain form of the same arithmetic
    // does not (scripts/attic/gpu_r4g.sh isolates it).  The fused kernels keep the asm: they do not spill.
    typedef float f32x2_p __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_p __attribute__((ext_vector_type(2)));
    const f16x2_p h0 = __builtin_convertvector((f32x2_p){x[0], x[1]}, f16x2_p), h1 = __builtin_convertvector((f32x2_p){x[2], x[3]}, f16x2_p);
    const f16x2_p l0 = __builtin_convertvector((f32x2_p){x[0] - (float)h0[0], x[1] - (float)h0[1]}, f16x2_p);      // the subtractions are exact
    const f16x2_p l1 = __builtin_convertvector((f32x2_p){x[2] - (float)h1[0], x[3] - (float)h1[1]}, f16x2_p);
    hi = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
    lo = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
}
__device__ __forceinline__ void h2_pack(float4 v, float act_scale, uint2& hi, uint2& lo) {
    float mx = 0.f;
    h2_pack(v, act_scale, hi, lo, mx);
}
__device__ __forceinline__ float4 h2_unpack(uint2 hi, uint2 lo, float inv_act_scale) {
    const f16x4 h = __builtin_bit_cast(f16x4, hi), l = __builtin_bit_cast(f16x4, lo);
    return make_float4(((float)h[0] + (float)l[0]) * inv_act_scale, ((float)h[1] + (float)l[1]) * inv_act_scale,
                       ((float)h[2] + (float)l[2]) * inv_act_scale, ((float)h[3] + (float)l[3]) * inv_act_scale);
}

constexpr int EPI_ROW = 144;                         // epilogue staging: bytes per pixel row: 32 floats + 16 (conflict-free 16-byte columns)
constexpr int EPI_WAVE = 32 * EPI_ROW;               // per-wave staging tile
constexpr int EPI_BYTES = 4 * EPI_WAVE;

template <int KS, int S, int MT, int NT, int TW, int CK, int NWV = 4>
struct ConvCfg {                                     // NWV: waves per workgroup (each owns MT pixel blocks x NT channel blocks)
    // KS = 1: 1x1, 2: 2x2 (one output parity of a ConvTranspose2d k4 s2), 3: 3x3, 13: 1x3 (Conv1d k=3 along W; rows of
    // the "image" are independent sequences).  The zero padding before the first tap is a run-time parameter.
    static constexpr int KH = (KS == 13) ? 1 : KS;
    static constexpr int KW = (KS == 13) ? 3 : KS;
    static constexpr int TAPS = KH * KW;
    static constexpr int RPB = 32 / TW;              // tile rows per 32-pixel block
    static constexpr int TH = NWV * MT * RPB;        // output tile rows
    static constexpr int HR = (TH - 1) * S + KH;     // haloed input rows
    static constexpr int HC = (TW - 1) * S + KW;
    static constexpr int PS = CK + 4;                // LDS floats per pixel (padded)
    static constexpr int NW = NT * 32;               // output channels per work item
    static constexpr int QC = CK / 4;                // float4 per pixel per chunk
    static constexpr int A_VEC = HR * HC * QC;
    static constexpr int B_VEC = TAPS * QC * NW;
    static constexpr int NA = (A_VEC + 255) / 256;
    static constexpr int NB = (B_VEC + 255) / 256;
    static constexpr int LDS_MAIN = (HR * HC * PS + TAPS * CK * NW + 4 * NW) * 4 + 16;      // pixels, weights, scale/shift slots, mailbox
    static constexpr int LDS_BYTES = LDS_MAIN + EPI_BYTES;                                   // + the epilogue's staging tiles
};

struct Item { int b, ty, tx, n0, g; };

constexpr int TRACE_SLOTS = 64, TRACE_WAVES = 4096;
// one stamp per wave (lane 0): word 0 of the wave's slot block counts the stamps, words 1.. hold them.  The kernel defines
// `int tr_n = 0` and `constexpr int tr_wpw` = its waves per workgroup.
#define ROMP_TRACE(code)                                                                             \
    do {                                                                                             \
        if (p.trace) {                                                                               \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                              \
            const unsigned w_ = blockIdx.x * (unsigned)tr_wpw + (threadIdx.x >> 6);                  \
            if ((threadIdx.x & 63) == 0 && w_ < (unsigned)TRACE_WAVES && tr_n < TRACE_SLOTS - 1) {   \
                p.trace[(size_t)w_ * TRACE_SLOTS + 1 + tr_n] = (t_ << 8) | (unsigned)(code);         \
                p.trace[(size_t)w_ * TRACE_SLOTS] = (unsigned long long)(tr_n + 1);                  \
            }                                                                                        \
            ++tr_n;                                                                                  \
        }                                                                                            \
    } while (0)

__device__ __forceinline__ Item decode_item(const ConvParams& p, int q, int j, int NW) {
    const int s = j % p.ns_total, tl = j / p.ns_total;
    // A queue (= an XCD, blockIdx % 8) owns a contiguous run of tiles_total / 8 tiles, i.e. whole images / image halves: the
    // workgroups of one XCD then work on spatially adjacent tiles at the same time and the 3x3 halo re-reads hit that XCD's L2
    // (round 3: with every 8th tile per queue the measured FETCH_SIZE of the 16x16-tile 3x3 kernels was 1.27-1.31x algorithmic =
    // exactly their haloed / plain pixel ratio -- every halo row came over the fabric again).
    int t = q * (p.tiles_total / p.n_queues) + tl;
    Item it;
    it.g = s / p.nslices;
    it.n0 = (s % p.nslices) * NW;
    it.tx = t % p.tiles_x; t /= p.tiles_x;
    it.ty = t % p.tiles_y;
    it.b = t / p.tiles_y;
    return it;
}

// Epilogue of one work item: y = acc*scale + shift (+ residual) (ReLU).
// The MFMA leaves a lane with pixel li of pixel-block m and channels n0 + n*32 + 8*g4 + 4*lh + {0..3}: stored from there, one
// store instruction covers 32 bytes of each of 32 pixels.  Instead every 32-pixel x 32-channel block goes through a per-wave LDS
// staging tile (32 rows of 128 + 16 bytes) and comes back TRANSPOSED: lane L owns channel octet L & 3 of pixels (L >> 2) and
// (L >> 2) + 16, i.e. 32 contiguous bytes of a pixel, four lanes cover the block's whole 128-byte pixel row.  Residual loads and
// output stores are then 16-byte accesses, 128 contiguous bytes per pixel -- in the float32 format and in H2 alike (an octet's
// high and low units are the 32 bytes its floats would be).  All residual loads of the item are issued up front in one batch
// under one uniform branch; ReLU is branch-free (max with 0 or -inf).
template <int MT, int NT>
struct EpiRes { float4 ra[MT][NT][2], rb[MT][NT][2]; };   // residual values of an item in the transposed (store) ownership

// Issue the residual loads of an item early (e.g. before its last MFMA block): the epilogue then finds them in registers.
template <int KS, int S, int MT, int NT, int TW, int CK, int NWV = 4>
__device__ __forceinline__ void conv_epilogue_prefetch(const ConvParams& p, const Item& cur, int wave, int lane, EpiRes<MT, NT>& pre) {
    using C = ConvCfg<KS, S, MT, NT, TW, CK, NWV>;
    const float* res = p.res + (size_t)cur.b * p.Ho * p.Wo * p.res_cs + p.res_co + cur.g * p.res_gs;
    const int oc = lane & 3;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pp = (lane >> 2) + 16 * j, mb = wave * MT + m;
            const int oy = cur.ty * C::TH + mb * C::RPB + pp / TW, ox = cur.tx * TW + pp % TW;
            const unsigned pixo = oy < p.Ho ? (unsigned)(oy * p.Wo + ox) : 0u;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float* rp = res + (pixo * (unsigned)p.res_cs + (unsigned)(cur.n0 + n * 32 + oc * 8));
                pre.ra[m][n][j] = ldg4(rp);
                pre.rb[m][n][j] = ldg4(rp + 4);
            }
        }
}

template <int KS, int S, int MT, int NT, int TW, int CK, int NWV = 4>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, const Item& cur, f32x16 (&acc)[MT][NT],
                                              const float* sSc, char* sE, int wave, int li, int lh, const EpiRes<MT, NT>& pre, bool use_pre) {
    using C = ConvCfg<KS, S, MT, NT, TW, CK, NWV>;
    float* out = p.out + (size_t)cur.b * p.out_bs + p.out_co + cur.g * p.out_gs;
    const float* res = p.res ? p.res + (size_t)cur.b * p.Ho * p.Wo * p.res_cs + p.res_co + cur.g * p.res_gs : nullptr;
    // ReLU is a max with 0 or -inf, per 32-channel block: a merged conv (romp_op.relu_from) mixes blocks with and without it
    auto floor_of = [&](int co) { return (p.relu && co >= p.relu_from) ? 0.f : -__builtin_inff(); };
    float sat_mx = 0.f;
    if (p.vec_io) {
        const int lane = lh * 32 + li;
        const int oc = lane & 3;                      // channel octet of the 32-channel block this lane stores
        unsigned pixo[MT][2], outo[MT][2];            // residual pixel index; output offset (row / pixel strides may be sparse)
        bool rowok[MT][2];                            // partial tiles along H (e.g. Conv1d over B < TH sequences)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pp = (lane >> 2) + 16 * j, mb = wave * MT + m;
                const int oy = cur.ty * C::TH + mb * C::RPB + pp / TW, ox = cur.tx * TW + pp % TW;
                rowok[m][j] = oy < p.Ho;
                pixo[m][j] = rowok[m][j] ? (unsigned)(oy * p.Wo + ox) : 0u;
                outo[m][j] = rowok[m][j] ? (unsigned)(oy * p.out_rs + ox * p.out_cs) : 0u;
            }
        float4 ra[MT][NT][2], rb[MT][NT][2];          // residual: channels 8*oc .. +3 / +4 .. +7 (float32), or high / low unit (H2)
        if (res && use_pre) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int j = 0; j < 2; ++j) { ra[m][n][j] = pre.ra[m][n][j]; rb[m][n][j] = pre.rb[m][n][j]; }
        } else if (res) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {     // masked rows read pixel 0 (valid memory)
                        const float* rp = res + (pixo[m][j] * (unsigned)p.res_cs + (unsigned)(cur.n0 + n * 32 + oc * 8));
                        ra[m][n][j] = ldg4(rp);
                        rb[m][n][j] = ldg4(rp + 4);
                    }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float floor_v = floor_of(cur.n0 + n * 32);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int cl = n * 32 + g4 * 8 + lh * 4;
                    const float4 sc = *reinterpret_cast<const float4*>(sSc + cl);
                    const float4 sh = *reinterpret_cast<const float4*>(sSc + C::NW + cl);
                    float4 v;
                    v.x = fmaf(acc[m][n][g4 * 4 + 0], sc.x, sh.x);
                    v.y = fmaf(acc[m][n][g4 * 4 + 1], sc.y, sh.y);
                    v.z = fmaf(acc[m][n][g4 * 4 + 2], sc.z, sh.z);
                    v.w = fmaf(acc[m][n][g4 * 4 + 3], sc.w, sh.w);
                    *reinterpret_cast<float4*>(sE + li * EPI_ROW + (g4 * 8 + lh * 4) * 4) = v;
                }
                __builtin_amdgcn_wave_barrier();      // same wave: the DS unit executes its writes and reads in order
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int pp = (lane >> 2) + 16 * j;
                    float4 va = *reinterpret_cast<const float4*>(sE + pp * EPI_ROW + oc * 32);
                    float4 vb = *reinterpret_cast<const float4*>(sE + pp * EPI_ROW + oc * 32 + 16);
                    if (res) {
                        float4 qa = ra[m][n][j], qb = rb[m][n][j];
                        if (p.res_h2) {
                            const uint4 hi = __builtin_bit_cast(uint4, qa), lo = __builtin_bit_cast(uint4, qb);
                            qa = h2_unpack(make_uint2(hi.x, hi.y), make_uint2(lo.x, lo.y), p.inv_act_scale);
                            qb = h2_unpack(make_uint2(hi.z, hi.w), make_uint2(lo.z, lo.w), p.inv_act_scale);
                        }
                        va.x += qa.x; va.y += qa.y; va.z += qa.z; va.w += qa.w;
                        vb.x += qb.x; vb.y += qb.y; vb.z += qb.z; vb.w += qb.w;
                    }
                    va.x = fmaxf(va.x, floor_v); va.y = fmaxf(va.y, floor_v); va.z = fmaxf(va.z, floor_v); va.w = fmaxf(va.w, floor_v);
                    vb.x = fmaxf(vb.x, floor_v); vb.y = fmaxf(vb.y, floor_v); vb.z = fmaxf(vb.z, floor_v); vb.w = fmaxf(vb.w, floor_v);
                    float* op_ = out + (outo[m][j] + (unsigned)(cur.n0 + n * 32 + oc * 8));
                    if (p.out_h2) {
                        uint2 ha, la, hb, lb;
                        h2_pack(va, p.act_scale, ha, la, sat_mx);
                        h2_pack(vb, p.act_scale, hb, lb, sat_mx);
                        va = __builtin_bit_cast(float4, make_uint4(ha.x, ha.y, hb.x, hb.y));      // high unit
                        vb = __builtin_bit_cast(float4, make_uint4(la.x, la.y, lb.x, lb.y));      // low unit
                    }
                    if (rowok[m][j]) {
                        *reinterpret_cast<float4*>(op_) = va;
                        *reinterpret_cast<float4*>(op_ + 4) = vb;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        if (p.out_h2) sat_report(p.sat, sat_mx);
    } else {
        // scalar path: output convs of the head (Cout = 142 / 1 / 3 into unaligned NHWC slots)
        unsigned pixo[MT], outo[MT];
        bool rowok[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int mb = wave * MT + m;
            const int oy = cur.ty * C::TH + mb * C::RPB + li / TW, ox = cur.tx * TW + li % TW;
            rowok[m] = oy < p.Ho;
            pixo[m] = rowok[m] ? (unsigned)(oy * p.Wo + ox) : 0u;
            outo[m] = rowok[m] ? (unsigned)(oy * p.out_rs + ox * p.out_cs) : 0u;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int cl = n * 32 + g4 * 8 + lh * 4;
                    const int co = cur.n0 + cl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (co + e < p.Cout && rowok[m]) {
                            float t = fmaf(acc[m][n][g4 * 4 + e], sSc[cl + e], sSc[C::NW + cl + e]);
                            if (res) t += res[pixo[m] * (unsigned)p.res_cs + (unsigned)(co + e)];
                            out[outo[m] + (unsigned)(co + e)] = fmaxf(t, floor_of(co + e));
                        }
                    }
                }
    }
}

// without prefetched residuals
template <int KS, int S, int MT, int NT, int TW, int CK, int NWV = 4>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, const Item& cur, f32x16 (&acc)[MT][NT],
                                              const float* sSc, char* sE, int wave, int li, int lh) {
    EpiRes<MT, NT> none;
    conv_epilogue<KS, S, MT, NT, TW, CK, NWV>(p, cur, acc, sSc, sE, wave, li, lh, none, false);
}

// ---- the direct H2 epilogue (round 4; conv_h2r.hip / conv_h2s.hip) ----------------------------------------------------------
// The phase traces of the register-weight kernels put 25-30 % of a work item into conv_epilogue above: four 16-byte LDS stores, a
// wave barrier and four loads per 32 x 32 block just to re-own the accumulators, the split as ~7 VALU per value, a uniform branch per
// option.  For the one case those kernels almost always run -- H2 output, no residual or an H2 residual -- nothing needs to move
// through LDS: the MFMA leaves lane (li, lh) with channels 8 g4 + 4 lh .. + 3 of pixel li, i.e. HALF of octet g4; the lane splits
// its four values (packed conversions: 2 instructions per value), and one v_permlane32_swap pair trades halves with lane li of the
// other half-wave so that each lane holds one whole 16-byte unit (lower half-wave: the octet's eight high pieces, upper: the low
// pieces) and stores it: 32 contiguous bytes per lane pair, the four g4 steps of a block fill each pixel's 128-byte line.  The
// residual comes in the same way in reverse (one 16-byte unit per lane, two swaps).  Values are formed in the scaled domain
// (scale and shift pre-multiplied by 2^act_shift, as in the fused-block kernels).
__device__ __forceinline__ float h2_add_pieces_clamp(float x, unsigned rh, unsigned rl, int half, float lo_b, float top) {
    float d;                      // min(max(x + piece `half` of rh + piece `half` of rl, lo_b), top)
    if (half == 0)
        asm("v_fma_mix_f32 %0, %1, 1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %0, %2, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_med3_f32 %0, %0, %4, %5" : "=&v"(d) : "v"(rh), "v"(rl), "v"(x), "v"(lo_b), "v"(top));
    else
        asm("v_fma_mix_f32 %0, %1, 1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %0, %2, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_med3_f32 %0, %0, %4, %5" : "=&v"(d) : "v"(rh), "v"(rl), "v"(x), "v"(lo_b), "v"(top));
    return d;
}

template <int KS, int S, int MT, int TW, int NWV>
__device__ __forceinline__ void conv_epilogue_h2direct(const ConvParams& p, const Item& cur, f32x16 (&acc)[MT][1], const float* sSc,
                                                       int wave, int li, int lh) {
    using C = ConvCfg<KS, S, MT, 1, TW, 16, NWV>;
    typedef unsigned u32x2_e __attribute__((ext_vector_type(2)));
    float* out = p.out + (size_t)cur.b * p.out_bs + p.out_co + cur.g * p.out_gs + cur.n0 + lh * 4;
    const float* res = p.res ? p.res + (size_t)cur.b * p.Ho * p.Wo * p.res_cs + p.res_co + cur.g * p.res_gs + cur.n0 + lh * 4 : nullptr;
    const float lo_b = (p.relu && cur.n0 >= p.relu_from) ? 0.f : -H2_MAX;
    unsigned pixo[MT], outo[MT];
    bool rowok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mb = wave * MT + m;
        const int oy = cur.ty * C::TH + mb * C::RPB + li / TW, ox = cur.tx * TW + li % TW;
        rowok[m] = oy < p.Ho;
        pixo[m] = rowok[m] ? (unsigned)(oy * p.Wo + ox) : 0u;
        outo[m] = rowok[m] ? (unsigned)(oy * p.out_rs + ox * p.out_cs) : 0u;
    }
    uint4 ru[MT][4];                                 // residual: this lane's unit (lower half-wave: high pieces, upper: low pieces) of octet g4
    if (res) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) ru[m][g4] = *reinterpret_cast<const uint4*>(res + (pixo[m] * (unsigned)p.res_cs + (unsigned)(g4 * 8)));
    }
    float sat_mx = 0.f;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const int cl = g4 * 8 + lh * 4;
        float4 sc = *reinterpret_cast<const float4*>(sSc + cl), sh = *reinterpret_cast<const float4*>(sSc + 32 + cl);
        sc.x *= p.act_scale; sc.y *= p.act_scale; sc.z *= p.act_scale; sc.w *= p.act_scale;
        sh.x *= p.act_scale; sh.y *= p.act_scale; sh.z *= p.act_scale; sh.w *= p.act_scale;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float v[4] = {fmaf(acc[m][0][g4 * 4 + 0], sc.x, sh.x), fmaf(acc[m][0][g4 * 4 + 1], sc.y, sh.y),
                          fmaf(acc[m][0][g4 * 4 + 2], sc.z, sh.z), fmaf(acc[m][0][g4 * 4 + 3], sc.w, sh.w)};
            if (res) {
                // unit dwords (x, y, z, w) = pieces of channels (0,1) (2,3) (4,5) (6,7); after the swaps: a = high pieces, b = low pieces of
                // THIS lane's channels (4 lh .. 4 lh + 3): [0] the first pair, [1] the second
                const u32x2_e s0 = __builtin_amdgcn_permlane32_swap(ru[m][g4].x, ru[m][g4].z, false, false);
                const u32x2_e s1 = __builtin_amdgcn_permlane32_swap(ru[m][g4].y, ru[m][g4].w, false, false);
                v[0] = h2_add_pieces_clamp(v[0], s0[0], s0[1], 0, lo_b, H2_MAX);
                v[1] = h2_add_pieces_clamp(v[1], s0[0], s0[1], 1, lo_b, H2_MAX);
                v[2] = h2_add_pieces_clamp(v[2], s1[0], s1[1], 0, lo_b, H2_MAX);
                v[3] = h2_add_pieces_clamp(v[3], s1[0], s1[1], 1, lo_b, H2_MAX);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], lo_b, H2_MAX);
            }
            sat_track(sat_mx, v[0], v[1]);            // (after the clamp: |v| == H2_MAX iff it clamped, or hit the limit exactly)
            sat_track(sat_mx, v[2], v[3]);
            const unsigned h0 = h2_high_pair(v[0], v[1]), h1 = h2_high_pair(v[2], v[3]);
            const unsigned l0 = h2_low_pair(h0, v[0], v[1]), l1 = h2_low_pair(h1, v[2], v[3]);
            const u32x2_e a = __builtin_amdgcn_permlane32_swap(h0, l0, false, false);
            const u32x2_e b = __builtin_amdgcn_permlane32_swap(h1, l1, false, false);
            if (rowok[m]) *reinterpret_cast<uint4*>(out + (outo[m] + (unsigned)(g4 * 8))) = make_uint4(a[0], b[0], a[1], b[1]);
        }
    }
    sat_report(p.sat, sat_mx);
}

typedef void (*conv_fn)(ConvParams);
// math: 0 f32 MFMA, 1 bf16x3 (register-staged weights), 2 bf16x3 (LDS-DMA weight rows), 3 / 4 the same two for f16x2,
// 8 f16x2 with register-resident weights (conv_h2r.hip), 9 its stride-2 form on parity planes (conv_h2s.hip),
// 10 the input channels split across the workgroup's waves (conv_h2k.hip: single-image plans), 11 the 1x1 streamed-K GEMM form (conv_h2g.hip).  (5 / 6 / 7 were round 2's LDS-DMA pipeline kernels: never
// faster than 3 / 4 / 8 inside the network, deleted in round 6; their measurements are in profiles/r02_*.)  threads: workgroup size (0 = 256, or 512 for the ping-pong kernels).
// o4: the same kernel compiled for four workgroups per CU (128 VGPRs; named conv_h2o).
struct ConvVariant { int ks, s, mt, nt, tw, ck; conv_fn fn; int lds; int th; int occ; int pp; int math; int threads; int o4; };

// per translation unit: its table of instantiated kernels
ConvVariant* conv_variants_f32(int* n);
ConvVariant* conv_variants_bx3(int* n);
ConvVariant* conv_variants_h2(int* n);
ConvVariant* conv_variants_h2d(int* n);
ConvVariant* conv_variants_h2r(int* n);
ConvVariant* conv_variants_h2s(int* n);
ConvVariant* conv_variants_h2k(int* n);
ConvVariant* conv_variants_h2g(int* n);

}  // namespace romp
