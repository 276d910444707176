/* romp_hip.h -- C ABI of libromp_hip.so, the MI355X (gfx950) implementation of the ROMP
 * inference hot path.
 *
 * Drop-in seams (reference = Arthur151/ROMP, simple_romp/romp):
 *   seam #1  network      ROMPv1.forward                    model.py:470-481
 *                         (the reference already abstracts it as an ONNX session,
 *                          main.py:109: image (B,512,512,3) -> center_maps, params_maps)
 *   seam #2  parsing      parsing_outputs / CenterMap       post_parser.py:27-47,135-146
 *                         + 1.1**scale (main.py:113) + rot6D->axis-angle (utils.py:471-682)
 *   seam #3  SMPL         SMPL.forward / lbs                smpl.py:62-108,111-290
 *   seam #4  projection   batch_orth_proj + to-original-image   utils.py:309-315,
 *                         post_parser.py:81-88, convert_cam_to_3d_trans utils.py:303-307
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host;
 *   - memory is owned by the caller (PyTorch-ROCm tensors); contexts keep their own packed
 *     copies of constants and never free caller memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *     enqueued on it, nothing synchronises unless stated;
 *   - every entry point returns 0 on success or a negative ROMP_E* code; the message for
 *     the calling thread is available from romp_last_error().  No exceptions, no exit().
 *   - contexts are not thread-safe (one per device/stream, like the reference's one module
 *     per process).
 */
#ifndef ROMP_HIP_H
#define ROMP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ROMP_OK          0
#define ROMP_EINVAL     -1   /* bad argument / unsupported shape            */
#define ROMP_EHIP       -2   /* a HIP runtime call failed                   */
#define ROMP_ENOMEM     -3   /* workspace allocation failed                 */
#define ROMP_ECAPACITY  -4   /* batch larger than the context was built for */

#define ROMP_ABI_VERSION 7   /* 7: ROMP_OP_STEM7P;  6: ROMP_OPF_SEAM_DS / unknown flag bits rejected, romp_net_plan_kind, romp_parse_watch, romp_net_sat_counter */

int         romp_abi_version(void);
const char* romp_last_error(void);

/* ------------------------------------------------------------------ seam #1: network */

/* One step of the layer program.  The host (romp_amd/plan.py) lowers the model definition
 * (HRNet-32 + ROMP head: model.py:246-481) into this flat list; the executor runs it.
 * Activations are NHWC float32; `*_buf` are indices into the context's activation arena,
 * or one of the ROMP_BUF_* pseudo buffers. */
#define ROMP_BUF_NONE    (-1)
#define ROMP_BUF_IMAGE   (-2)   /* forward()'s image argument  (B,512,512,3) 0..255      */
#define ROMP_BUF_CENTER  (-3)   /* forward()'s center_maps out (B,64,64)                  */
#define ROMP_BUF_PARAMS  (-4)   /* forward()'s params_maps out (B,64,64,145) NHWC         */

/* Activation tensor formats.  F32: NHWC float32.  H2: the same addressing, but every channel octet (32 bytes) holds eight
 * high fp16 pieces then eight low fp16 pieces of x * 2^act_shift (x = (h1 + h2) * 2^-act_shift up to 2^-22 relative): what the
 * f16x2 conv kernels compute with, stored once by the producer instead of being re-split by every consumer (plan.py decides per
 * tensor; channel strides / offsets of an H2 tensor are multiples of 8). */
#define ROMP_FMT_F32     0
#define ROMP_FMT_H2      1

#define ROMP_OP_STEM      1     /* x/255*2-1 + conv3x3 s2 (Cin=3) + BN + ReLU  (model.py:384-387) */
#define ROMP_OP_CONV      2     /* conv KxK (K=1|3, stride 1|2) + scale/shift (+res) (+ReLU)      */
#define ROMP_OP_FUSESUM   3     /* y = relu(sum_t up_nearest(T_t))   (model.py:233-244)           */
#define ROMP_OP_FORK      4     /* side streams 1..Cin start after everything enqueued so far     */
#define ROMP_OP_JOIN      5     /* the main stream waits for side streams 1..Cin                  */
/* BEV head (bev/model.py:188-215).  For these three kinds `weight/scale/shift` are HOST pointers
 * (small constant tables passed to the kernels by value). */
#define ROMP_OP_BEV_PACK  6     /* in_buf maps_fv (B,128,128,4) + res_buf feats (..,16) -> out_buf (B,128,2560)   */
#define ROMP_OP_BEV_MAPS  7     /* in_buf maps_fv, res_buf bv (B,128,128) -> out_buf center3d, term_buf[0] cam3d;
                                   weight = 64 scale anchors (get_cam3dmap_anchor, bev/model.py:77-87)          */
#define ROMP_OP_STEM7     9     /* (x/255-mean)/std + conv7x7 s2 p3 (Cin=3) + BN + ReLU   (romp/lib/models/resnet_50.py:32-44,56) */
#define ROMP_OP_MAXPOOL   10    /* MaxPool2d(3, 2, 1) on NHWC                             (resnet_50.py:45,56)                   */
#define ROMP_OP_CONV3D    8     /* 3x3x3 conv Cin->Cin (1|3) on (B,C,64,128,128) + scale/shift (+res) (+ReLU)     */
#define ROMP_OP_KSUM      11    /* tail of a split-K conv (single-image latency plans): in_buf holds `groups` float32 partial sums of
                                   Cout channels each per pixel (a grouped ROMP_OP_CONV over input-channel slices wrote them);
                                   y = scale * sum_g partial_g + shift (+res) (+ReLU), H x W = the OUTPUT size                  */
#define ROMP_OP_NOP       12    /* executes nothing (its fields may describe a layer that the next op runs fused)                 */
#define ROMP_OP_BBLOCK32  13    /* a whole 32-channel BasicBlock (model.py:54-83) in one kernel: this op holds the block's SECOND
                                   conv (3x3 s1 32->32 + BN + residual + ReLU; res_buf = the block input x, out_buf = y), the op
                                   right before it -- kind ROMP_OP_NOP, every other field intact -- the FIRST (3x3 s1 32->32 + BN +
                                   ReLU on x).  H2 tensors, f16x2 weights, H and W multiples of 16.  plan.py fuses the pairs.    */
#define ROMP_OP_BBLOCK64  14    /* the same for a 64-channel BasicBlock (csrc/conv_h2c.hip): both ops carry, in weight_aux, their
                                   f16x2 weights repacked per wave for 16-channel MFMA rows (plan.pack_h2_wave16); H a multiple of
                                   8, W of 16                                                                                   */
#define ROMP_OP_SEAM1X1   15    /* two 1x1 convs across a Bottleneck seam of layer1 as one kernel (csrc/conv_h2x.hip,
                                   plan.fuse_bottleneck_seams): the op before it (NOP) is a
                                   64->256 conv + residual + ReLU, this op the 256->64 conv + ReLU reading its output; both
                                   outputs are written; weight_aux = per-group packs.  With ROMP_OPF_SEAM_DS: see the flag     */
#define ROMP_OP_FUSEUP    16    /* a fuse-layer output with its 1x1 up-convs inside (csrc/conv_fup.hip, plan.fuse_up_sums; model.py:186-196,233-244):
                                   y = relu(sum of the terms), terms in op order: first the tensors already at the output's resolution
                                   (term_shift 0), then for term_shift s = 1, 2, ..: nearest_up_2^s(bn(W_s . x_s)) where term_buf is the
                                   SOURCE x_s (Cout << s dense channels at 1 / 2^s resolution).  weight_aux: the f16x2 weights W_s, each
                                   repacked per 16-channel group (plan.pack_h2_wave16), concatenated; scale_h2 / shift: [s][Cout].
                                   H2 tensors throughout; Cout 32 | 64 | 128; H x W = the OUTPUT size                            */
#define ROMP_OP_RECORD    17    /* stream `stream` (0 = main, 1..3 = side) records event number `Cin` here ..                     */
#define ROMP_OP_WAIT      18    /* .. and stream `stream` goes on only when event `Cin` (recorded EARLIER in op order) has fired:
                                   point-to-point edges between the streams of an open FORK .. JOIN region (plan.hr_module, round 4:
                                   an HRNet stage is one region; a fuse output waits for exactly the tensors it sums, the next
                                   module's branch follows its own fuse output in stream order).  Event numbers are unique per
                                   program; with streams off both kinds do nothing (op order is a valid serial order)           */
#define ROMP_OP_STEM2     19    /* HRNet's whole stem as one kernel (csrc/stem2.hip, plan.fuse_stem2; model.py:384-390): the op before it
                                   (NOP, fields intact, in_buf = ROMP_BUF_IMAGE) is the ROMP_OP_STEM 3 -> 64 conv, this op the 3x3 stride-2
                                   64 -> 64 conv + BN + ReLU that read its output; the 64-channel half-resolution tensor between them
                                   is never written.  weight_aux = this conv's per-wave f16x2 pack (ROMP_OPF_WAVE16), H2 output        */
#define ROMP_OP_STEM7P    20    /* ResNet-50's whole stem as one kernel (csrc/stem7p.hip, plan.fuse_stem7p; romp/lib/models/resnet_50.py:32-45,56):
                                   normalisation + conv7x7 s2 p3 3 -> 64 + BN + ReLU + MaxPool2d(3, 2, 1) on the matrix cores.  The op itself
                                   carries everything (H x W of the IMAGE, in_buf = ROMP_BUF_IMAGE, weight [ky][kx][ci][co] float32, scale,
                                   shift; out_* the POOLED tensor, float32 or H2); the op before it is the NOP the ROMP_OP_STEM7 turned into:
                                   the half-resolution 64-channel tensor between conv and pool is never written                         */
/* ROMP_OP_CONV with ksize == 13 is a Conv1d(k=3) along W whose rows are the B batch items. */

/* romp_op.flags */
#define ROMP_OPF_WAVE16     1   /* weight_aux holds the f16x2 weights repacked per wave for 16-channel MFMA rows
                                   (plan.pack_h2_wave16: BBLOCK64, SEAM1X1 and the row-pipelined BBLOCK32 kernel), not a
                                   bf16x3 pack: the fused kernels dispatch on this bit, never on weight_aux != NULL   */
#define ROMP_OPF_SEAM_DS    4   /* SEAM1X1 behind Bottleneck 0 (model.py:289-301): the residual of the 64->256 conv is itself a conv,
                                   bn_d(conv1x1 64->256(x0)) -- the `downsample` branch -- and the seam kernel computes it too: the
                                   op TWO before this one (NOP, fields intact) is that conv, its in_buf x0 the kernel's third input;
                                   its 256-channel output tensor is never written                                        */
#define ROMP_OPF_STEM_VALU  2   /* STEM: take the float32 VALU kernel even for an H2 output (A/B runs, tests; also chosen
                                   when 256 * |w| does not fit the fp16 pieces of the MFMA form)                      */

#define ROMP_OPF_ALL        7   /* every bit defined above: romp_net_create refuses a program with any other bit set */

typedef struct romp_op {
    int32_t kind;
    int32_t in_buf, out_buf, res_buf;
    int32_t H, W;                 /* input spatial size                                   */
    int32_t Cin, Cout;            /* per group                                            */
    int32_t ksize, stride, relu;
    int32_t groups;               /* >1: independent convs sharing one launch (head towers)*/
    int32_t in_cstride, in_coff, in_gstride;     /* channel stride / offset / per-group offset */
    int32_t out_cstride, out_coff, out_gstride;
    int32_t res_cstride, res_coff, res_gstride;
    int32_t cin_pad, cout_pad;    /* padded dims of the packed weight (see plan.py)       */
    int32_t n_terms;              /* FUSESUM: number of terms (<=4)                       */
    int32_t term_buf[4];
    int32_t term_shift[4];        /* log2 of the nearest-upsample factor of each term     */
    int32_t term_cstride[4];
    int32_t stream;               /* 0 = main stream, 1..3 = side stream (between FORK and JOIN):
                                     independent HRNet branches run concurrently (model.py:230-231) */
    int32_t pad_h, pad_w;         /* CONV: zero rows / columns before the first tap; -1 = ksize/2 ('same' padding).
                                     ksize 2 (one output parity of ConvTranspose2d k4 s2 p1): 1 or 0               */
    int32_t out_rstride, out_bstride;  /* CONV: output row / image stride in floats; 0 = dense (Wo*out_cstride,
                                     Ho*Wo*out_cstride).  Sparse strides interleave the parity outputs of a transposed conv */
    int32_t in_fmt, out_fmt, res_fmt;   /* CONV / STEM / FUSESUM (out_fmt): ROMP_FMT_F32 or ROMP_FMT_H2 */
    int32_t term_fmt[4];          /* FUSESUM: format of each term */
    int32_t act_shift;            /* CONV, f16x2 kernels: activations are multiplied by 2^act_shift before they are split
                                     into fp16 pieces (keeps the low piece out of the fp16 subnormal range; |x| must stay
                                     below 65504 / 2^act_shift).  scale_h2 carries the inverse.                       */
    int32_t flags;                /* ROMP_OPF_* bits                                                                  */
    int32_t relu_from;            /* CONV with relu != 0: ReLU applies to output channels >= relu_from only (0: all).
                                     Sibling convs that read one tensor run as ONE conv with concatenated output channels
                                     (plan.merge_sibling_convs): the fuse layers' stride-2 chains (model.py:198-221) mix a
                                     last conv (no ReLU) with first convs (ReLU).  A multiple of 32.                  */
    int32_t term_coff[4];         /* FUSESUM: channel offset of each term inside its buffer (a slice of a merged conv's output) */
    const float* weight;          /* packed [group][chunk][tap][cin/4][cout_pad][4]       */
    const float* scale;           /* [group][cout_pad]  gamma/sqrt(var+eps)  (or 1)       */
    const float* shift;           /* [group][cout_pad]  beta-mean*scale (+scale*bias)     */
    const void*  weight_aux;      /* optional: the same weights split into 3 bf16 pieces,
                                     [group][tap][cin_pad/16][piece 3][kg 2][cout_pad][8] (bf16x3 kernels) */
    const void*  weight_h2;       /* optional: the weights, multiplied by a per-group power of two 2^ws, split into
                                     2 fp16 pieces, [group][tap][cin_pad/16][piece 2][kg 2][cout_pad][8] (f16x2 kernels) */
    const float* scale_h2;        /* [group][cout_pad]  scale * 2^-(ws + act_shift): the f16x2 kernels' epilogue scale */
} romp_op;

typedef struct romp_net romp_net;

/* buf_floats_per_image[i] = size of arena buffer i for ONE image, in floats. */
int  romp_net_create(romp_net** out, const romp_op* ops_host, int n_ops,
                     const int64_t* buf_floats_per_image_host, int n_bufs, int max_batch);
/* image (B,512,512,3) float 0..255 -> center_maps (B,64,64), params_maps (B,64,64,145).
 * Replaces ort_session.run / self.model(...) at main.py:109-112. */
int  romp_net_forward(romp_net* net, const float* image_nhwc, int B,
                      float* center_maps, float* params_maps_nhwc, void* stream);
/* Debug/inspection: copy arena buffer `buf` (B images) to `dst` (device). */
int  romp_net_read_buffer(romp_net* net, int buf, int B, float* dst, int64_t n_floats, void* stream);
/* Initialise arena buffer `buf` (all max_batch images) from `src` (device); used once for the
 * constant CoordConv channels of the head input (model.py:473). */
int  romp_net_write_buffer(romp_net* net, int buf, const float* src, int64_t n_floats, void* stream);
/* 0: tuned MFMA kernels (default)   1: naive direct-conv kernels (bring-up cross-check) */
int  romp_net_set_mode(romp_net* net, int mode);
/* 1 (default): independent HRNet branches (FORK/JOIN regions) run on side HIP streams. */
int  romp_net_set_streams(romp_net* net, int enable);
/* 1: turn the layer program into a hipGraph per (B, pointers) and replay it (built node by node from single-op captures with the
 * program's own dependencies: csrc/net.hip build_graph).  Needs a non-default stream. */
int  romp_net_set_graph(romp_net* net, int enable);
/* Measure every valid kernel variant of every conv layer at batch B (HIP events on `stream`,
 * `iters` timed runs each) and use the fastest from now on for that batch size. */
int  romp_net_autotune(romp_net* net, int B, int iters, void* stream);
/* Batch lanes: lanes == 2 runs a forward of an even batch B as two independent half-batch sequences on
 * two HIP streams (lane 0 on the caller's stream; each with its own branch side streams), every conv
 * limited to `wg_cap` workgroups per CU (0: no limit) so that kernels of the two lanes share each CU
 * in different phases.  Kernel variants are then looked up for batch B/2.  The per-image float counts
 * of the caller's image / center / params tensors give the lane offsets.  lanes == 1: off. */
int  romp_net_set_split(romp_net* net, int lanes, int wg_cap, int64_t image_floats, int64_t center_floats,
                        int64_t params_floats);
/* Variant index chosen by romp_net_autotune for op `op_index` at batch B (-1: heuristic). */
int  romp_net_tuned_variant(romp_net* net, int B, int op_index);
/* Install a variant table for batch B without measuring (e.g. one saved from an earlier autotune):
 * variants[n_ops], -1 = heuristic.  An index that is not valid for its op is ROMP_EINVAL. */
int  romp_net_set_tuned(romp_net* net, int B, const int32_t* variants, int n_ops);

/* A net from a plan file written by romp_amd/export.py (the lowered program with its packed constants, buffer
 * initialisers and measured variant tables): what the reference's exported ROMP.onnx is to its ort_session
 * (model.py:484-497, main.py:89,109) -- the model is converted once, the inference host needs only this library.
 * The net owns the constants.  romp_net_plan_info: the input side length and the per-image sizes of the two outputs
 * recorded in the file (0 for a net that was not loaded from a file), and the op count. */
int  romp_net_load(romp_net** out, const char* path, int max_batch);
int  romp_net_plan_info(romp_net* net, int32_t* input_size, int64_t* center_floats, int64_t* params_floats, int32_t* n_ops);
/* The KIND of plan the file holds (plan-file version 2 header): *split_k_items > 0 = a single-image plan (the layers with few
 * pixels were lowered with their input channels split until they had that many work items: meant for max_batch <= 2, correct but
 * slow beyond), 0 = a batch plan, or a net that was not loaded from a file. */
int  romp_net_plan_kind(romp_net* net, int32_t* split_k_items);
/* Time a forward per op (HIP events on `stream` around every op, ops serialised on that stream): ms_out_host[n_ops] = the
 * median of `iters` passes. */
int  romp_net_profile(romp_net* net, const float* image_nhwc, int B, float* center_maps,
                      float* params_maps_nhwc, void* stream, float* ms_out_host, int iters);
/* Activation range scan (f16x2 range safety): runs the program op by op on `stream` and reports, for every op that writes an
 * arena buffer, max|x| (maxabs_out_host[n_ops]; H2 tensors decoded) and the number of non-finite values
 * (nonfinite_out_host[n_ops]) of the REGION the op wrote (its pixels x channel slice of B images) right after the op; 0 for ops
 * without an arena output.  saturated_out_host[n_ops] (may be NULL): saturation events each op reported (see romp_net_saturated;
 * the fused-block kernels run their counting builds during a scan).  The host (plan.assign_formats) keeps tensors whose range
 * does not fit the fp16 pieces of ROMP_FMT_H2 in float32 and their consumers on the f32 / bf16x3 kernels; the kernels themselves
 * saturate at +-65504 instead of producing inf / NaN pieces.  Synchronises. */
int  romp_net_range_scan(romp_net* net, const float* image_nhwc, int B, float* center_maps, float* params_maps_nhwc,
                         void* stream, float* maxabs_out_host, int32_t* nonfinite_out_host, int32_t* saturated_out_host);
/* Saturation is observable: every kernel that splits values into the fp16 pieces of ROMP_FMT_H2 clamps at +-65504 / 2^act_shift
 * and reports a clamp into the net's device counter (one increment per wave and work item).  *count_host = events since
 * romp_net_create or the last call with reset != 0; 0 for a net inside its calibrated range.  Every kernel counts in every build
 * (ABI 6: the two register-resident fused BasicBlock kernels too; romp_net_set_sat_check, which switched their counting builds on
 * in ABI 5, is kept and changes nothing).  romp_net_saturated synchronises `stream`. */
int  romp_net_saturated(romp_net* net, int64_t* count_host, int reset, void* stream);
int  romp_net_set_sat_check(romp_net* net, int enable);
/* Device address of that counter (an int32, cumulative like romp_net_saturated without reset): what romp_parse_watch's `watch`
 * argument is for -- the default-on range guard of the Python API reads it with the detection count, for free (main.py). */
const int32_t* romp_net_sat_counter(romp_net* net);
void romp_net_destroy(romp_net* net);

/* Stand-alone conv launcher (tests / microbenchmarks of one layer).  variant < 0: heuristic. */
int  romp_conv_forward(const romp_op* op_host, const float* in, const float* res, float* out,
                       int B, int mode, int variant, void* stream);
int  romp_conv_num_variants(void);
/* Kernel variants of one family in THIS build (ConvVariant.math: 0 f32 MFMA, 1 / 2 bf16x3, 3 / 4 f16x2 generic, 8 conv_h2r, 9 conv_h2s,
 * 10 conv_h2k, 11 conv_h2g).  The bf16x3 family is optional (ROMP_WITH_BX3=1 python -m romp_amd.build): 0 means `--conv_math bf16x3`
 * cannot be served by this library. */
int  romp_conv_family_variants(int math);
/* Developer aid: with env ROMP_CONV_TRACE=1 the split-precision conv kernels stamp their phases (s_memtime) per wave;
 * copies the stamps of the most recent launch (64 words per wave: count, then (time << 8 | event)) to the host and
 * returns the number of words, or < 0.  Synchronises the device. */
int  romp_conv_trace_read(unsigned long long* dst_host, int max_words);

/* Name of kernel variant `variant` (or, if < 0, of the heuristic choice at batch B) for `op`;
 * ROMP_EINVAL if that variant cannot run this op. */
int  romp_conv_describe(const romp_op* op_host, int B, int variant, char* out_host, int n);

/* ------------------------------------------------------------------ seam #2: parsing */

/* parsing_outputs (post_parser.py:135-146) on device.  Capacity-bounded: row arrays hold
 * B*max_person rows; *count_host receives N (this call synchronises the stream once, like
 * the reference's torch.where at post_parser.py:45).  Rows are batch-major and
 * score-descending inside an image; ties broken by lower flat index.
 *   center_maps (B,64,64)      params_maps (B,64,64,145) NHWC, scale channel NOT yet 1.1**
 *   batch_ids/flat_inds int32 (N)   scores (N)   params_pred (N,145) (scale already 1.1**s)
 *   cam (N,3)  thetas (N,72)  betas (N,10)  center_preds int32 (N,2)
 * count_host == NULL: asynchronous form (no copy, no synchronisation): image b's count stays on the device in
 * workspace[b*(2*max_person+2) + 2*max_person]; rows past the count are left untouched.  The single-image path uses it to
 * enqueue the whole post-processing for `max_person` rows while the network still runs, with ONE sync at the end. */
int  romp_parse(const float* center_maps, const float* params_maps_nhwc, int B,
                float conf_thresh, int max_person, int32_t* count_host,
                int32_t* batch_ids, int32_t* flat_inds, float* scores, float* params_pred,
                float* cam, float* thetas, float* betas, int32_t* center_preds,
                int32_t* workspace /* B*(2*max_person+2) int32 */, void* stream);
/* romp_parse with a WATCHED device word riding on the count read-back: the packing kernel copies *watch (an int32 on the device;
 * the API passes romp_net_sat_counter(net)) into workspace[B*(2*max_person+2)], and with count_host != NULL it comes back in
 * *watch_host inside the SAME device-to-host copy and synchronisation as the counts (count_host == NULL: it stays in the
 * workspace for the caller's own download).  The reference's network is float32 and has no range to leave
 * (simple_romp/romp/main.py:106-115); the f16x2 kernels clamp beyond 65504 / 2^act_shift, and this is how every forward of the API
 * learns -- without a second round trip -- whether one did, so that it can re-run the call on the exact-f32 program.
 * workspace: B*(2*max_person+2) + 2 int32.  watch == NULL: exactly romp_parse. */
int  romp_parse_watch(const float* center_maps, const float* params_maps_nhwc, int B,
                      float conf_thresh, int max_person, int32_t* count_host,
                      int32_t* batch_ids, int32_t* flat_inds, float* scores, float* params_pred,
                      float* cam, float* thetas, float* betas, int32_t* center_preds,
                      int32_t* workspace /* B*(2*max_person+2) + 2 int32 */, void* stream,
                      const int32_t* watch, int32_t* watch_host);
/* rot6D_to_angular (utils.py:471-475): x6 (n,6) -> aa (n,3) */
int  romp_rot6d_to_aa(const float* x6, int n, float* aa, void* stream);

/* ------------------------------------------------------------------ seam #2b: BEV 3-D parsing */

/* CenterMap3D.parse_3dcentermap (bev/post_parser.py:44-66): MaxPool3d(5) NMS + top-K over
 * center_maps_3d (B,64,128,128).  conf_thresh must be > 0.  Rows batch-major, score-descending,
 * ties by lower flat zyx index.  workspace: romp_bev_workspace_ints(B, max_person) int32.
 * Synchronises the stream once (count). */
int  romp_bev_workspace_ints(int B, int max_person);
int  romp_bev_parse(const float* center_maps_3d, int B, float conf_thresh, int max_person, int32_t* count_host,
                    int32_t* batch_ids, int32_t* czyx /* (N,3) */, float* confs, int32_t* workspace, void* stream);
/* cam gather (bev/model.py:242) + mesh_parameter_regression (:225-230) + pack_params_dict
 * (bev/post_parser.py:240-253) + denormalize_cam_params_to_trans (:114-128) for N detections.
 * cam_maps_3d (B,3,64,128,128); fv_features (B,128,128,feat_cstride>=128) NHWC; MLP weights are
 * TRANSPOSED ([in][out]): w1t (128,512), w2t (512,512), w3t (512,143); emb (128,128).
 * -> params_pred (N,146), cam_czyx int32 (N,3), cam (N,3), thetas (N,72), betas (N,11), cam_trans (N,3) */
int  romp_bev_regress(const float* cam_maps_3d, const float* fv_features, int feat_cstride, int N,
                      const int32_t* batch_ids, const int32_t* czyx, const float* anchors_host, const float* emb,
                      const float* w1t, const float* b1, const float* w2t, const float* b2, const float* w3t,
                      const float* b3, float* params_pred, int32_t* cam_czyx, float* cam, float* thetas, float* betas,
                      float* cam_trans, void* stream);
/* Device address of arena buffer `buf` (e.g. the BEV front-view feature map). */
float* romp_net_buffer_ptr(romp_net* net, int buf);

/* ------------------------------------------------------------------ seam #3: SMPL */

typedef struct smpl_ctx smpl_ctx;
/* Built from the tensors of the packed SMPL file (schema: pack_smpl_info.py:70-111).
 * parents_host: 24 int64 (kintree_table), extra_idx_host: 21 int64. */
int  smpl_ctx_create(smpl_ctx** out, const float* v_template, const float* shapedirs, int n_betas,
                     const float* posedirs, const float* J_regressor, const float* lbs_weights,
                     const int64_t* parents_host, const float* J_regressor_extra9,
                     const float* J_regressor_h36m17, const int64_t* extra_idx_host,
                     int max_persons, void* stream);
/* SMPL.forward (smpl.py:62-108): betas (N,n_betas), thetas (N,72) ->
 * verts (N,6890,3), joints (N,71,3). */
int  smpl_forward(smpl_ctx* ctx, const float* betas, int n_betas, const float* thetas, int N,
                  int root_align, float* verts, float* joints, void* stream);
void smpl_ctx_destroy(smpl_ctx* ctx);

/* ------------------------------------------------------------------ seam #4: projection */

/* batch_orth_proj + convert_proejection_from_input_to_orgimg + convert_cam_to_3d_trans.
 * joints (N,J,3), cam (N,3), pad_info_host[6] = top,bottom,left,right,h,w  ->
 * pj2d (N,J,2) normalised, pj2d_org (N,J,2) original-image pixels, cam_trans (N,3). */
int  romp_project(const float* joints, int N, int J, const float* cam, const float* pad_info_host,
                  float* pj2d, float* pj2d_org, float* cam_trans, void* stream);

/* convert_cam_to_3d_trans (utils.py:303-307): cam (N,3) = (s, tx, ty) -> trans (N,3) = (tx/s, ty/s, 1/s) * weight. */
int  romp_cam_to_trans(const float* cam, int N, float weight, float* trans, void* stream);

/* convert_cam_to_3d_trans2 (post_parser.py:96-101) with the reference's linear least-squares estimator
 * (estimate_translation_np, utils.py:347-389: its path when OpenCV's PnP is not available): joints (N,J,3), the first K
 * of them against their normalised orthographic projections pj2d (N,J,2) mapped to (pj2d + 1) * img_size / 2 pixels
 * -> trans (N,3); (-1,-1,-1) for a singular system. */
int  romp_estimate_translation(const float* joints, int N, int J, int K, const float* pj2d, float focal_length,
                               float img_size, float* trans, void* stream);

/* Vertices of the meshes for rendering: verts_camed = batch_orth_proj(verts, cam, '3d', keep_dim) and its
 * original-image version (post_parser.py:81-88,108,113).  verts (N,V,3); verts_camed may be NULL. */
int  romp_project_verts(const float* verts, int N, int V, const float* cam, const float* pad_info_host,
                        float* verts_camed, float* verts_camed_org, void* stream);

/* BEV's rendering vertices (bev/post_parser.py:144-151): perspective_projection(verts, translation = cam_trans, focal 443.4,
 * normalised by 256) with the untranslated vertex z appended, mapped to original-image pixels.  verts (N,V,3), cam_trans (N,3). */
int  romp_bev_project_verts(const float* verts, int N, int V, const float* cam_trans, const float* pad_info_host,
                            float* verts_camed_org, void* stream);

/* ------------------------------------------------------------------ callers either side */

/* img_preprocess (utils.py:16-30) on device: BGR uint8 (H,W,3) -> RGB float32 (S,S,3) 0..255, centred zero
 * pad to square + cv::resize(INTER_CUBIC) in OpenCV's fixed-point arithmetic (11-bit coefficients, replicated
 * border, saturate).  pad_info_host[6] receives top,bottom,left,right,h,w. */
int  romp_preprocess(const unsigned char* bgr_u8, int H, int W, float* out_rgb_f32, int out_size,
                     float* pad_info_host, void* stream);
/* The same for B frames of one size, (B,H,W,3) -> (B,S,S,3), in one launch (video / batch mode). */
int  romp_preprocess_batch(const unsigned char* bgr_u8, int B, int H, int W, float* out_rgb_f32, int out_size,
                           float* pad_info_host, void* stream);
/* BEV per-image post-processing (bev/post_parser.py:68-136,167-222): camera translation, perspective
 * projection (normalised and original-image pixels), projection-based duplicate suppression, outlier
 * removal.  Persons of image b are rows offsets[b]..offsets[b+1]-1 (offsets: B+1 int32, device);
 * pad_info: (B,6) device.  keep[row] = 1 if the person survives. */
int  romp_bev_postprocess(const float* joints /* (N,71,3) */, const float* cam /* (N,3) */, const int32_t* offsets,
                          int B, const float* pad_info, float nms_thresh, float relative_scale_thresh,
                          float* pj2d, float* pj2d_org, float* cam_trans, int32_t* keep, void* stream);

/* ---- Sim3DR mesh renderer -------------------------------------------------
 * Replaces the Cython/C++ extension `Sim3DR_Cython` (simple_romp/vis_human/sim3drender/lib/rasterize.pyx,
 * rasterize_kernel.cpp) and the per-vertex lighting of renderer.py:64-110.  Bit-identical images.
 * All pointers are device pointers unless named *_host. */
/* _get_normal (rasterize_kernel.cpp:171-229).  adj_off[nver+1] / adj_ent[3*ntri]: for each vertex the
 * ascending list of flattened corner indices 3*t+k that reference it (built once per topology). */
int  romp_sim3dr_normals(const float* verts, const int32_t* tris, const int32_t* adj_off, const int32_t* adj_ent,
                         int nver, float* normals, void* stream);
/* Sim3DR.render lighting (renderer.py:77-110).  cfg_host[14] = ambient rgb (intensity_ambient*color as
 * float32), intensity_directional, intensity_specular (0 = off), color_directional rgb, light_pos xyz,
 * view_pos xyz; specular_exp is 1. */
int  romp_sim3dr_light(const float* verts, const float* normals, int nver, const float* cfg_host, float* light,
                       void* stream);
/* _rasterize (rasterize_kernel.cpp:233-300) with alpha = 1, in place on image (h,w,c) uint8; vertices in
 * pixel coordinates, larger z = nearer; keys: h*w 64-bit words of scratch. */
int  romp_sim3dr_rasterize(unsigned char* image, const float* verts, const int32_t* tris, const float* colors,
                           int ntri, int h, int w, int c, int reverse, unsigned long long* keys, void* stream);

/* ---- temporal smoothing ---------------------------------------------------
 * OneEuro filters of smooth_results (simple_romp/romp/utils.py:188-269) for N tracked persons, in place on
 * thetas (N,72), betas (N,n_betas), cam (N,3).  state: caller-owned device buffer of
 * romp_oneeuro_state_floats(n_betas) floats per track slot, zeroed when a slot is (re)assigned;
 * slots[N]: the slot of each row (device).  smooth_coeff: --smooth_coeff (mincutoff of the pose filters). */
int  romp_oneeuro_state_floats(int n_betas);
int  romp_oneeuro_smooth(float* state, const int32_t* slots, int N, int n_betas, float smooth_coeff,
                         float* thetas, float* betas, float* cam, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ROMP_HIP_H */
