"""Lower a ROMP state_dict (HRNet-32 + ROMP head) to the layer program of libromp_hip.so.

Host-side mirror of the reference's model definition (simple_romp/romp/model.py):
``HigherResolutionNet`` :246-417 (stem :338-344, layer1 :345, transitions :254-287, stages
:305-334 / ``HighResolutionModule`` :129-244) and ``ROMPv1`` head :427-481.  The state_dict
key layout (1 851 entries) is the weight interface; this module folds
every inference BatchNorm into a per-channel (scale, shift), re-packs conv weights for the
MFMA implicit-GEMM kernels and assigns NHWC activation buffers with liveness-based reuse.

What is NOT here: any arithmetic of the hot path.  The program is executed by
``csrc/net.hip``; this file only decides *what* is launched on *which* buffers.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from .lib import (OP_RECORD, OP_WAIT, OPF_WAVE16, OPF_STEM_VALU, OPF_SEAM_DS, OP_FUSEUP, OP_NOP, OP_BBLOCK32, OP_BBLOCK64, OP_SEAM1X1, BUF_CENTER, BUF_IMAGE, BUF_NONE, BUF_PARAMS, FMT_F32, FMT_H2, OP_BEV_MAPS, OP_CONV, OP_FORK,
                  OP_FUSESUM, OP_JOIN, OP_KSUM, OP_STEM, OP_STEM2, OP_STEM7, OP_MAXPOOL, OP_STEM7P, RompOp)

BN_EPS = 1e-5
HEAD_IN_CH = 48          # 32 backbone + 2 CoordConv channels, zero-padded to a multiple of 16 (the f16x2 kernels' channel chunk:
                         # with 40 the head's first conv was the one layer left on the f32 kernels)


@dataclass
class Act:
    """An NHWC activation living in arena buffer `buf` (channel stride may exceed C)."""
    buf: int
    C: int
    H: int
    W: int
    cstride: int
    coff: int = 0


def _round_up(x, m):
    return (x + m - 1) // m * m


def fold_bn(sd, bn, cout, bias=None):
    """scale = gamma/sqrt(var+eps), shift = beta - mean*scale (+ scale*bias); float64 -> float32."""
    if bn is None:
        scale = torch.ones(cout, dtype=torch.float64)
        shift = torch.zeros(cout, dtype=torch.float64)
    else:
        g, b = sd[bn + '.weight'].double(), sd[bn + '.bias'].double()
        m, v = sd[bn + '.running_mean'].double(), sd[bn + '.running_var'].double()
        scale = g / torch.sqrt(v + BN_EPS)
        shift = b - m * scale
    if bias is not None:
        shift = shift + scale * bias.double()
    return scale.float(), shift.float()


def conv_pads(cin, cout, ksize):
    """Padded (cin, cout) of the packed weight: cin to the kernel's channel-chunk size,
    cout to the 32-wide MFMA N-block (64 when the layer has >= 64 channels)."""
    if ksize == 1:
        ck = 32 if cin % 32 == 0 else 16
    else:                                   # 3 (3x3), 2 (2x2) and 13 (1x3, Conv1d)
        ck = 16 if cin % 16 == 0 else 8
    # (a merged sibling conv may have 96 output channels: three whole 32-wide blocks, no padding -- padded channels would cost the
    # vector epilogue and with it the H2 output)
    return _round_up(cin, ck), _round_up(cout, 64 if (cout >= 64 and cout % 32) or cout % 64 == 0 else 32)


def pack_conv_weight(w, cin_pad, cout_pad):
    """OIHW (cout,cin,kh,kw) -> [tap][cin_pad/4][cout_pad][4] (zero padded); Conv1d weights
    (cout,cin,3) are treated as kh=1, kw=3."""
    if w.dim() == 3:
        w = w.unsqueeze(2)
    cout, cin, kh, kw = w.shape
    t = w.permute(2, 3, 1, 0).reshape(kh * kw, cin, cout)
    full = torch.zeros(kh * kw, cin_pad, cout_pad, dtype=torch.float32)
    full[:, :cin, :cout] = t
    return full.reshape(kh * kw, cin_pad // 4, 4, cout_pad).permute(0, 1, 3, 2).contiguous()


def pack_conv_weight_bx3(w, cin_pad, cout_pad):
    """Weights split into three bf16 pieces (w = w1 + w2 + w3 exactly, round-to-nearest-even like the
    device-side activation split) for the bf16x3 conv kernels:
    -> int16 tensor [tap][cin_pad/16][piece 3][kg 2][cout_pad][8]."""
    if w.dim() == 3:
        w = w.unsqueeze(2)
    cout, cin, kh, kw = w.shape
    full = torch.zeros(kh * kw, cin_pad, cout_pad, dtype=torch.float32)
    full[:, :cin, :cout] = w.permute(2, 3, 1, 0).reshape(kh * kw, cin, cout)
    p1 = full.bfloat16()
    r1 = full - p1.float()
    p2 = r1.bfloat16()
    p3 = (r1 - p2.float()).bfloat16()
    t = torch.stack([p1, p2, p3]).reshape(3, kh * kw, cin_pad // 16, 2, 8, cout_pad)
    return t.permute(1, 2, 0, 3, 5, 4).contiguous().view(torch.int16)


def pack_conv_weight_h2(w, cin_pad, cout_pad):
    """Weights for the f16x2 conv kernels: multiplied by a power of two 2^ws that puts max|w| in (128, 256] (exact; keeps
    the low piece of every weight that matters out of the fp16 subnormal range), then split into two fp16 pieces
    (w*2^ws = h1 + h2 up to 2^-22 relative, round-to-nearest-even like the device-side activation split).
    -> (int16 tensor [tap][cin_pad/16][piece 2][kg 2][cout_pad][8], ws)."""
    if w.dim() == 3:
        w = w.unsqueeze(2)
    cout, cin, kh, kw = w.shape
    full = torch.zeros(kh * kw, cin_pad, cout_pad, dtype=torch.float32)
    full[:, :cin, :cout] = w.permute(2, 3, 1, 0).reshape(kh * kw, cin, cout)
    m = float(full.abs().max())
    ws = 0 if m == 0.0 or not math.isfinite(m) else int(math.floor(math.log2(256.0 / m)))
    ws = max(-24, min(24, ws))
    full = full * (2.0 ** ws)
    p1 = full.half()
    p2 = (full - p1.float()).half()
    t = torch.stack([p1, p2]).reshape(2, kh * kw, cin_pad // 16, 2, 8, cout_pad)
    return t.permute(1, 2, 0, 3, 5, 4).contiguous().view(torch.int16), ws


ACT_SHIFT = 4            # f16x2 kernels: activations are split as fp16 pieces of 16*x (|x| < 4094)
# Range of max|x| over a tensor for which the fp16 pieces of x * 2^ACT_SHIFT are as good as float32 (DESIGN.md section 4,
# "range safety"): the high piece leaves fp16 at 65504 = 4094 * 16 (the kernels saturate there); H2_HI keeps a factor CALIB_MARGIN
# of head-room over the CALIBRATION data (round 3: 2; round 4: 4 -- the default calibration frames are synthetic, and what still
# escapes is counted: RompNet.saturated); below H2_LO most high pieces are fp16 subnormals and the pair holds fewer than 22 bits
# relative to the tensor's largest values.  assign_formats keeps tensors outside it in float32 and takes the f16x2 weights away
# from their consumers (they run on the f32 / bf16x3 kernels).
CALIB_MARGIN = 4.0
H2_HI = 2.0 ** (16 - ACT_SHIFT) / CALIB_MARGIN
H2_LO = 2.0 ** (-2 - ACT_SHIFT)


def h2_range_ok(maxabs, margin=CALIB_MARGIN):
    """May a tensor whose largest calibrated magnitude is `maxabs` (None: not measured) be split into fp16 pieces of
    x * 2^ACT_SHIFT, with a factor `margin` of head-room below the fp16 limit?"""
    return maxabs is None or maxabs == 0.0 or (H2_LO <= maxabs < 2.0 ** (16 - ACT_SHIFT) / margin)


def set_conv_math(P, conv_math):
    """Which split-precision weight packs a Program carries (the kernels are then offered to the autotuner):
    False / 'f32': none; True / 'bf16x3': 3 bf16 pieces; 'f16x2': 2 fp16 pieces; 'all': both."""
    if conv_math is True:
        conv_math = 'bf16x3'
    if conv_math in (False, None):
        conv_math = 'f32'
    assert conv_math in ('f32', 'bf16x3', 'f16x2', 'all'), conv_math
    P.conv_math = conv_math
    # the bf16x3 family is an optional part of the library (round 6: no committed table selects it): 'bf16x3' needs a build that has it,
    # 'all' means "every split family this build offers"
    from . import lib as _L
    if conv_math == 'bf16x3' and not _L.has_bf16x3():
        raise _L.RompHipError("conv_math='bf16x3': this libromp_hip.so was built without the bf16x3 kernel family "
                              "(ROMP_WITH_BX3=1 python -m romp_amd.build --force); the default split arithmetic is 'f16x2'")
    P.bf16x3 = conv_math == 'bf16x3' or (conv_math == 'all' and _L.has_bf16x3())
    P.f16x2 = conv_math in ('f16x2', 'all')


def encode_h2(x, act_shift=ACT_SHIFT):
    """float32 (..., C) with C % 8 == 0 -> the H2 format (include/romp_hip.h ROMP_FMT_H2) as a float32-typed tensor of the same
    shape: per channel octet eight high fp16 pieces then eight low pieces of x * 2^act_shift.  Host-side helper for constants
    and tests (the kernels produce / consume the format themselves)."""
    assert x.shape[-1] % 8 == 0
    xs = x.float() * (2.0 ** act_shift)
    h1 = xs.half()
    h2 = (xs - h1.float()).half()
    o = torch.stack([h1.reshape(*x.shape[:-1], -1, 8), h2.reshape(*x.shape[:-1], -1, 8)], -2)     # (..., C/8, 2, 8)
    return o.contiguous().view(torch.float32).reshape(x.shape)


def decode_h2(t, act_shift=ACT_SHIFT):
    """Inverse of encode_h2 (to 22 significant bits): float32-typed H2 tensor (..., C) -> float32 values."""
    h = t.contiguous().view(torch.float16).reshape(*t.shape[:-1], -1, 2, 8).float()
    return ((h[..., 0, :] + h[..., 1, :]) * (2.0 ** -act_shift)).reshape(t.shape)


def assign_formats(P):
    """Decide per activation tensor whether it lives in HBM as float32 or already split (H2), and write the choice into the
    ops.  A tensor (one live range of an arena buffer) is H2 iff conv_math offers the f16x2 kernels, every producer can write
    H2 (vector-epilogue convs, stem, fuse sums) and every consumer reads it through an f16x2 kernel or a fuse sum; tensors
    touched by anything else (max-pool, BEV head pieces, Conv1d, the host through romp_net_buffer_ptr) stay float32."""
    P.buf_fmt = {}                                           # buffer -> format of its LAST live range (for read_buffer)
    P.range_fallback = []                                    # (op index, name, max|x|): ops taken off the f16x2 kernels
    if not getattr(P, 'f16x2', False):
        return
    gens, cur = [], {}
    maxabs = getattr(P, 'op_maxabs', None)                   # per op: max|x| of its output (RompNet.calibrate), or None

    def gen_for_write(buf):
        g = cur.get(buf)
        if g is None or gens[g]['read']:
            gens.append(dict(buf=buf, ok=buf not in getattr(P, 'exported_bufs', ()), read=False, uses=[]))
            cur[buf] = g = len(gens) - 1
        return gens[g]

    def gen_for_read(buf):
        if buf not in cur:                                   # initialised from outside the program
            gens.append(dict(buf=buf, ok=False, read=False, uses=[]))
            cur[buf] = len(gens) - 1
        g = gens[cur[buf]]
        g['read'] = True
        return g

    def oct_ok(*v):
        return all((int(x) & 7) == 0 for x in v)

    for i, op in enumerate(P.ops):
        if op.kind == OP_CONV:
            h2_kernel = bool(op.weight_h2) and op.ksize != 13
            if op.in_buf >= 0:
                g = gen_for_read(op.in_buf)
                g['uses'].append((i, 'in'))
                g['ok'] &= h2_kernel and oct_ok(op.in_cstride, op.in_coff, op.in_gstride)
            vec = op.Cout == op.cout_pad and oct_ok(op.Cout, op.out_cstride, op.out_coff, op.out_gstride) and op.ksize != 13
            if op.res_buf >= 0:
                g = gen_for_read(op.res_buf)
                g['uses'].append((i, 'res'))
                g['ok'] &= vec and oct_ok(op.res_cstride, op.res_coff, op.res_gstride)
            if op.out_buf >= 0:
                g = gen_for_write(op.out_buf)
                g['uses'].append((i, 'out'))
                g['ok'] &= vec
        elif op.kind == OP_FUSESUM:
            for k in range(op.n_terms):
                g = gen_for_read(op.term_buf[k])
                g['uses'].append((i, ('term', k)))
                g['ok'] &= oct_ok(op.term_cstride[k], op.term_coff[k])
            g = gen_for_write(op.out_buf)
            g['uses'].append((i, 'out'))
            g['ok'] &= oct_ok(op.out_cstride, op.out_coff)
        elif op.kind == OP_STEM:
            g = gen_for_write(op.out_buf)
            g['uses'].append((i, 'out'))
            g['ok'] &= oct_ok(op.out_cstride, op.out_coff)
        elif op.kind == OP_MAXPOOL and _stem7p_pair(P, i):   # ResNet-50's stem + pool, about to become one MFMA kernel (fuse_stem7p): it writes either format
            gen_for_read(op.in_buf)['ok'] = False
            g = gen_for_write(op.out_buf)
            g['uses'].append((i, 'out'))
            g['ok'] &= oct_ok(op.out_cstride, op.out_coff)
        elif op.kind == OP_KSUM:                             # float32 partial sums in, either format for the residual and the result
            gen_for_read(op.in_buf)['ok'] = False
            if op.res_buf >= 0:
                g = gen_for_read(op.res_buf)
                g['uses'].append((i, 'res'))
                g['ok'] &= oct_ok(op.res_cstride, op.res_coff)
            g = gen_for_write(op.out_buf)
            g['uses'].append((i, 'out'))
            g['ok'] &= oct_ok(op.out_cstride, op.out_coff)
        elif op.kind in (OP_FORK, OP_JOIN, OP_RECORD, OP_WAIT):
            continue
        else:                                                # any other op: its tensors stay float32
            for b in (op.in_buf, op.res_buf):
                if b >= 0:
                    gen_for_read(b)['ok'] = False
            for b in [op.out_buf] + ([op.term_buf[0]] if op.kind == OP_BEV_MAPS else []):     # BEV_MAPS has a second output
                if b >= 0:
                    gen_for_write(b)['ok'] = False
    for g in gens:
        if maxabs is not None:                               # range safety: measured max|x| of the tensor's producers
            ms = [maxabs[i] for i, r in g['uses'] if r == 'out' and maxabs[i] is not None]
            if not any(r == 'out' for _, r in g['uses']):    # initialised from outside the program: the host may state its range
                ms = [v for v in [getattr(P, 'buf_maxabs', {}).get(g['buf'])] if v is not None]
            m = max(ms) if ms else None
            if not h2_range_ok(m, getattr(P, 'calib_margin', CALIB_MARGIN)):
                g['ok'] = False
                for i, r in g['uses']:
                    if r == 'in' and P.ops[i].kind == OP_CONV and P.ops[i].weight_h2:
                        # an f16x2 kernel would split this float32 tensor in registers with the same 2^ACT_SHIFT: not this op
                        P.ops[i].weight_h2, P.ops[i].scale_h2 = 0, 0
                        P.range_fallback.append((i, P.names[i] if i < len(P.names) else 'op%d' % i, m))
                        o = P.ops[i]
                        if getattr(P, 'split_k_items', 0) and o.groups == 1 and o.Cin % 64 == 0 and o.Cout % 32 == 0 and o.ksize in (1, 3):
                            # split_k_groups left this layer whole for conv_h2k, which only runs on H2 tensors: on the f32 kernels it
                            # is a handful of work items with a serial loop over all input channels (slow at B = 1, still correct)
                            import warnings
                            warnings.warn('single-image plan: %s fell back to the float32 kernels (max|x| %.3g outside the fp16 pieces) and '
                                          'runs WITHOUT its input-channel split -- expect a slower frame; calibrate on representative '
                                          'frames or lower with ROMP_KSPLIT_WG=0' % (P.names[i] if i < len(P.names) else 'op%d' % i, m))
        fmt = FMT_H2 if (g['ok'] and any(r == 'out' for _, r in g['uses'])) else FMT_F32
        P.buf_fmt[g['buf']] = fmt
        for i, role in g['uses']:
            op = P.ops[i]
            op.act_shift = ACT_SHIFT
            if role == 'in':
                op.in_fmt = fmt
            elif role == 'res':
                op.res_fmt = fmt
            elif role == 'out':
                op.out_fmt = fmt
            else:
                op.term_fmt[role[1]] = fmt


def pack_h2_wave16(t):
    """An f16x2 weight pack ([tap T][Cin/16][piece 2][k-half 2][Cout][8] int16, pack_conv_weight_h2; Cin a multiple of 32, Cout of 16)
    re-ordered for the kernels whose waves own 16-channel groups (csrc/conv_h2c.h, conv_h2x.hip): per group, tap, 32-input-channel
    chunk kc and piece ONE 16-byte unit per lane -- the A operand of v_mfma_f32_16x16x32_f16: lane = 16 * kq + oc holds input
    channels 32 kc + 8 kq .. + 7 of output channel 16 g + oc.  -> [group Cout/16][tap T][kc Cin/32][piece 2][lane 64][8] (a pure
    permutation)."""
    T, c16, _, _, Cout, _ = t.shape
    assert tuple(t.shape) == (T, c16, 2, 2, Cout, 8) and c16 % 2 == 0 and Cout % 16 == 0, tuple(t.shape)
    v = t.reshape(T, c16 // 2, 2, 2, 2, Cout // 16, 16, 8)       # tap, kc, kq / 2, piece, kq % 2, group, oc, 8
    return v.permute(5, 0, 1, 3, 2, 4, 6, 7).contiguous().reshape(Cout // 16, T, c16 // 2, 2, 64, 8)


def fuse_bottleneck_seams(P):
    """Peephole like fuse_basic_blocks (csrc/conv_h2x.hip, ROMP_OP_SEAM1X1): the last conv of a layer1 Bottleneck (1x1 64 -> 256 +
    residual + ReLU, model.py:103-123) and the first conv of the next one (1x1 256 -> 64 + ReLU) as one launch that writes the
    256-channel tensor (the next block's residual) but does not read it back: 1.87 -> 1.34 GB per seam at B = 32, 2 709 -> 2 802
    images/s.  Batch plans (single-image plans: no measurable difference, 2.22-2.48 vs 2.33-2.35 ms per frame); env ROMP_FUSE_SEAMS=0
    switches it off, =all also fuses in single-image plans (A/B runs).  -> number of fused seams.

    Round 5 (OPF_SEAM_DS): behind Bottleneck 0 the residual is the output of the block's `downsample` conv (1x1 64 -> 256 + BN on the
    block input x0, model.py:289-301), the op right before the seam's first conv.  The seam kernel computes it as a second product
    from an x0 tile instead of loading it: that conv turns into a NOP as well (fields intact, two ops before the SEAM1X1), its
    256-channel output is never written or read (2 x 537 MB and one launch less at B = 32).  Only when nothing else reads that
    output and x0's buffer is still intact when the seam runs (build_hrnet32_backbone frees it one conv late for this);
    env ROMP_SEAM_DS=0 keeps the downsample a launch of its own (A/B runs)."""
    import os
    P.fused_seams = 0
    if not getattr(P, 'f16x2', False) or os.environ.get('ROMP_FUSE_SEAMS', '1') not in ('1', 'all') or (getattr(P, 'split_k_items', 0) and os.environ.get('ROMP_FUSE_SEAMS', '1') != 'all'):
        return 0
    by_ptr = {c.data_ptr(): c for c in P.consts if isinstance(c, torch.Tensor)}
    for i in range(len(P.ops) - 1):
        a, b = P.ops[i], P.ops[i + 1]
        if not (a.kind == OP_CONV and b.kind == OP_CONV):
            continue
        ok = (a.ksize == 1 and a.stride == 1 and a.groups == 1 and a.Cin == 64 and a.Cout == 256 and a.cin_pad == 64 and a.cout_pad == 256 and
              a.relu and a.res_buf >= 0 and a.weight_h2 and a.scale_h2 and
              b.ksize == 1 and b.stride == 1 and b.groups == 1 and b.Cin == 256 and b.Cout == 64 and b.cin_pad == 256 and b.cout_pad == 64 and
              b.relu and b.res_buf < 0 and b.weight_h2 and b.scale_h2 and
              b.in_buf == a.out_buf and (b.in_cstride, b.in_coff) == (a.out_cstride, a.out_coff) and a.stream == b.stream and
              (a.H, a.W) == (b.H, b.W) and (a.H * a.W) % 64 == 0 and
              all(o.out_rstride == 0 and o.out_bstride == 0 for o in (a, b)) and
              a.in_fmt == FMT_H2 and a.res_fmt == FMT_H2 and a.out_fmt == FMT_H2 and b.in_fmt == FMT_H2 and b.out_fmt == FMT_H2 and
              a.act_shift == b.act_shift)
        if not ok:
            continue
        for o, shape in ((a, (1, 4, 2, 2, 256, 8)), (b, (1, 16, 2, 2, 64, 8))):
            t = pack_h2_wave16(by_ptr[o.weight_h2].view(*shape))
            P.consts.append(t)
            o.weight_aux = t.data_ptr()
            o.flags |= OPF_WAVE16
        a.kind, b.kind = OP_NOP, OP_SEAM1X1
        P.flops[i + 1] += P.flops[i]
        P.flops[i] = 0.0
        P.bytes[i + 1] = 4.0 * a.H * a.W * (64 + 256 + 256 + 64)
        P.bytes[i] = 0.0
        P.fused_seams += 1
        d = P.ops[i - 1] if i >= 1 else None
        if d is None or os.environ.get('ROMP_SEAM_DS', '1') == '0':
            continue
        fold = (d.kind == OP_CONV and d.ksize == 1 and d.stride == 1 and d.groups == 1 and d.Cin == 64 and d.Cout == 256 and d.cin_pad == 64 and
                d.cout_pad == 256 and not d.relu and d.res_buf < 0 and d.weight_h2 and d.scale_h2 and d.out_buf == a.res_buf and
                (d.out_cstride, d.out_coff) == (a.res_cstride, a.res_coff) and d.out_rstride == 0 and d.out_bstride == 0 and
                d.stream == a.stream and (d.H, d.W) == (a.H, a.W) and d.in_fmt == FMT_H2 and d.out_fmt == FMT_H2 and
                d.act_shift == a.act_shift and d.in_cstride % 8 == 0 and d.in_coff % 8 == 0 and
                d.in_buf >= 0 and d.in_buf not in (a.out_buf, b.out_buf))           # x0 must still be there when the seam runs
        for j in range(i + 1, len(P.ops)):                       # nobody else reads the downsample's output (until the buffer's next writer)
            o = P.ops[j]
            if not fold or (o.out_buf == d.out_buf and o.kind not in (OP_NOP, OP_FORK, OP_JOIN, OP_RECORD, OP_WAIT)):
                break
            if d.out_buf in [o.in_buf, o.res_buf] + [o.term_buf[k] for k in range(o.n_terms)]:
                fold = False
        if not fold:
            continue
        t = pack_h2_wave16(by_ptr[d.weight_h2].view(1, 4, 2, 2, 256, 8))
        P.consts.append(t)
        d.weight_aux = t.data_ptr()
        d.flags |= OPF_WAVE16
        d.kind = OP_NOP
        b.flags |= OPF_SEAM_DS
        P.flops[i + 1] += P.flops[i - 1]
        P.flops[i - 1] = 0.0
        P.bytes[i + 1] = 4.0 * a.H * a.W * (64 + 64 + 256 + 64)
        P.bytes[i - 1] = 0.0
        P.folded_downsamples = getattr(P, 'folded_downsamples', 0) + 1
    return P.fused_seams


FUP_TILE = {32: (8, 64), 64: (8, 32), 128: (4, 32)}          # csrc/conv_fup.hip: output tile (rows, columns) by channel count
FUP_MAX_DIRECT = {32: 1, 64: 2, 128: 3}                      # FupCfg::MAXD: direct terms per output channel count
FUP_KERNELS = {(32, 1), (32, 2), (32, 3), (64, 1), (64, 2), (128, 1)}   # (Cout, up-terms) launch_fuseup instantiates


def fuse_up_sums(P):
    """Peephole (after assign_formats; csrc/conv_fup.hip, ROMP_OP_FUSEUP): a fuse-layer output whose up-terms are 1x1 convs of the
    lower-resolution branch outputs (model.py:186-196,233-244) computes them itself: the FUSESUM turns into FUSEUP with the convs'
    SOURCES as its up-terms and their weights (the slice of a merged conv that belongs to this output, repacked per 16-channel
    group) attached; a 1x1 conv all of whose consumers were fused turns into ROMP_OP_NOP.  H2 tensors throughout; HRNet's shapes
    (Cout 32 / 64 / 128, source k of Cout << k channels at 1 / 2^k resolution).  Batch plans only: a single image is 32 tiles of
    this kernel, on the launch-bound chain that costs more than the small convs did (frame 2.10 -> 2.54 ms in a same-box A/B).
    Measured at B = 32 (same box): serial kernel sum 12.02 -> 11.83 ms, images/s 2 996 -> 3 001: the up-convs used to hide on side
    streams.  Env ROMP_FUSEUP=0: off, =all: single-image plans too.  -> number of fused outputs."""
    import os
    P.fused_ups = 0
    mode = os.environ.get('ROMP_FUSEUP', '1')
    if not getattr(P, 'f16x2', False) or mode == '0' or (getattr(P, 'split_k_items', 0) and mode != 'all'):
        return 0
    by_ptr = {c.data_ptr(): c for c in P.consts if isinstance(c, torch.Tensor)}
    consumed = {}                                             # conv op index -> [fusesum op indices that took a slice of it]
    for fi, F in enumerate(P.ops):
        if F.kind != OP_FUSESUM or F.out_fmt != FMT_H2 or F.Cout not in FUP_TILE or F.n_terms < 2:
            continue
        th, tw = FUP_TILE[F.Cout]
        shifts = [F.term_shift[k] for k in range(F.n_terms)]
        n_dir = sum(1 for s in shifts if s == 0)
        ups = shifts[n_dir:]
        # exactly the kernels csrc/conv_fup.hip instantiates (launch_fuseup: (Cout, up-terms); FupCfg::MAXD direct terms) --
        # any other branch layout stays on FUSESUM + separate 1x1 convs
        if not (1 <= n_dir <= FUP_MAX_DIRECT[F.Cout] and (F.Cout, len(ups)) in FUP_KERNELS and shifts[:n_dir] == [0] * n_dir and
                ups == list(range(1, len(ups) + 1)) and F.H % th == 0 and F.W % tw == 0 and
                all(F.term_fmt[k] == FMT_H2 for k in range(F.n_terms)) and (th >> len(ups)) >= 1):
            continue
        prods = []
        for u, k in enumerate(range(n_dir, F.n_terms)):
            # the conv that wrote this term: the last writer of the buffer before the sum, covering the term's channel slice
            cand = [j for j in range(fi) if P.ops[j].kind == OP_CONV and P.ops[j].out_buf == F.term_buf[k]]
            U = P.ops[cand[-1]] if cand else None
            c0 = F.term_coff[k] - (U.out_coff if U is not None else 0)
            ok = (U is not None and U.ksize == 1 and U.stride == 1 and U.groups == 1 and not U.relu and U.res_buf < 0 and U.weight_h2 and U.scale_h2 and
                  U.in_fmt == FMT_H2 and U.out_fmt == FMT_H2 and U.Cin == (F.Cout << (u + 1)) and U.cin_pad == U.Cin and U.in_cstride == U.Cin and
                  U.in_coff == 0 and U.in_buf >= 0 and U.H == (F.H >> (u + 1)) and U.W == (F.W >> (u + 1)) and U.out_cstride == F.term_cstride[k] and
                  0 <= c0 and c0 + F.Cout <= U.Cout and c0 % 16 == 0 and U.act_shift == F.act_shift and
                  U.out_rstride == 0 and U.out_bstride == 0)
            if not ok:
                prods = None
                break
            prods.append((cand[-1], U, c0))
        if not prods:
            continue
        packs, scs, shs = [], [], []
        for j, U, c0 in prods:
            w = by_ptr[U.weight_h2].view(1, U.Cin // 16, 2, 2, U.cout_pad, 8)[:, :, :, :, c0:c0 + F.Cout, :]
            packs.append(pack_h2_wave16(w.contiguous()).reshape(-1))
            scs.append(by_ptr[U.scale_h2].reshape(-1)[c0:c0 + F.Cout])
            shs.append(by_ptr[U.shift].reshape(-1)[c0:c0 + F.Cout])
        pw, ps, pb = torch.cat(packs).contiguous(), torch.stack(scs).contiguous(), torch.stack(shs).contiguous()
        P.consts += [pw, ps, pb]
        F.kind = OP_FUSEUP
        F.weight_aux, F.scale_h2, F.shift = pw.data_ptr(), ps.data_ptr(), pb.data_ptr()
        F.flags |= OPF_WAVE16
        up_bytes = 0.0
        for (j, U, c0), k in zip(prods, range(n_dir, F.n_terms)):
            # the term is now the conv's SOURCE; account the conv's share of flops / bytes to this op
            P.bytes[fi] += 4.0 * U.H * U.W * (U.Cin - F.Cout)          # reads x_s instead of the small term
            P.flops[fi] += 2.0 * U.H * U.W * U.Cin * F.Cout
            F.term_buf[k], F.term_cstride[k], F.term_coff[k] = U.in_buf, U.in_cstride, 0
            consumed.setdefault(j, []).append(fi)
        P.fused_ups += 1
    # a conv whose every reader was fused no longer runs
    for j, fis in consumed.items():
        U = P.ops[j]
        readers = [i for i in range(j + 1, len(P.ops))
                   if any(P.ops[i].kind in (OP_FUSESUM,) and P.ops[i].term_buf[k] == U.out_buf for k in range(P.ops[i].n_terms))
                   or (P.ops[i].kind in (OP_CONV, OP_KSUM) and (P.ops[i].in_buf == U.out_buf or P.ops[i].res_buf == U.out_buf))]
        # (readers of a LATER live range of the same arena buffer come after its next writer: stop there)
        nxt = [i for i in range(j + 1, len(P.ops)) if P.ops[i].out_buf == U.out_buf and P.ops[i].kind not in (OP_NOP, OP_FORK, OP_JOIN, OP_RECORD, OP_WAIT)]
        if nxt:
            readers = [i for i in readers if i <= nxt[0]]
        if not readers:
            share = P.flops[j]
            U.kind = OP_NOP
            P.flops[j] = 0.0
            P.bytes[j] = 0.0
    return P.fused_ups


def stream_races(P):
    """Happens-before check of a lowered program: stream order + FORK / JOIN + RECORD / WAIT must order every pair of conflicting
    accesses to an arena buffer (write -> read, read -> write, write -> write; whole buffers, channel slices ignored: conservative
    for merged convs, whose slices share their writer anyway).  -> list of (kind, buffer, earlier op, later op).  Program.op_array()
    refuses a program with races; tests/test_cpu_host.py runs it on every plan kind."""
    NS = 4
    ops = P.ops
    vc = [[-1] * NS for _ in range(NS)]                      # vc[s][t]: the latest op of stream t that happens-before stream s's next op
    ev, at = {}, [None] * len(ops)
    in_par = False
    for i, op in enumerate(ops):
        if op.kind == OP_FORK:
            in_par = True
            for k in range(1, op.Cin + 1):
                vc[k] = [max(a, b) for a, b in zip(vc[k], vc[0])]
        elif op.kind == OP_JOIN:
            for k in range(1, op.Cin + 1):
                vc[0] = [max(a, b) for a, b in zip(vc[0], vc[k])]
            in_par = False
        elif op.kind == OP_RECORD:
            ev[op.Cin] = list(vc[op.stream])
        elif op.kind == OP_WAIT:
            vc[op.stream] = [max(a, b) for a, b in zip(vc[op.stream], ev[op.Cin])]
        elif op.kind != OP_NOP:
            s = op.stream if in_par else 0
            vc[s][s] = i
            at[i] = (s, list(vc[s]))

    def touched(i):
        op, r, w = ops[i], [], []
        srcs = [op] + ([ops[i - 1]] if op.kind in (OP_BBLOCK32, OP_BBLOCK64, OP_SEAM1X1) else [])    # a fused pair: the NOP before it holds the first conv
        seam_ds = op.kind == OP_SEAM1X1 and (op.flags & OPF_SEAM_DS)
        if seam_ds:                                              # the folded downsample: its input is read, its output never materialises
            r.append(ops[i - 2].in_buf)
        for o in srcs:
            if o.kind == OP_STEM2:                               # reads the caller's image only (its in_buf names the tensor that no longer exists)
                w += [o.out_buf] if o.out_buf >= 0 else []
                continue
            r += [b for b in ((o.in_buf,) if (seam_ds and o is not op) else (o.in_buf, o.res_buf)) if b >= 0]
            if o.kind in (OP_FUSESUM, OP_FUSEUP):
                r += [o.term_buf[k] for k in range(o.n_terms) if o.term_buf[k] >= 0]
            w += [b for b in [o.out_buf] + ([o.term_buf[0]] if o.kind == OP_BEV_MAPS else []) if b >= 0]
        return r, w

    ordered = lambda i, j: at[j][1][at[i][0]] >= i           # op i happens-before op j (i earlier in op order)

    def interleaved(i, j):
        """Two convs that write DIFFERENT parities of one tensor through the same sparse output strides (the four 2x2 parity convs
        of a transposed conv, resnet_plan.py: out_cstride = two pixels, out_rstride = two rows, out_coff = a rows + b pixels) touch
        disjoint elements of the buffer: not a write-write conflict."""
        a, b = ops[i], ops[j]
        if not (a.kind == OP_CONV and b.kind == OP_CONV and a.out_rstride > 0 and a.Cout * 2 <= a.out_cstride and
                (a.out_rstride, a.out_bstride, a.out_cstride, a.Cout) == (b.out_rstride, b.out_bstride, b.out_cstride, b.Cout)):
            return False
        par = lambda o: (o.out_coff // (o.out_rstride // 2), (o.out_coff % (o.out_rstride // 2)) // (o.out_cstride // 2))
        return par(a) != par(b)

    last_w, readers, races = {}, {}, []
    for j in range(len(ops)):
        if at[j] is None:
            continue
        r, w = touched(j)
        for b in r:
            i = last_w.get(b)
            if i is not None and not ordered(i, j):
                races.append(('RAW', b, P.names[i], P.names[j]))
            readers.setdefault(b, []).append(j)
        for b in w:
            i = last_w.get(b)
            if i is not None and i != j and not ordered(i, j) and not interleaved(i, j):
                races.append(('WAW', b, P.names[i], P.names[j]))
            races += [('WAR', b, P.names[i], P.names[j]) for i in readers.get(b, []) if i != j and not ordered(i, j)]
            last_w[b], readers[b] = j, []
    return races


def fuse_basic_blocks(P):
    """Peephole over the lowered program (after assign_formats): a 32-channel BasicBlock -- conv 3x3 s1 32->32 + BN + ReLU followed
    by conv 3x3 s1 32->32 + BN + (block input) + ReLU, model.py:54-83 -- whose tensors are all H2 becomes ONE launch
    (csrc/conv_h2b.hip: the intermediate tile stays in LDS): the first conv's op turns into ROMP_OP_NOP (fields intact: the
    kernel takes its weights from there), the second into ROMP_OP_BBLOCK32.  Op indices, names and the flop / byte lists keep
    their length; the pair's algorithmic bytes become x in + y out.  The same for 64-channel blocks (csrc/conv_h2c.h,
    ROMP_OP_BBLOCK64; their weights are repacked per wave into weight_aux).  Single-image plans fuse the 32-channel blocks only
    (64 tiles there, a quarter of the CUs, but 32 fewer launches on a launch-bound chain: network 2.15 -> 1.95 ms at B = 1; the
    64-channel kernel would run 32 tiles: 2.25 -> 2.45 ms per frame).  Env ROMP_FUSE_BLOCKS: 0 off, 32 / 64 one class, all
    both everywhere (A/B runs).  -> number of fused blocks."""
    import os
    P.fused_blocks = 0
    if not getattr(P, 'f16x2', False) or os.environ.get('ROMP_FUSE_BLOCKS', '1') == '0':
        return 0
    fuse_c = {'1': (32, 64), '32': (32,), '64': (64,), 'all': (32, 64)}.get(os.environ.get('ROMP_FUSE_BLOCKS', '1'), (32, 64))
    if getattr(P, 'split_k_items', 0) and os.environ.get('ROMP_FUSE_BLOCKS', '1') == '1':
        fuse_c = (32,)                                         # a single image is 32 tiles of the 64-channel kernel: slower than its two convs
    readers = {}
    for i, op in enumerate(P.ops):
        for b in [op.in_buf, op.res_buf] + [op.term_buf[k] for k in range(op.n_terms if op.kind == OP_FUSESUM else 0)]:
            if b >= 0:
                readers.setdefault(b, []).append(i)
    for i in range(len(P.ops) - 1):
        a, b = P.ops[i], P.ops[i + 1]
        if not (a.kind == OP_CONV and b.kind == OP_CONV):
            continue
        C_ = a.Cin
        plain = C_ in fuse_c and all(o.ksize == 3 and o.stride == 1 and o.groups == 1 and o.Cin == C_ and o.Cout == C_ and o.cin_pad == C_ and
                    o.cout_pad == C_ and o.relu and o.weight_h2 and o.scale_h2 and o.pad_h == -1 and o.pad_w == -1 and
                    o.out_rstride == 0 and o.out_bstride == 0 and o.H % (16 if C_ == 32 else 8) == 0 and o.W % 16 == 0 for o in (a, b))
        chained = (a.res_buf < 0 and b.in_buf == a.out_buf and a.out_buf >= 0 and b.res_buf == a.in_buf and a.in_buf >= 0 and b.out_buf >= 0 and
                   (b.res_cstride, b.res_coff) == (a.in_cstride, a.in_coff) and (b.in_cstride, b.in_coff) == (a.out_cstride, a.out_coff) and
                   a.stream == b.stream and (a.H, a.W) == (b.H, b.W))
        h2 = a.in_fmt == FMT_H2 and a.out_fmt == FMT_H2 and b.in_fmt == FMT_H2 and b.res_fmt == FMT_H2 and b.out_fmt == FMT_H2
        # the intermediate tensor must die in the second conv: nobody else reads that live range of its buffer
        later = [j for j in readers.get(a.out_buf, []) if j > i + 1]
        writers_between = [j for j in range(i + 2, later[0] + 1) if P.ops[j].out_buf == a.out_buf] if later else [0]
        private = not later or bool(writers_between)
        if plain and chained and h2 and private and a.act_shift == b.act_shift:
            # the row-pipelined kernels (conv_h2c.h) read their weights per wave (16 output channels each).  32 channels: batch plans only
            # (two workgroups per CU; a single image's tiles are better off on conv_h2b.hip's kernel: 2.10 vs 2.15 ms at B = 1)
            # (round 4 A/B runs, both neutral: conv_h2b's kernel in batch plans, the row-pipelined one at B = 1 -- 1.436 / 1.479 vs 1.458 / 1.461 ms)
            if C_ == 64 or not getattr(P, 'split_k_items', 0):
                by_ptr = {c.data_ptr(): c for c in P.consts if isinstance(c, torch.Tensor)}
                for o in (a, b):
                    t = pack_h2_wave16(by_ptr[o.weight_h2].view(9, C_ // 16, 2, 2, C_, 8))
                    P.consts.append(t)
                    o.weight_aux = t.data_ptr()          # (replaces the bf16x3 pack of conv_math='all': a fused op runs no bf16x3 kernel)
                    o.flags |= OPF_WAVE16                # the dispatch flag of launch_bblock32 / 64: never weight_aux != NULL alone
            a.kind, b.kind = OP_NOP, (OP_BBLOCK32 if C_ == 32 else OP_BBLOCK64)
            P.flops[i + 1] += P.flops[i]
            P.flops[i] = 0.0
            P.bytes[i + 1] = 4.0 * a.H * a.W * C_ * 2
            P.bytes[i] = 0.0
            P.fused_blocks += 1
    return P.fused_blocks


def _stem7p_pair(P, i):
    """ops[i - 1], ops[i] = ResNet-50's stem conv (MFMA-eligible) and the max-pool that is the only reader of its output?"""
    import os
    if not getattr(P, 'f16x2', False) or os.environ.get('ROMP_FUSE_STEM7P', '1') == '0' or i != 1:
        return False
    a, b = P.ops[0], P.ops[1]
    if not (a.kind == OP_STEM7 and b.kind == OP_MAXPOOL and not (a.flags & OPF_STEM_VALU) and a.out_buf >= 0 and b.in_buf == a.out_buf and
            b.in_cstride == a.out_cstride and a.out_coff == 0 and a.Cout == 64 and b.Cin == 64 and a.H % 32 == 0 and a.W % 32 == 0 and
            (b.H, b.W) == (a.H // 2, a.W // 2) and a.stream == b.stream and b.out_buf >= 0 and (b.out_cstride & 3) == 0):
        return False
    for j in range(2, len(P.ops)):                               # the conv's output must die in the pool
        o = P.ops[j]
        if o.kind in (OP_FORK, OP_JOIN, OP_RECORD, OP_WAIT):
            continue
        reads = [o.in_buf, o.res_buf] + [o.term_buf[k] for k in range(o.n_terms if o.kind in (OP_FUSESUM, OP_FUSEUP) else 0)]
        if a.out_buf in reads:
            return False
        if o.out_buf == a.out_buf and o.kind != OP_NOP:
            break
    return True


def fuse_stem7p(P):
    """Peephole (after assign_formats): ResNet-50's stem -- ROMP_OP_STEM7 (normalisation + conv7x7 s2 + BN + ReLU) followed by
    ROMP_OP_MAXPOOL (romp/lib/models/resnet_50.py:32-45,56) -- becomes ONE launch on the matrix cores (csrc/stem7p.hip): the conv's op
    turns into ROMP_OP_NOP, the pool's into ROMP_OP_STEM7P carrying the conv's fields and its own output (float32 or H2, as the format
    pass decided).  The 64-channel half-resolution tensor (16.8 MB per image) is never written or read.  f16x2 programs only (the
    float32 / calibration programs keep the exact VALU conv), weights within the fp16 pieces (ROMP_OPF_STEM_VALU otherwise, set by
    resnet_plan._stem7; env ROMP_STEM=valu forces it), env ROMP_FUSE_STEM7P=0: off (A/B runs).  -> 1 if fused."""
    P.fused_stem7p = 0
    if len(P.ops) < 2 or not _stem7p_pair(P, 1):
        return 0
    a, b = P.ops[0], P.ops[1]
    b.weight, b.scale, b.shift = a.weight, a.scale, a.shift
    b.H, b.W, b.Cin, b.Cout, b.ksize, b.stride, b.relu, b.groups = a.H, a.W, 3, 64, 7, 2, 1, 1
    b.in_buf, b.in_cstride, b.in_coff = BUF_IMAGE, 3, 0
    a.kind, b.kind = OP_NOP, OP_STEM7P
    P.flops[1] += P.flops[0]
    P.flops[0] = 0.0
    P.bytes[1] = 4.0 * (a.H * a.W * 3 + (a.H // 4) * (a.W // 4) * 64)        # the image in, the pooled tensor out
    P.bytes[0] = 0.0
    P.fused_stem7p = 1
    return 1


def fuse_stem2(P):
    """Peephole (after assign_formats): HRNet's stem -- ROMP_OP_STEM 3 -> 64 (MFMA form, H2 output) followed by the 3x3 stride-2
    64 -> 64 conv + BN + ReLU that is the only reader of its output (model.py:384-390) -- becomes ONE launch (csrc/stem2.hip): the
    stem's op turns into ROMP_OP_NOP (fields intact: the kernel takes weights / BN from there), the conv into ROMP_OP_STEM2 with
    its weights repacked per wave (pack_h2_wave16).  The 64-channel half-resolution tensor is never written or read: 2 x 16.8 MB
    per image off the serial head of the graph.  Env ROMP_FUSE_STEM2=0: off (A/B runs).  -> 1 if fused."""
    import os
    P.fused_stem2 = 0
    if not getattr(P, 'f16x2', False) or os.environ.get('ROMP_FUSE_STEM2', '1') == '0' or len(P.ops) < 2:
        return 0
    a, b = P.ops[0], P.ops[1]
    if not (a.kind == OP_STEM and b.kind == OP_CONV):
        return 0
    ok = (not (a.flags & OPF_STEM_VALU) and a.out_fmt == FMT_H2 and a.out_buf >= 0 and a.Cout == 64 and a.H % 64 == 0 and a.W % 64 == 0 and
          b.in_buf == a.out_buf and (b.in_cstride, b.in_coff) == (a.out_cstride, a.out_coff) and b.ksize == 3 and b.stride == 2 and
          b.groups == 1 and b.Cin == 64 and b.Cout == 64 and b.cin_pad == 64 and b.cout_pad == 64 and b.relu and b.relu_from == 0 and
          b.res_buf < 0 and b.weight_h2 and b.scale_h2 and b.in_fmt == FMT_H2 and b.out_fmt == FMT_H2 and b.pad_h == -1 and b.pad_w == -1 and
          b.out_rstride == 0 and b.out_bstride == 0 and b.out_buf >= 0 and (b.H, b.W) == (a.H // 2, a.W // 2) and a.stream == b.stream and
          a.act_shift == b.act_shift and ((b.out_cstride | b.out_coff) & 7) == 0)
    if not ok:
        return 0
    for j in range(2, len(P.ops)):                               # the stem's output must die in that conv: nobody else reads this live range
        o = P.ops[j]
        if o.kind in (OP_FORK, OP_JOIN, OP_RECORD, OP_WAIT):
            continue
        reads = [o.in_buf, o.res_buf] + [o.term_buf[k] for k in range(o.n_terms if o.kind in (OP_FUSESUM, OP_FUSEUP) else 0)]
        if a.out_buf in reads:
            return 0
        if o.out_buf == a.out_buf and o.kind != OP_NOP:
            break                                                  # (the buffer starts a new life)
    by_ptr = {c.data_ptr(): c for c in P.consts if isinstance(c, torch.Tensor)}
    t = pack_h2_wave16(by_ptr[b.weight_h2].view(9, 4, 2, 2, 64, 8))
    P.consts.append(t)
    b.weight_aux = t.data_ptr()
    b.flags |= OPF_WAVE16
    a.kind, b.kind = OP_NOP, OP_STEM2
    P.flops[1] += P.flops[0]
    P.flops[0] = 0.0
    P.bytes[1] = 4.0 * (a.H * a.W * 3 + (b.H // 2) * (b.W // 2) * 64)      # the image in, y out
    P.bytes[0] = 0.0
    P.fused_stem2 = 1
    return 1


class Program:
    """The lowered network: ops (ctypes), packed constants (kept alive here), buffer sizes."""

    def __init__(self, device):
        self.device = device
        self.ops: List[RompOp] = []
        self.names: List[str] = []
        self.flops: List[float] = []          # per image
        self.bytes: List[float] = []          # algorithmic HBM bytes per image (in + out + res)
        self.consts: List[torch.Tensor] = []
        self.buf_floats: List[int] = []
        self._free: Dict[int, List[int]] = {}
        self._pfree: Dict[int, Dict[int, List[int]]] = {}     # per-stream free lists inside a fork/join region
        self._later: List[List] = [[], []]                     # free_later(): [this epoch, the one before] of (buffer, stream)
        self.n_events = 0
        self._event_stream: Dict[int, int] = {}
        self.cur_stream = 0
        self.in_parallel = False
        self.parallel = True                                   # emit FORK/JOIN (False: one stream)
        self.bf16x3 = False                                    # also pack bf16x3-split weights (conv_bx3 / conv_bxd kernels)
        self.f16x2 = False                                     # also pack f16x2-split weights (conv_h2 / conv_h2d kernels)
        self.conv_math = 'f32'
        self.persistent = set()
        self.exported_bufs = set()                             # arena buffers the host reads through romp_net_buffer_ptr: float32
        self.buf_fmt: Dict[int, int] = {}                      # filled by assign_formats
        self.head_in_buf: Optional[int] = None
        self.head_in_ch, self.coord_off = HEAD_IN_CH, None     # coord_off: first of the two constant CoordConv channels
        self.split_k_items = 0                                 # > 0 (single-image plans): split a conv's input channels until it has this many work items

    # ---- buffers -------------------------------------------------------------------------
    def alloc(self, floats, persistent=False):
        if not persistent:
            if self.in_parallel:                                # buffers this stream released itself
                lst = self._pfree.get(self.cur_stream, {}).get(floats)
                if lst:
                    return lst.pop()
            lst = self._free.get(floats)                        # released before the fork: nobody uses them
            if lst:
                return lst.pop()
        self.buf_floats.append(int(floats))
        b = len(self.buf_floats) - 1
        if persistent:
            self.persistent.add(b)
        return b

    def free(self, act: Act):
        """Return a buffer for reuse.  Inside a fork/join region it only becomes visible to the
        stream that released it (other streams run concurrently); join() publishes it to all."""
        if act.buf >= 0 and act.buf not in self.persistent:
            pool = self._pfree.setdefault(self.cur_stream, {}) if self.in_parallel else self._free
            lst = pool.setdefault(self.buf_floats[act.buf], [])
            assert act.buf not in lst, 'double free of buffer %d' % act.buf
            lst.append(act.buf)

    def free_later(self, act: Act, stream=None):
        """Inside an OPEN region (hr_module's dataflow form): a buffer that OTHER streams read goes back to `stream`'s pool (default:
        the current one) only at the end of the NEXT epoch (epoch() = a module boundary).  By then every stream has passed a fuse
        output of the next module, which waited for every other stream's work of that module, which follows -- in stream order --
        that stream's reads of this module: the next writer is ordered after every reader without an edge of its own."""
        if act.buf >= 0 and act.buf not in self.persistent:
            if not self.in_parallel:
                return self.free(act)
            self._later[0].append((act.buf, self.cur_stream if stream is None else stream))

    def epoch(self):
        for buf, s in self._later[1]:
            lst = self._pfree.setdefault(s, {}).setdefault(self.buf_floats[buf], [])
            assert buf not in lst, 'double free of buffer %d' % buf
            lst.append(buf)
        self._later = [[], self._later[0]]

    def record(self):
        """ROMP_OP_RECORD on the current stream -> event number for wait() (None outside a region: one stream, op order rules)."""
        if not self.in_parallel:
            return None
        self._marker(OP_RECORD, self.n_events)
        self._event_stream[self.n_events] = self.cur_stream
        self.n_events += 1
        return self.n_events - 1

    def wait(self, event):
        if self.in_parallel and event is not None and self._event_stream[event] != self.cur_stream:
            self._marker(OP_WAIT, event)

    def _marker(self, kind, n):
        op = RompOp()
        op.kind, op.Cin = kind, n
        op.stream = self.cur_stream if kind in (OP_RECORD, OP_WAIT) else 0
        op.in_buf = op.out_buf = op.res_buf = BUF_NONE
        self.ops.append(op)
        self.names.append({OP_FORK: 'fork', OP_JOIN: 'join', OP_RECORD: 'record', OP_WAIT: 'wait'}[kind])
        self.flops.append(0.0)
        self.bytes.append(0.0)

    def fork(self, n_side):
        """Side streams 1..n_side may run concurrently with the main stream until join()."""
        assert not self.in_parallel
        if self.parallel and n_side > 0:
            self._marker(OP_FORK, n_side)
            self.in_parallel, self._n_side = True, n_side

    def on(self, stream):
        self.cur_stream = stream if self.in_parallel else 0

    def join(self):
        if self.in_parallel:
            self._marker(OP_JOIN, self._n_side)
            self.epoch()
            self.epoch()                                       # (a join is a full barrier: everything deferred is free now)
            for pool in self._pfree.values():
                for size, lst in pool.items():
                    self._free.setdefault(size, []).extend(lst)
            self._pfree = {}
            self.in_parallel, self.cur_stream = False, 0

    def new_act(self, C_, H, W):
        return Act(self.alloc(C_ * H * W), C_, H, W, C_)

    def _dev(self, t):
        t = t.to(self.device).contiguous()
        self.consts.append(t)
        return t

    # ---- ops -----------------------------------------------------------------------------
    def split_k_groups(self, cin, cout, ksize, stride, Ho, Wo):
        """Input-channel slices for a conv of a single-image plan: the smallest-tile kernels give ceil(Ho/8) * (Wo/16) *
        ceil(cout/32) work items with a serial loop over cin; slices of >= 32 channels multiply the items (csrc/stem_fuse.hip
        ksum_kernel adds the partial sums)."""
        if self.split_k_items <= 0 or ksize not in (1, 3) or Wo % 16 or cout % 8:
            return 1
        import os
        if self.f16x2 and (stride == 1 or ksize == 3) and cin % 64 == 0 and cout % 32 == 0 and os.environ.get('ROMP_KSPLIT_WG', '1') != '0':
            # round 4: csrc/conv_h2k.hip splits the input channels across the WAVES of a workgroup and reduces in LDS -- the layer
            # stays one conv op (no float32 partial tensors, no ksum launch); the autotuner picks it wherever the tensors are H2
            return 1
        items = -(-Ho // 8) * (Wo // 16) * -(-cout // 32)
        g = 1
        while items * g < self.split_k_items and cin % (2 * g) == 0 and cin // (2 * g) >= 32 and (cin // (2 * g)) % 16 == 0:
            g *= 2
        return g

    def conv(self, name, x: Act, w, scale, shift, ksize, stride, relu, res: Optional[Act] = None,
             out: Optional[Act] = None, groups=1, out_buf_special=None, out_cstride=None, out_coff=0,
             pad=(-1, -1), out_rstride=0, out_bstride=0, relu_from=0):
        """One conv layer; in a single-image plan (split_k_items) a layer with few pixels and many input channels becomes a
        grouped conv over input-channel slices writing float32 partial sums + a ksum op with the layer's epilogue.
        `relu_from` (with relu): ReLU on output channels >= relu_from only (merged sibling convs, a multiple of 32)."""
        assert relu_from == 0 or (relu and relu_from % 32 == 0 and groups == 1), relu_from
        G = 1
        if groups == 1 and out_buf_special is None and not out_rstride and not out_bstride and tuple(pad) == (-1, -1) and x.coff % 8 == 0:
            cout, cin = w[0].shape[0], w[0].shape[1]
            if cin == x.C and x.cstride == x.C:
                Ho = (x.H + 2 * (ksize // 2) - ksize) // stride + 1
                G = self.split_k_groups(cin, cout, ksize, stride, Ho, (x.W + 2 * (ksize // 2) - ksize) // stride + 1)
        if G == 1:
            o = self._conv_op(name, x, w, scale, shift, ksize, stride, relu, res, out, groups, out_buf_special, out_cstride,
                              out_coff, pad, out_rstride, out_bstride)
            self.ops[-1].relu_from = relu_from
            return o
        cg = cin // G
        part = self._conv_op(name + '.splitk', x, [w[0][:, g * cg:(g + 1) * cg].contiguous() for g in range(G)],
                             [torch.ones(cout)] * G, [torch.zeros(cout)] * G, ksize, stride, False, groups=G)
        if out is None:
            out = self.new_act(cout, part.H, part.W)
        op = RompOp()
        op.kind, op.in_buf, op.out_buf, op.res_buf = OP_KSUM, part.buf, out.buf, (res.buf if res is not None else BUF_NONE)
        op.H, op.W, op.Cin, op.Cout, op.groups, op.relu = part.H, part.W, G * cout, cout, G, int(relu)
        op.in_cstride, op.out_cstride, op.out_coff = part.cstride, out.cstride, out.coff
        if res is not None:
            op.res_cstride, op.res_coff = res.cstride, res.coff
        ps, pb = self._dev(scale[0].float()), self._dev(shift[0].float())
        op.scale, op.shift = ps.data_ptr(), pb.data_ptr()
        op.relu_from = relu_from
        op.stream = self.cur_stream
        self.ops.append(op)
        self.names.append(name + '.ksum')
        self.flops.append(float(part.H * part.W * cout * (G + 1)))
        self.bytes.append(4.0 * part.H * part.W * cout * (G + 1 + (1 if res is not None else 0)))
        self.free(part)
        return out

    def _conv_op(self, name, x: Act, w, scale, shift, ksize, stride, relu, res: Optional[Act] = None,
                 out: Optional[Act] = None, groups=1, out_buf_special=None, out_cstride=None, out_coff=0,
                 pad=(-1, -1), out_rstride=0, out_bstride=0):
        """w: list (per group) of OIHW tensors; scale/shift: list of per-group vectors.  pad: zero rows / columns
        before the first tap (-1: ksize//2); out_rstride / out_bstride: sparse output row / image strides (floats) for
        the interleaved parity outputs of a transposed conv (`out` then is the full-resolution tensor)."""
        cout, cin = w[0].shape[0], w[0].shape[1]
        cin_phys = x.C // groups if groups > 1 else x.C      # channels the loader may touch
        cin_pad, cout_pad = conv_pads(cin_phys, cout, ksize)
        assert cin <= cin_phys
        pw = torch.stack([pack_conv_weight(wi, cin_pad, cout_pad) for wi in w])
        ps = torch.zeros(groups, cout_pad)
        pb = torch.zeros(groups, cout_pad)
        for g in range(groups):
            ps[g, :cout], pb[g, :cout] = scale[g], shift[g]
        pw, ps, pb = self._dev(pw), self._dev(ps), self._dev(pb)
        paux = None
        if self.bf16x3 and cin_pad % 16 == 0:
            paux = self._dev(torch.stack([pack_conv_weight_bx3(wi, cin_pad, cout_pad) for wi in w]))
        kh, kw = (1, 3) if ksize == 13 else (ksize, ksize)
        if ksize == 2:
            Ho, Wo = x.H // stride, x.W // stride
        else:
            Ho = (x.H + 2 * (kh // 2) - kh) // stride + 1
            Wo = (x.W + 2 * (kw // 2) - kw) // stride + 1
        if out is None and out_buf_special is None:
            out = self.new_act(cout * groups, Ho, Wo)
        op = RompOp()
        op.kind, op.in_buf, op.res_buf = OP_CONV, x.buf, (res.buf if res is not None else BUF_NONE)
        op.out_buf = out_buf_special if out_buf_special is not None else out.buf
        op.H, op.W, op.Cin, op.Cout = x.H, x.W, cin_phys, cout
        op.ksize, op.stride, op.relu, op.groups = ksize, stride, int(relu), groups
        op.in_cstride, op.in_coff, op.in_gstride = x.cstride, x.coff, (cin_phys if groups > 1 else 0)
        if out_buf_special is not None:
            op.out_cstride, op.out_coff, op.out_gstride = out_cstride, out_coff, 0
        else:
            op.out_cstride, op.out_coff, op.out_gstride = out.cstride, out.coff, (cout if groups > 1 else 0)
        if res is not None:
            op.res_cstride, op.res_coff, op.res_gstride = res.cstride, res.coff, (cout if groups > 1 else 0)
        op.cin_pad, op.cout_pad = cin_pad, cout_pad
        op.pad_h, op.pad_w, op.out_rstride, op.out_bstride = int(pad[0]), int(pad[1]), int(out_rstride), int(out_bstride)
        op.weight, op.scale, op.shift = pw.data_ptr(), ps.data_ptr(), pb.data_ptr()
        if paux is not None:
            op.weight_aux = paux.data_ptr()
        if self.f16x2 and cin_pad % 16 == 0:
            packs = [pack_conv_weight_h2(wi, cin_pad, cout_pad) for wi in w]
            ph = self._dev(torch.stack([pk for pk, _ in packs]))
            psh = torch.zeros(groups, cout_pad, dtype=torch.float64)
            for g in range(groups):
                psh[g, :cout] = scale[g].double() * (2.0 ** -(packs[g][1] + ACT_SHIFT))      # exact
            psh = self._dev(psh.float())
            op.weight_h2, op.scale_h2, op.act_shift = ph.data_ptr(), psh.data_ptr(), ACT_SHIFT
        op.stream = self.cur_stream
        self.ops.append(op)
        self.names.append(name)
        self.flops.append(2.0 * Ho * Wo * cout * cin * kh * kw * groups)
        nbytes = 4.0 * (x.H * x.W * cin_phys * groups + Ho * Wo * cout * groups * (2 if res is not None else 1))
        self.bytes.append(nbytes)
        return out

    def stem(self, name, w, scale, shift, H, W):
        out = self.new_act(64, H // 2, W // 2)
        pw = self._dev(w.permute(2, 3, 1, 0).reshape(27, 64).contiguous())
        ps, pb = self._dev(scale), self._dev(shift)
        op = RompOp()
        op.kind, op.in_buf, op.out_buf, op.res_buf = OP_STEM, BUF_IMAGE, out.buf, BUF_NONE
        op.H, op.W, op.Cin, op.Cout, op.ksize, op.stride, op.relu, op.groups = H, W, 3, 64, 3, 2, 1, 1
        op.in_cstride, op.out_cstride = 3, 64
        op.weight, op.scale, op.shift = pw.data_ptr(), ps.data_ptr(), pb.data_ptr()
        # stem_mfma_kernel splits 256 w into fp16 pieces: a weight beyond +-255.9 would be clamped -> the exact float32
        # VALU kernel for such a checkpoint; env ROMP_STEM=valu forces it (A/B runs)
        import os
        wmax = float(w.abs().max()) if w.numel() else 0.0
        if os.environ.get('ROMP_STEM', '') == 'valu' or not (wmax * 256.0 < 65504.0):
            op.flags |= OPF_STEM_VALU
        self.ops.append(op)
        self.names.append(name)
        self.flops.append(2.0 * (H // 2) * (W // 2) * 64 * 27)
        self.bytes.append(4.0 * (H * W * 3 + (H // 2) * (W // 2) * 64))
        return out

    def fusesum(self, name, terms: List[Act], shifts: List[int], relu=True, out: Optional[Act] = None):
        t0 = terms[0]
        H, W = t0.H << shifts[0], t0.W << shifts[0]
        if out is None:
            out = self.new_act(t0.C, H, W)
        op = RompOp()
        op.kind, op.in_buf, op.out_buf, op.res_buf = OP_FUSESUM, BUF_NONE, out.buf, BUF_NONE
        op.H, op.W, op.Cin, op.Cout, op.relu = H, W, t0.C, t0.C, int(relu)
        op.out_cstride, op.out_coff = out.cstride, out.coff
        op.n_terms = len(terms)
        nbytes = 4.0 * H * W * t0.C
        for k, (t, s) in enumerate(zip(terms, shifts)):
            assert t.C == t0.C and t.coff % 8 == 0 and (t.H << s) == H
            op.term_buf[k], op.term_shift[k], op.term_cstride[k], op.term_coff[k] = t.buf, s, t.cstride, t.coff
            nbytes += 4.0 * t.H * t.W * t.C
        op.stream = self.cur_stream
        self.ops.append(op)
        self.names.append(name)
        self.flops.append(float(H * W * t0.C * (len(terms) - 1)))
        self.bytes.append(nbytes)
        return out

    def op_array(self):
        """The lowered op list as a ctypes array.  Lowering (tensor formats, block fusion) happens on the first call; later calls
        (export.save_plan on a program its net already created) return the same ops -- the format pass reads conv kinds and
        must not see the fused ones."""
        if not getattr(self, '_lowered', False):
            assign_formats(self)
            fuse_stem2(self)                 # (first: its reader analysis wants every op still a plain conv)
            fuse_stem7p(self)
            fuse_basic_blocks(self)
            fuse_bottleneck_seams(self)
            fuse_up_sums(self)
            races = stream_races(self)
            assert not races, 'the program races across its streams: %s' % (races[:4],)
            self._lowered = True
        arr = (RompOp * len(self.ops))()
        for i, o in enumerate(self.ops):
            arr[i] = o
        return arr


def _clean(sd):
    return {k: v.detach().float().cpu() for k, v in sd.items() if not k.endswith('num_batches_tracked')}


def build_hrnet32_backbone(P: Program, sd, input_size=512, out_cstride=32) -> Act:
    """HigherResolutionNet (model.py:246-417) -> ops in P; returns the 32-channel 1/4-resolution output
    (written with channel stride `out_cstride` into a persistent buffer)."""

    def cbr(name, x, conv, bn, k, stride, relu, res=None, out=None):
        w = sd[conv + '.weight']
        s, b = fold_bn(sd, bn, w.shape[0], sd.get(conv + '.bias'))
        return P.conv(name, x, [w], [s], [b], k, stride, relu, res=res, out=out)

    bb = 'backbone.'
    # ---- stem (model.py:384-390)
    s, b = fold_bn(sd, bb + 'bn1', 64)
    x = P.stem('stem.conv1', sd[bb + 'conv1.weight'], s, b, input_size, input_size)
    y = cbr('stem.conv2', x, bb + 'conv2', bb + 'bn2', 3, 2, True)
    P.free(x)
    x = y
    # ---- layer1: 4 Bottlenecks (model.py:103-123, :345)
    late_free = None
    for i in range(4):
        p = f'{bb}layer1.{i}.'
        t1 = cbr(p + 'conv1', x, p + 'conv1', p + 'bn1', 1, 1, True)
        if late_free is not None:                                # the stem output: when the downsample conv is folded into the seam kernel
            P.free(late_free)                                    # (fuse_bottleneck_seams, OPF_SEAM_DS) that launch -- THIS conv's -- still reads it
            late_free = None
        t2 = cbr(p + 'conv2', t1, p + 'conv2', p + 'bn2', 3, 1, True)
        P.free(t1)
        if (p + 'downsample.0.weight') in sd:
            r = cbr(p + 'downsample', x, p + 'downsample.0', p + 'downsample.1', 1, 1, False)
            late_free = x
        else:
            r = x
        y = cbr(p + 'conv3', t2, p + 'conv3', p + 'bn3', 1, 1, True, res=r)
        P.free(t2)
        P.free(r)
        x = y
    # ---- transition1 (model.py:393-398)
    # (the two transition convs read the same tensor and are independent: two streams -- scripts/timeline.py counts 3.9 ms of a
    # forward with a single kernel in flight; every pair that can co-run takes some of it back)
    import os
    merge_s2 = os.environ.get('ROMP_MERGE_S2', '1') != '0'       # A/B switch: 0 = one launch per stride-2 conv as in rounds 1-3
    # Round 4: the stages form ONE open fork .. join region with point-to-point edges (ROMP_OP_RECORD / ROMP_OP_WAIT) instead of two
    # full fork/join barriers per module: branch j lives on stream j from transition1 to the last module; a fuse output waits for
    # exactly the tensors it sums; the next module's branch follows its own fuse output in stream order.  A cross-stream hand-over
    # costs 5-10 us on this runtime (profiles/r04_notes.md, section 8) and a barrier idles every stream until the slowest branch is done.
    # ROMP_DATAFLOW=0: the barrier form of rounds 1-3 (A/B runs).
    dataflow = P.parallel and os.environ.get('ROMP_DATAFLOW', '1') != '0'
    P.fork(3 if dataflow else 1)
    P.on(0)
    t10 = cbr('transition1.0', x, bb + 'transition1.0.0', bb + 'transition1.0.1', 3, 1, True)
    P.on(1)
    t11 = cbr('transition1.1', x, bb + 'transition1.1.0.0', bb + 'transition1.1.0.1', 3, 2, True)
    xs = [t10, t11]
    if dataflow:
        P.free_later(x, 0)                                       # (stream 1 reads it too)
    else:
        P.join()
        P.free(x)

    def hr_module(prefix, xs, n_out, final_out: Optional[Act] = None):
        """HighResolutionModule.forward (model.py:226-244).  The branches are independent until the
        fuse (model.py:230-231) and so are the fuse outputs: each runs on its own HIP stream, so the
        persistent conv kernels of different branches share the CUs and fill each other's barrier /
        load gaps and launch tails.

        Region 1 (one stream per branch j): the 4 BasicBlocks, then everything of the fuse layers that reads ONLY xs[j]: the
        FIRST stride-2 conv of every chain that starts at branch j (model.py:198-221) -- as ONE conv with concatenated output
        channels when there are several (round 4: towards outputs j+1, j+2, j+3 they all read the same tensor; it is now read
        once: [no-ReLU channels of the chain that ends here | ReLU channels of the longer chains], romp_op.relu_from) -- and the
        1x1 up-convs towards the outputs above (model.py:186-196).  Region 2 (one stream per output i): the rest of the chains
        and the sum.

        `dataflow` (round 4, the default): no fork / join here -- the caller holds ONE region open over all stages.  Stream j
        records an event behind its stride-2 convs and one behind its up-convs; output i waits for the first kind from the branches
        below it and the second kind from those above it; what another stream reads is released an epoch late (Program.free_later),
        what only its own stream touches at once."""
        nb = len(xs)
        xs = list(xs)
        ch = [x.C for x in xs]
        first, ups, temps = {}, {}, []                            # (i, j) -> Act
        ev_s2, ev_up = {}, {}                                     # dataflow: the events after branch j's stride-2 convs / up-convs
        if not dataflow:
            P.fork(nb - 1)
        for br in range(nb):
            P.on(br)
            for k in range(4):                                   # branch: 4 BasicBlocks
                q = f'{prefix}branches.{br}.{k}.'
                t = cbr(q + 'conv1', xs[br], q + 'conv1', q + 'bn1', 3, 1, True)
                y = cbr(q + 'conv2', t, q + 'conv2', q + 'bn2', 3, 1, True, res=xs[br])
                P.free(t)
                P.free(xs[br])
                xs[br] = y
            targets = list(range(br + 1, n_out))                 # chains br -> i start with a conv on xs[br]
            if len(targets) > 1 and merge_s2:
                ws, ss, bs = [], [], []
                for i in targets:                                # i = br + 1 first: its single conv is the chain's LAST (no ReLU)
                    q = f'{prefix}fuse_layers.{i}.{br}.0.'
                    w = sd[q + '0.weight']
                    sc, sh = fold_bn(sd, q + '1', w.shape[0], sd.get(q + '0.bias'))
                    ws.append(w); ss.append(sc); bs.append(sh)
                name = f'{prefix}fuse_layers.{targets[0]}-{targets[-1]}.{br}.0'
                m = P.conv(name, xs[br], [torch.cat(ws, 0)], [torch.cat(ss)], [torch.cat(bs)], 3, 2, True, relu_from=ws[0].shape[0])
                temps.append(m)
                off = 0
                for i, w in zip(targets, ws):
                    first[(i, br)] = Act(m.buf, w.shape[0], m.H, m.W, m.cstride, m.coff + off)
                    off += w.shape[0]
            else:
                for i in targets:
                    q = f'{prefix}fuse_layers.{i}.{br}.'
                    first[(i, br)] = cbr(f'{q}0', xs[br], f'{q}0.0', f'{q}0.1', 3, 2, i - br != 1)
                    temps.append(first[(i, br)])
            if dataflow and targets:
                ev_s2[br] = P.record()
            ups_to = list(range(min(br, n_out)))                 # 1x1 conv + BN towards every output i < br; the upsample is folded into fusesum
            if len(ups_to) > 1 and merge_s2:                     # the same merge for the up-convs: one launch, xs[br] read once
                ws, ss, bs = [], [], []
                for i in ups_to:
                    q = f'{prefix}fuse_layers.{i}.{br}.'
                    w = sd[q + '0.weight']
                    sc, sh = fold_bn(sd, q + '1', w.shape[0], sd.get(q + '0.bias'))
                    ws.append(w); ss.append(sc); bs.append(sh)
                m = P.conv(f'{prefix}fuse_layers.{ups_to[0]}-{ups_to[-1]}.{br}.up', xs[br], [torch.cat(ws, 0)], [torch.cat(ss)], [torch.cat(bs)], 1, 1, False)
                temps.append(m)
                off = 0
                for i, w in zip(ups_to, ws):
                    ups[(i, br)] = Act(m.buf, w.shape[0], m.H, m.W, m.cstride, m.coff + off)
                    off += w.shape[0]
            else:
                for i in ups_to:
                    q = f'{prefix}fuse_layers.{i}.{br}.'
                    ups[(i, br)] = cbr(q + 'up', xs[br], q + '0', q + '1', 1, 1, False)
                    temps.append(ups[(i, br)])
            if dataflow and ups_to:
                ev_up[br] = P.record()
            if dataflow:
                for t in temps:                                  # written here, read on other streams: back to THIS stream's pool an epoch late
                    P.free_later(t, br)
                temps = []
        outs = []
        if not dataflow:
            P.join()
            P.fork(n_out - 1)
        for i in range(n_out):
            P.on(i)
            if dataflow:
                for j in range(nb):
                    if j != i:
                        P.wait(ev_s2[j] if j < i else ev_up[j])
            terms, shifts = [], []
            for j in range(nb):
                q = f'{prefix}fuse_layers.{i}.{j}.'
                if j == i:
                    terms.append(xs[j]); shifts.append(0)
                elif j > i:
                    terms.append(ups[(i, j)]); shifts.append(j - i)
                else:                                            # the rest of the chain of stride-2 3x3 convs
                    t = first[(i, j)]
                    for k in range(1, i - j):
                        t2 = cbr(f'{q}{k}', t, f'{q}{k}.0', f'{q}{k}.1', 3, 2, k != i - j - 1)
                        if k > 1:
                            P.free(t)
                        t = t2
                    if i - j > 1:
                        temps.append(t)
                    terms.append(t); shifts.append(0)
            outs.append(P.fusesum(f'{prefix}fuse.{i}', terms, shifts, True, out=final_out if n_out == 1 else None))
            if dataflow:                                         # this stream wrote and read them: its own pool, at once
                for t in temps:
                    P.free(t)
                temps = []
        if dataflow:
            for j in range(nb):                                  # branch 0's output is read on its own stream only (convs above, fuse.0);
                P.on(j)                                          # the others also by the FUSEUP kernels of the outputs above them
                (P.free if j == 0 else P.free_later)(xs[j])      # (plan.fuse_up_sums moves their 1x1 up-convs into the sums)
            P.epoch()
            return outs
        P.join()
        for t in temps:
            P.free(t)
        for xj in xs:
            P.free(xj)
        return outs

    def transition(name, src, conv, bn, new_stream):
        """A new branch from the lowest-resolution one: the conv runs on its SOURCE's stream (the source tensor is that stream's to
        release), the new stream picks the result up through an edge."""
        if not dataflow:
            return cbr(name, src, conv, bn, 3, 2, True)
        P.on(new_stream - 1)
        t = cbr(name, src, conv, bn, 3, 2, True)
        e = P.record()
        P.on(new_stream)
        P.wait(e)
        return t

    ys = hr_module(bb + 'stage2.0.', xs, 2)
    # ---- transition2 / stage3 (model.py:401-407)
    xs = [ys[0], ys[1], transition('transition2.2', ys[-1], bb + 'transition2.2.0.0', bb + 'transition2.2.0.1', 2)]
    for m in range(4):
        xs = hr_module(f'{bb}stage3.{m}.', xs, 3)
    # ---- transition3 / stage4 (model.py:409-416)
    xs = [xs[0], xs[1], xs[2], transition('transition3.3', xs[-1], bb + 'transition3.3.0.0', bb + 'transition3.3.0.1', 3)]
    for m in range(2):
        xs = hr_module(f'{bb}stage4.{m}.', xs, 4)
    # last module emits branch 0 only -> straight into the (persistent) head input buffer
    fs = input_size // 4
    P.head_in_buf = P.alloc(out_cstride * fs * fs, persistent=True)
    head_in = Act(P.head_in_buf, 32, fs, fs, out_cstride)
    hr_module(f'{bb}stage4.2.', xs, 1, final_out=head_in)
    if dataflow:
        P.join()
    return head_in


def build_romp_hrnet32(sd: Dict[str, torch.Tensor], device, input_size=512, bf16x3=False, split_k_items=0) -> Program:
    """state_dict of ROMPv1 (model.py:420-481) -> Program.  `bf16x3`: the conv_math setting (see set_conv_math);
    `split_k_items` > 0: single-image plan (Program.conv)."""
    sd = _clean(sd)
    P = Program(device)
    P.split_k_items = split_k_items
    set_conv_math(P, bf16x3)
    # backbone output lands in 32 of the 40 channels of the head input buffer; channels 32,33 hold the
    # constant CoordConv maps (model.py:473), 34..39 are zero padding.
    build_hrnet32_backbone(P, sd, input_size, out_cstride=HEAD_IN_CH)
    fs = input_size // 4
    head_x = Act(P.head_in_buf, HEAD_IN_CH, fs, fs, HEAD_IN_CH)
    P.head_in_ch, P.coord_off = HEAD_IN_CH, 32
    build_romp_head(P, sd, head_x, 34)
    return P


def build_romp_head(P: Program, sd, head_x: Act, cin: int):
    """The three head towers (model.py:445-481; training tree romp_model.py:78-103) on `head_x`, whose first `cin`
    channels are [backbone features | 2 CoordConv maps]: the towers share their input, so the first conv runs as one
    cin->192 conv, the BasicBlocks as 3-group convs, then three 1x1 output convs into the caller's tensors."""
    heads = (1, 2, 3)                                            # params(142), center(1), cam(3)
    w0, s0, b0 = [], [], []
    for h in heads:
        p = f'final_layers.{h}.0.'
        w = sd[p + '0.weight']
        wp = torch.zeros(64, head_x.cstride, 3, 3)
        wp[:, :cin] = w
        s, b = fold_bn(sd, p + '1', 64, sd[p + '0.bias'])
        w0.append(wp); s0.append(s); b0.append(b)
    t = P.conv('head.conv0', head_x, [torch.cat(w0, 0)], [torch.cat(s0)], [torch.cat(b0)], 3, 2, True)
    P.flops[-1] *= cin / float(head_x.C)                         # algorithmic work: the reference's `cin` channels, not the zero padding
    for blk in range(2):
        ws1, ss1, bs1, ws2, ss2, bs2 = [], [], [], [], [], []
        for h in heads:
            q = f'final_layers.{h}.1.{blk}.0.'
            ws1.append(sd[q + 'conv1.weight']); s, b = fold_bn(sd, q + 'bn1', 64); ss1.append(s); bs1.append(b)
            ws2.append(sd[q + 'conv2.weight']); s, b = fold_bn(sd, q + 'bn2', 64); ss2.append(s); bs2.append(b)
        u = P.conv(f'head.block{blk}.conv1', t, ws1, ss1, bs1, 3, 1, True, groups=3)
        v = P.conv(f'head.block{blk}.conv2', u, ws2, ss2, bs2, 3, 1, True, res=t, groups=3)
        P.free(u)
        P.free(t)
        t = v
    P.fork(len(heads) - 1)                                       # the three output convs are independent: one stream each
    for gi, h in enumerate(heads):
        P.on(gi)
        p = f'final_layers.{h}.2'
        w = sd[p + '.weight']
        s, b = fold_bn(sd, None, w.shape[0], sd[p + '.bias'])
        xin = Act(t.buf, 64, t.H, t.W, t.cstride, 64 * gi)
        if h == 2:
            P.conv('head.center', xin, [w], [s], [b], 1, 1, False, out_buf_special=BUF_CENTER, out_cstride=1, out_coff=0)
        else:   # params_maps = cat([cam_maps, params_maps], 1)  (model.py:480)
            P.conv('head.params' if h == 1 else 'head.cam', xin, [w], [s], [b], 1, 1, False,
                   out_buf_special=BUF_PARAMS, out_cstride=145, out_coff=3 if h == 1 else 0)
    P.join()
    P.free(t)


def coord_channels(max_batch, size, device, n_ch=HEAD_IN_CH, off=32):
    """Initial content of the head input buffer: CoordConv maps in channels off, off+1
    (get_coord_maps model.py:8-37: ch0 varies along W, ch1 along H), zeros elsewhere."""
    r = torch.arange(size, dtype=torch.float32) / (size - 1) * 2 - 1
    buf = torch.zeros(max_batch, size, size, n_ch)
    buf[..., off] = r.view(1, 1, size)
    buf[..., off + 1] = r.view(1, size, 1)
    return buf.to(device)
