"""ROMP with the ResNet-50 backbone lowered to the op program (BASELINE configs[0]).

Reference (training tree only; simple_romp ships HRNet-32 alone): romp/lib/models/resnet_50.py ResNet_50 :19-120,
romp/lib/models/basic_modules.py Bottleneck :90-128, romp/lib/models/romp_model.py head :35-103.

  * stem     (x/255 - mean)/std + conv7x7 s2 p3 + BN + ReLU -> ROMP_OP_STEM7; MaxPool2d(3,2,1) -> ROMP_OP_MAXPOOL
  * layer1-4 Bottlenecks [3,4,6,3]: 1x1, 3x3 (stride on the 3x3), 1x1 (+1x1 strided downsample) -> conv ops
  * deconv   ConvTranspose2d(k4, s2, p1) + BN + ReLU x3: each is FOUR 2x2 convolutions, one per output parity
             (out[2y+a, 2x+b] only sees kernel taps ky in {3,1} (a=0) or {2,0} (a=1), likewise kx), written
             interleaved into the full-resolution tensor through the conv op's sparse output strides -- exactly the
             transposed conv's FLOPs, no zero-stuffing, no col2im pass
  * head     the three towers of plan.build_romp_head on [64 features | 2 CoordConv | 6 pad] channels
"""
import torch

from .lib import RompOp, OPF_STEM_VALU
from .plan import Act, Program, build_romp_head, fold_bn, _clean, set_conv_math, BUF_IMAGE, BUF_NONE

OP_STEM7, OP_MAXPOOL = 9, 10
LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))
RESNET_HEAD_CH = 80       # 64 backbone + 2 CoordConv channels, zero-padded to a multiple of 16 (f16x2 channel chunk)


def _stem7(P: Program, name, w, scale, shift, H, W):
    out = P.new_act(64, H // 2, W // 2)
    pw = P._dev(w.permute(2, 3, 1, 0).reshape(147, 64).contiguous())          # [ky][kx][ci][co]
    ps, pb = P._dev(scale), P._dev(shift)
    op = RompOp()
    op.kind, op.in_buf, op.out_buf, op.res_buf = OP_STEM7, BUF_IMAGE, out.buf, BUF_NONE
    op.H, op.W, op.Cin, op.Cout, op.ksize, op.stride, op.relu, op.groups = H, W, 3, 64, 7, 2, 1, 1
    op.in_cstride, op.out_cstride = 3, 64
    op.weight, op.scale, op.shift = pw.data_ptr(), ps.data_ptr(), pb.data_ptr()
    # the MFMA form (plan.fuse_stem7p: conv + pool as one kernel) splits 256 w into fp16 pieces: a weight beyond +-255.9 would be clamped
    # -> the exact float32 VALU kernel + the pool for such a checkpoint; env ROMP_STEM=valu forces it (A/B runs, tests)
    import os
    wmax = float(w.abs().max()) if w.numel() else 0.0
    if os.environ.get('ROMP_STEM', '') == 'valu' or not (wmax * 256.0 < 65504.0):
        op.flags |= OPF_STEM_VALU
    P.ops.append(op); P.names.append(name)
    P.flops.append(2.0 * (H // 2) * (W // 2) * 64 * 147)
    P.bytes.append(4.0 * (H * W * 3 + (H // 2) * (W // 2) * 64))
    return out


def _maxpool(P: Program, name, x: Act):
    out = P.new_act(x.C, x.H // 2, x.W // 2)
    op = RompOp()
    op.kind, op.in_buf, op.out_buf, op.res_buf = OP_MAXPOOL, x.buf, out.buf, BUF_NONE
    op.H, op.W, op.Cin, op.Cout, op.ksize, op.stride = x.H, x.W, x.C, x.C, 3, 2
    op.in_cstride, op.out_cstride = x.cstride, out.cstride
    P.ops.append(op); P.names.append(name)
    P.flops.append(9.0 * out.H * out.W * x.C)
    P.bytes.append(4.0 * (x.H * x.W * x.C + out.H * out.W * x.C))
    return out


def build_romp_resnet50(sd, device, input_size=512, bf16x3=False, split_k_items=0) -> Program:
    sd = _clean(sd)
    P = Program(device)
    P.split_k_items = split_k_items                           # > 0: single-image plan (plan.Program.conv)
    set_conv_math(P, bf16x3)
    bb = 'backbone.'

    def cbr(name, x, conv, bn, k, stride, relu, res=None):
        w = sd[conv + '.weight']
        s, b = fold_bn(sd, bn, w.shape[0])
        return P.conv(name, x, [w], [s], [b], k, stride, relu, res=res)

    s, b = fold_bn(sd, bb + 'bn1', 64)
    x = _stem7(P, 'stem.conv1', sd[bb + 'conv1.weight'], s, b, input_size, input_size)
    y = _maxpool(P, 'stem.maxpool', x)
    P.free(x)
    x = y
    for li, (planes, blocks, stride) in enumerate(LAYERS, 1):
        for i in range(blocks):
            p = f'{bb}layer{li}.{i}.'
            st = stride if i == 0 else 1
            t1 = cbr(p + 'conv1', x, p + 'conv1', p + 'bn1', 1, 1, True)
            t2 = cbr(p + 'conv2', t1, p + 'conv2', p + 'bn2', 3, st, True)
            P.free(t1)
            if (p + 'downsample.0.weight') in sd:
                r = cbr(p + 'downsample', x, p + 'downsample.0', p + 'downsample.1', 1, st, False)
                P.free(x)
            else:
                r = x
            y = cbr(p + 'conv3', t2, p + 'conv3', p + 'bn3', 1, 1, True, res=r)
            P.free(t2)
            P.free(r)
            x = y
    # ---- three transposed convs (resnet_50.py:93-120), the last one straight into the head input buffer
    fs = input_size // 4
    P.head_in_buf = P.alloc(RESNET_HEAD_CH * fs * fs, persistent=True)
    P.head_in_ch, P.coord_off = RESNET_HEAD_CH, 64
    KY = ((3, 1), (2, 0))                                   # kernel taps seen by output parity a: (tap of input y-1+a, tap of input y+a)
    for d in range(3):
        w = sd[f'{bb}deconv_layers.{3 * d}.weight']         # (Cin, Cout, 4, 4)
        co = w.shape[1]
        s, b = fold_bn(sd, f'{bb}deconv_layers.{3 * d + 1}', co)
        H2, W2 = 2 * x.H, 2 * x.W
        if d == 2:
            out = Act(P.head_in_buf, co, H2, W2, RESNET_HEAD_CH)
        else:
            out = P.new_act(co, H2, W2)
        # the four parities are independent and write disjoint (interleaved) elements: one stream each -- a parity conv of the first
        # layer (2048 -> 256 @16^2) is 128-256 work items on 512 workgroup slots, four of them fill the chip (round 6)
        P.fork(3)
        for a in range(2):
            for bpar in range(2):
                P.on(2 * a + bpar)
                w2 = torch.stack([torch.stack([w[:, :, KY[a][dy], KY[bpar][dx]] for dx in range(2)], -1) for dy in range(2)], -2)
                w2 = w2.permute(1, 0, 2, 3).contiguous()     # (Cout, Cin, 2, 2)
                P.conv(f'deconv{d}.p{a}{bpar}', x, [w2], [s], [b], 2, 1, True, out_buf_special=out.buf,
                       out_cstride=2 * out.cstride, out_coff=a * W2 * out.cstride + bpar * out.cstride,
                       pad=(1 - a, 1 - bpar), out_rstride=2 * W2 * out.cstride, out_bstride=H2 * W2 * out.cstride)
        P.join()
        P.free(x)
        x = out
    head_x = Act(P.head_in_buf, RESNET_HEAD_CH, fs, fs, RESNET_HEAD_CH)
    build_romp_head(P, sd, head_x, 66)
    return P
