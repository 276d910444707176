"""Lower a BEV state_dict (HRNet-32 + BEV head) to the layer program of libromp_hip.so.

Host-side mirror of ``simple_romp/bev/model.py``: ``BEVv1._build_head`` :115-186 and the
dataflow of ``coarse2fine_localization`` :199-215 / ``fv_conditioned_bv_estimation`` :188-197.
The program ends with ``center_maps_3d (B,64,128,128)`` and ``cam_maps_3d (B,3,64,128,128)`` in
the caller's output tensors and leaves the front-view feature map of ``param_head`` in a
persistent arena buffer for ``romp_bev_regress``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .lib import BUF_CENTER, BUF_NONE, BUF_PARAMS, OP_BEV_MAPS, OP_BEV_PACK, OP_CONV3D, RompOp
from .plan import Act, Program, _clean, build_hrnet32_backbone, fold_bn, set_conv_math

MAP, DEPTH, VOX = 128, 64, 64 * 128 * 128


def cam3dmap_anchor(fov=60.0, size=MAP):
    """get_cam3dmap_anchor (bev/model.py:77-87): 64 scale anchors, computed exactly as the reference
    computes them (numpy), then cast to float32 like ``torch.from_numpy(...).float()`` (:127)."""
    depth_level = np.array([1, 10, 20, 100], dtype=np.float32)
    rng = (np.array([2 / 64., 25 / 64., 3 / 64., 2 / 64.], dtype=np.float32) * size).astype(np.int32)
    scale_level = 1 / np.tan(np.radians(fov / 2.)) / depth_level
    out, cache = [], 8
    for scale, n in zip(scale_level, rng):
        out.append(cache - np.arange(1, n + 1) / n * (cache - scale))
        cache = scale
    return np.concatenate(out).astype(np.float32)


def _host_floats(P: Program, values):
    """Small constant table kept alive on the HOST (passed to the kernel by value at launch)."""
    arr = (C.c_float * len(values))(*[float(v) for v in values])
    P.consts.append(arr)
    return C.cast(arr, C.c_void_p).value


def build_bev_hrnet32(sd, device, input_size=512, bf16x3=False, split_k_items=0) -> Program:
    assert input_size == 512, 'the BEV head is defined on a 128x128 map (bev/model.py:117)'
    sd = _clean(sd)
    P = Program(device)
    P.split_k_items = split_k_items                           # > 0: single-image plan (plan.Program.conv)
    set_conv_math(P, bf16x3)
    x = build_hrnet32_backbone(P, sd, input_size, out_cstride=32)          # (B,128,128,32)

    def bn(name, c, bias=None):
        return fold_bn(sd, name, c, bias)

    # ---- det_head / param_head first convs share x: one 32->256 conv (bev/model.py:154-157)
    heads = ('det_head', 'param_head')
    w1 = torch.cat([sd[f'{h}.0.0.conv1.weight'] for h in heads], 0)
    s1 = torch.cat([bn(f'{h}.0.0.bn1', 128)[0] for h in heads])
    b1 = torch.cat([bn(f'{h}.0.0.bn1', 128)[1] for h in heads])
    t = P.conv('bev.heads.conv1', x, [w1], [s1], [b1], 3, 1, True)
    # downsample = Conv2d 1x1 WITH bias, no BN (bev/model.py:156): the residual branch
    wd = torch.cat([sd[f'{h}.0.0.downsample.weight'] for h in heads], 0)
    bd = torch.cat([sd[f'{h}.0.0.downsample.bias'] for h in heads])
    r = P.conv('bev.heads.downsample', x, [wd], [torch.ones(256)], [bd], 1, 1, False)
    # conv2 + bn2 + residual + ReLU as a 2-group conv; output persistent (param_head features are
    # read by romp_bev_regress after the program finished)
    P.fv_buf = P.alloc(256 * MAP * MAP, persistent=True)
    P.exported_bufs.add(P.fv_buf)
    y = Act(P.fv_buf, 256, MAP, MAP, 256)
    P.conv('bev.heads.conv2', t, [sd[f'{h}.0.0.conv2.weight'] for h in heads], [bn(f'{h}.0.0.bn2', 128)[0] for h in heads],
           [bn(f'{h}.0.0.bn2', 128)[1] for h in heads], 3, 1, True, res=r, out=y, groups=2)
    P.free(t)
    P.free(r)
    P.fv_coff, P.fv_cstride = 128, 256
    # det_head output conv 128 -> 4 (center, 3 cam offsets), bias, no BN (bev/model.py:159-160)
    maps_fv = P.conv('bev.det_out', Act(y.buf, 128, MAP, MAP, 256, 0), [sd['det_head.1.weight']], [torch.ones(4)],
                     [sd['det_head.1.bias']], 1, 1, False)
    # ---- bv_pre_layers: 1x1 / 3x3 / 1x1, all with bias + BN + ReLU (bev/model.py:166-175)
    f = x
    for i, k in zip((0, 3, 6), (1, 3, 1)):
        w = sd[f'bv_pre_layers.{i}.weight']
        s, b = bn(f'bv_pre_layers.{i + 1}', 16, sd[f'bv_pre_layers.{i}.bias'])
        g = P.conv(f'bev.bv_pre.{i}', f, [w], [s], [b], k, 1, True)
        if f is not x:
            P.free(f)
        f = g
    # ---- summon_feats (bev/model.py:190): (B, (1+3+16)*128, 128) as (B, W=128, 2560) for the Conv1d stack
    packed = P.new_act(20 * MAP, 1, MAP)
    op = RompOp()
    op.kind, op.in_buf, op.res_buf, op.out_buf = OP_BEV_PACK, maps_fv.buf, f.buf, packed.buf
    op.in_cstride, op.res_cstride = maps_fv.cstride, f.cstride
    P.ops.append(op); P.names.append('bev.pack'); P.flops.append(0.0); P.bytes.append(4.0 * 2 * 20 * MAP * MAP)
    P.free(f)
    # ---- bv_out_layers: 3 x BasicBlock_1D = 6 x (Conv1d k3 + BN1d + ReLU), no residual (bev/model.py:24-45,179-182)
    h = packed
    for i in range(3):
        for cv, bnn in (('conv1', 'bn1'), ('conv2', 'bn2')):
            w = sd[f'bv_out_layers.{i}.{cv}.weight']
            s, b = bn(f'bv_out_layers.{i}.{bnn}', w.shape[0])
            g = P.conv(f'bev.bv_out.{i}.{cv}', h, [w], [s], [b], 13, 1, True)
            P.free(h)
            h = g
    # ---- outer product + coordinate maps (bev/model.py:195-196, 209-212)
    c3 = P.new_act(1, DEPTH * MAP, MAP)
    cam = P.new_act(3, DEPTH * MAP, MAP)
    op = RompOp()
    op.kind, op.in_buf, op.res_buf, op.out_buf = OP_BEV_MAPS, maps_fv.buf, h.buf, c3.buf
    op.in_cstride, op.res_cstride = maps_fv.cstride, h.cstride
    op.term_buf[0] = cam.buf
    op.weight = _host_floats(P, cam3dmap_anchor(60, MAP))
    P.ops.append(op); P.names.append('bev.maps'); P.flops.append(float(VOX)); P.bytes.append(4.0 * 4 * VOX)
    P.free(h)
    P.free(maps_fv)

    # ---- 3-D refiners: BasicBlock_3D = conv-bn-relu-conv-bn + residual, no final ReLU (bev/model.py:52-75)
    def conv3d(name, xin, prefix, cv, bnn, ch, relu, res, out_special):
        w = sd[f'{prefix}.0.{cv}.weight']
        s, b = bn(f'{prefix}.0.{bnn}', ch)
        out = None if out_special is not None else P.new_act(ch, DEPTH * MAP, MAP)
        o = RompOp()
        o.kind, o.in_buf, o.res_buf = OP_CONV3D, xin.buf, (res.buf if res is not None else BUF_NONE)
        o.out_buf = out_special if out_special is not None else out.buf
        o.Cin = o.Cout = ch
        o.relu = int(relu)
        o.weight = _host_floats(P, w.reshape(-1).tolist())
        o.scale, o.shift = _host_floats(P, s.tolist()), _host_floats(P, b.tolist())
        P.ops.append(o); P.names.append(name); P.flops.append(2.0 * VOX * ch * ch * 27)
        P.bytes.append(4.0 * VOX * ch * (3 if res is not None else 2))
        return out

    t1 = conv3d('bev.center_refine.conv1', c3, 'center_map_refiner', 'conv1', 'bn1', 1, True, None, None)
    conv3d('bev.center_refine.conv2', t1, 'center_map_refiner', 'conv2', 'bn2', 1, False, c3, BUF_CENTER)
    P.free(t1)
    t3 = conv3d('bev.cam_refine.conv1', cam, 'cam_map_refiner', 'conv1', 'bn1', 3, True, None, None)
    conv3d('bev.cam_refine.conv2', t3, 'cam_map_refiner', 'conv2', 'bn2', 3, False, cam, BUF_PARAMS)
    P.free(t3)
    P.free(c3)
    P.free(cam)
    return P
