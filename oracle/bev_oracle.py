"""CPU oracle for the BEV head / 3-D center parsing (BASELINE config 4) -- TEST INFRASTRUCTURE ONLY.

Restates ``simple_romp/bev/model.py`` (``BEVv1`` :104-250) and ``simple_romp/bev/post_parser.py``
(``CenterMap3D.parse_3dcentermap`` :44-66, ``pack_params_dict`` :240-253,
``denormalize_cam_params_to_trans`` :114-128, ``SMPLA_parser`` :255-278) on the CPU in float32.
Pinned against the reference itself by ``oracle/make_golden.py`` (fixture ``bev_b1.npz``).
Paths in citations are relative to ``/root/reference/simple_romp``.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import romp_oracle as R

BN_EPS = 1e-5
MAP = 128
DEPTH = 64


def cam3dmap_anchor(fov=60.0, size=MAP):
    """get_cam3dmap_anchor, bev/model.py:77-87 -> 64 scale anchors."""
    depth_level = np.array([1, 10, 20, 100], dtype=np.float32)
    rng = (np.array([2 / 64., 25 / 64., 3 / 64., 2 / 64.], dtype=np.float32) * size).astype(np.int32)
    scale_level = 1 / np.tan(np.radians(fov / 2.)) / depth_level
    out, cache = [], 8
    for scale, n in zip(scale_level, rng):
        out.append(cache - np.arange(1, n + 1) / n * (cache - scale))
        cache = scale
    return np.concatenate(out)


def coordmap_3d(size=MAP):
    """get_3Dcoord_maps_halfz, bev/model.py:9-17: (1,D,H,W,3) = [Z(anchor), Y, X]."""
    z = torch.from_numpy(cam3dmap_anchor(60, size)).float()
    r = torch.arange(size, dtype=torch.float32)
    Z = z.reshape(1, -1, 1, 1, 1).repeat(1, 1, size, size, 1)
    Y = r.reshape(1, 1, size, 1, 1).repeat(1, len(z), 1, size, 1) / size * 2 - 1
    X = r.reshape(1, 1, 1, size, 1).repeat(1, len(z), size, 1, 1) / size * 2 - 1
    return torch.cat([Z, Y, X], -1)


def bev_head_spec():
    """Ordered {key: (shape, kind)} of the BEVv1 head parameters (bev/model.py:115-186)."""
    sp = OrderedDict()

    def bn(n, c):
        for suf, kind in (('.weight', 'bn_w'), ('.bias', 'bn_b'), ('.running_mean', 'bn_m'), ('.running_var', 'bn_v')):
            sp[n + suf] = ((c,), kind)

    sp['position_embeddings.weight'] = ((128, 128), 'emb')
    for i, (co, ci) in zip((0, 3, 6), ((512, 128), (512, 512), (143, 512))):
        sp[f'transformer.{i}.weight'] = ((co, ci), 'lin_w')
        sp[f'transformer.{i}.bias'] = ((co,), 'conv_b')
    for head in ('det_head', 'param_head'):
        p = f'{head}.0.0.'
        sp[p + 'conv1.weight'] = ((128, 32, 3, 3), 'conv_w'); bn(p + 'bn1', 128)
        sp[p + 'conv2.weight'] = ((128, 128, 3, 3), 'conv_w'); bn(p + 'bn2', 128)
        sp[p + 'downsample.weight'] = ((128, 32, 1, 1), 'conv_w'); sp[p + 'downsample.bias'] = ((128,), 'conv_b')
        if head == 'det_head':
            sp['det_head.1.weight'] = ((4, 128, 1, 1), 'conv_w'); sp['det_head.1.bias'] = ((4,), 'conv_b')
    for i, (k, ci) in zip((0, 3, 6), ((1, 32), (3, 16), (1, 16))):
        sp[f'bv_pre_layers.{i}.weight'] = ((16, ci, k, k), 'conv_w'); sp[f'bv_pre_layers.{i}.bias'] = ((16,), 'conv_b')
        bn(f'bv_pre_layers.{i + 1}', 16)
    for i, (ci, co) in enumerate(((2560, 512), (512, 512), (512, 128))):
        p = f'bv_out_layers.{i}.'
        sp[p + 'conv1.weight'] = ((co, ci, 3), 'conv_w'); bn(p + 'bn1', co)
        sp[p + 'conv2.weight'] = ((co, co, 3), 'conv_w'); bn(p + 'bn2', co)
    for name, c in (('center_map_refiner', 1), ('cam_map_refiner', 3)):
        p = f'{name}.0.'
        sp[p + 'conv1.weight'] = ((c, c, 3, 3, 3), 'conv_w'); bn(p + 'bn1', c)
        sp[p + 'conv2.weight'] = ((c, c, 3, 3, 3), 'conv_w'); bn(p + 'bn2', c)
    return sp


def make_bev_state_dict(seed=0, center_gain=1.0, center_bias=1.5):
    """Synthetic BEVv1 weights: seeded ROMP backbone (oracle recipe) + seeded head.  `center_gain`
    scales the front-view center output so that a few 3-D centers pass the positive threshold."""
    sd = OrderedDict((k, v) for k, v in R.make_romp_state_dict(seed).items() if k.startswith('backbone.'))
    g = torch.Generator().manual_seed(seed + 1000)
    for k, (shp, kind) in bev_head_spec().items():
        if kind == 'bn_m':
            v = torch.randn(shp, generator=g) * 0.1
        elif kind == 'bn_v':
            v = torch.rand(shp, generator=g) + 0.5
        elif kind == 'bn_w':
            v = torch.rand(shp, generator=g) * 0.4 + 0.8
        elif kind == 'bn_b':
            v = torch.randn(shp, generator=g) * 0.1
        elif kind in ('conv_w', 'lin_w'):
            fan_in = int(np.prod(shp[1:]))
            v = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(fan_in)
        elif kind == 'emb':
            v = torch.randn(shp, generator=g) * 0.1
        else:
            v = (torch.rand(shp, generator=g) * 2 - 1) * 0.05
        sd[k] = v.float().contiguous()
    sd['det_head.1.weight'][0] *= center_gain
    sd['det_head.1.bias'][0] += center_bias           # front-view center map positive (raw conv output, no sigmoid)
    sd['coordmap_3d'] = coordmap_3d(MAP)
    return sd


def _bn(x, sd, name):
    return F.batch_norm(x, sd[name + '.running_mean'], sd[name + '.running_var'], sd[name + '.weight'], sd[name + '.bias'],
                        False, 0.0, BN_EPS)


def _head_block(x, sd, p):
    """BasicBlock(32,128, downsample=Conv2d 1x1 with bias, no BN): bev/model.py:154-157, romp/model.py:67-83."""
    y = torch.relu(_bn(F.conv2d(x, sd[p + 'conv1.weight'], None, padding=1), sd, p + 'bn1'))
    y = _bn(F.conv2d(y, sd[p + 'conv2.weight'], None, padding=1), sd, p + 'bn2')
    r = F.conv2d(x, sd[p + 'downsample.weight'], sd[p + 'downsample.bias'])
    return torch.relu(y + r)


def _block1d(x, sd, p):
    """BasicBlock_1D (no residual), bev/model.py:24-45."""
    y = torch.relu(_bn(F.conv1d(x, sd[p + 'conv1.weight'], None, padding=1), sd, p + 'bn1'))
    return torch.relu(_bn(F.conv1d(y, sd[p + 'conv2.weight'], None, padding=1), sd, p + 'bn2'))


def _block3d(x, sd, p):
    """BasicBlock_3D (residual, no final ReLU), bev/model.py:52-75."""
    y = torch.relu(_bn(F.conv3d(x, sd[p + 'conv1.weight'], None, padding=1), sd, p + 'bn1'))
    y = _bn(F.conv3d(y, sd[p + 'conv2.weight'], None, padding=1), sd, p + 'bn2')
    return y + x


@torch.no_grad()
def coarse2fine_localization(sd, x):
    """bev/model.py:188-215.  x: backbone output (B,32,128,128).
    -> center_maps_3d (B,64,128,128), cam_maps_3d (B,3,64,128,128), center_maps_fv (B,1,128,128)."""
    maps_fv = F.conv2d(_head_block(x, sd, 'det_head.0.0.'), sd['det_head.1.weight'], sd['det_head.1.bias'])
    center_fv, cam_off = maps_fv[:, :1], maps_fv[:, 1:4]
    f = x
    for i, pad in zip((0, 3, 6), (0, 1, 0)):
        f = torch.relu(_bn(F.conv2d(f, sd[f'bv_pre_layers.{i}.weight'], sd[f'bv_pre_layers.{i}.bias'], padding=pad),
                           sd, f'bv_pre_layers.{i + 1}'))
    summon = torch.cat([center_fv, cam_off, f], 1).reshape(x.shape[0], -1, MAP)          # :190
    bv = summon
    for i in range(3):
        bv = _block1d(bv, sd, f'bv_out_layers.{i}.')
    center_bv, cam_off_bv = bv[:, :DEPTH], bv[:, DEPTH:]
    center_3d = center_fv.repeat(1, DEPTH, 1, 1) * center_bv.unsqueeze(2).repeat(1, 1, MAP, 1)   # :195-196
    center_3d = _block3d(center_3d.unsqueeze(1), sd, 'center_map_refiner.0.').squeeze(1)
    cam_3d = sd['coordmap_3d'] + cam_off.unsqueeze(-1).transpose(4, 1).contiguous()              # :209-210
    cam_3d[:, :, :, :, 2] = cam_3d[:, :, :, :, 2] + cam_off_bv.unsqueeze(2).contiguous()         # :212
    cam_3d = _block3d(cam_3d.unsqueeze(1).transpose(5, 1).squeeze(-1), sd, 'cam_map_refiner.0.')
    return center_3d, cam_3d, center_fv


def parse_3dcentermap(center_maps_3d, conf_thresh, max_person=64):
    """CenterMap3D.parse_3dcentermap, bev/post_parser.py:44-66.  Order: batch-major, score
    descending (ties: lower flat zyx index first; the reference leaves ties unspecified)."""
    cm = torch.as_tensor(center_maps_3d).float()
    mx = F.max_pool3d(cm.unsqueeze(1), 5, 1, 2).squeeze(1)
    nmsd = (cm * (mx == cm).float()).numpy()
    B = cm.shape[0]
    bids, zyx, scs = [], [], []
    for b in range(B):
        flat = nmsd[b].reshape(-1)
        cand = np.nonzero(flat > conf_thresh)[0]
        order = cand[np.lexsort((cand, -flat[cand]))][:max_person]
        bids.append(np.full(order.size, b, np.int64))
        zyx.append(np.stack([order // (MAP * MAP), (order // MAP) % MAP, order % MAP], 1).astype(np.int64))
        scs.append(flat[order])
    return np.concatenate(bids), np.concatenate(zyx), np.concatenate(scs)


def cam_to_czyx(cams):
    """convert_cam_params_to_centermap_coords + denormalize_center, bev/model.py:89-102."""
    cams = np.asarray(cams, np.float32)
    anchor = cam3dmap_anchor(60, MAP).astype(np.float32)
    k = np.argmin(np.abs(cams[:, :1] - anchor[None]), 1).astype(np.float32)
    coords = np.concatenate([(k / 128 * 2. - 1.)[:, None], cams[:, 1:]], 1).astype(np.float32)
    c = (coords + 1) / 2 * MAP
    return np.clip(c, 1, MAP - 1).astype(np.int64)


@torch.no_grad()
def bev_forward(sd, image_nhwc, conf_thresh):
    """BEVv1.forward, bev/model.py:232-250 -> dict (numpy) or None."""
    x = R.backbone_forward(sd, image_nhwc)
    c3d, cam3d, c_fv = coarse2fine_localization(sd, x)
    bids, czyx, confs = parse_3dcentermap(c3d, conf_thresh)
    if bids.size == 0:
        return None
    cams = cam3d.numpy()[bids, :, czyx[:, 0], czyx[:, 1], czyx[:, 2]]                       # :242
    fv = _head_block(x, sd, 'param_head.0.0.').numpy()
    cam_czyx = cam_to_czyx(cams)
    feat = fv[bids, :, cam_czyx[:, 1], cam_czyx[:, 2]] + sd['position_embeddings.weight'].numpy()[cam_czyx[:, 0]]
    h = torch.from_numpy(feat)
    h = torch.relu(F.linear(h, sd['transformer.0.weight'], sd['transformer.0.bias']))
    h = torch.relu(F.linear(h, sd['transformer.3.weight'], sd['transformer.3.bias']))
    h = F.linear(h, sd['transformer.6.weight'], sd['transformer.6.bias'])
    params_pred = np.concatenate([cams, h.numpy()], 1).astype(np.float32)                   # (N,146)
    return {'params_pred': params_pred, 'cam_czyx': cam_czyx, 'center_map_3d': c3d.numpy(), 'cam_maps_3d': cam3d.numpy(),
            'center_map_fv': c_fv.numpy(), 'pred_batch_ids': bids, 'pred_czyxs': czyx, 'center_confs': confs}


def pack_params(params_pred):
    """pack_params_dict, bev/post_parser.py:240-253 (11 betas)."""
    p = np.asarray(params_pred, np.float32)
    go = R.rot6d_to_angular(p[:, 3:9])
    body = R.rot6d_to_angular(p[:, 9:135])
    thetas = np.concatenate([go, body, np.zeros((p.shape[0], 6), np.float32)], 1)
    return {'cam': p[:, :3].copy(), 'smpl_thetas': thetas, 'smpl_betas': p[:, 135:146].copy()}


TAN_FOV = np.tan(np.radians(60 / 2.))


def cam_to_trans(cams):
    """denormalize_cam_params_to_trans, bev/post_parser.py:109-128."""
    cams = np.asarray(cams, np.float32)
    depth = (1 / (cams[:, 0] * TAN_FOV + 1e-3))[:, None]
    xy = cams[:, 1:][:, ::-1] * depth * TAN_FOV
    return np.concatenate([xy, depth], 1).astype(np.float32)


def smpla_forward(smpla_model, smil_model, betas, thetas, baby_thresh=0.8):
    """SMPLA_parser.forward, bev/post_parser.py:255-278 (root_align=True)."""
    betas, thetas = np.asarray(betas, np.float32), np.asarray(thetas, np.float32)
    N = betas.shape[0]
    verts, joints = np.zeros((N, 6890, 3), np.float32), np.zeros((N, 71, 3), np.float32)
    baby = betas[:, 10] > baby_thresh
    if baby.any():
        verts[baby], joints[baby], _ = R.smpl_forward(smil_model, betas[baby, :10], thetas[baby])
    if (~baby).any():
        verts[~baby], joints[~baby], _ = R.smpl_forward(smpla_model, betas[~baby], thetas[~baby])
    root = joints[:, [45, 46]].mean(1, keepdims=True)
    return verts - root, joints - root
