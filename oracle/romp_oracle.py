"""CPU oracle for the ROMP inference hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement of the reference algorithm (Arthur151/ROMP,
``simple_romp/romp``).  It is the *checker* for the HIP path: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product package ``romp_amd`` never imports anything from ``oracle/``.

Pinning: the reference ships no golden vectors for this path.
The oracle is therefore pinned against *outputs of the reference itself*, generated in
the build container by ``oracle/make_golden.py`` (which imports the reference's own
``model.py`` / ``smpl.py`` / ``post_parser.py`` by file path) and committed under
``tests/golden/``.  ``tests/test_oracle_golden.py`` re-checks that pin on every run.

Arithmetic: float32 throughout (the reference's dtype).  Dense convolutions use
``torch.nn.functional.conv2d`` on CPU (the same ATen kernels the reference executes);
everything else (BN, parsing, rotation conversion, SMPL) is restated explicitly in
numpy/torch elementwise ops.

Each function cites the reference file:line it follows (paths relative to
``/root/reference/simple_romp/romp``).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# HRNet-32 + ROMP head structure (model.py:336-380, 427-468)
# --------------------------------------------------------------------------------------
STAGE_CFG = {
    2: dict(modules=1, channels=[32, 64]),
    3: dict(modules=4, channels=[32, 64, 128]),
    4: dict(modules=3, channels=[32, 64, 128, 256]),
}
HEAD_OUT = {1: 142, 2: 1, 3: 3}  # final_layers index -> out channels (model.py:436-443)
BN_EPS = 1e-5


def _bn_keys(prefix):
    return [prefix + s for s in ('.weight', '.bias', '.running_mean', '.running_var')]


def romp_param_spec():
    """Ordered {state_dict key: (shape, kind)} of ROMPv1 (HRNet-32 + head), float tensors
    only (``num_batches_tracked`` omitted); kind in conv_w|conv_b|bn_w|bn_b|bn_m|bn_v.
    Restates the constructors at model.py:246-380 and :427-468; checked key-for-key
    against the reference module in ``make_golden.py``."""
    sh = OrderedDict()

    def conv(name, cout, cin, k, bias=False):
        sh[name + '.weight'] = ((cout, cin, k, k), 'conv_w')
        if bias:
            sh[name + '.bias'] = ((cout,), 'conv_b')

    def bn(name, c):
        for k, kind in zip(_bn_keys(name), ('bn_w', 'bn_b', 'bn_m', 'bn_v')):
            sh[k] = ((c,), kind)

    b = 'backbone.'
    conv(b + 'conv1', 64, 3, 3); bn(b + 'bn1', 64)
    conv(b + 'conv2', 64, 64, 3); bn(b + 'bn2', 64)
    inpl = 64
    for i in range(4):  # layer1: 4 Bottlenecks, planes 64 (model.py:345)
        p = f'{b}layer1.{i}.'
        conv(p + 'conv1', 64, inpl, 1); bn(p + 'bn1', 64)
        conv(p + 'conv2', 64, 64, 3); bn(p + 'bn2', 64)
        conv(p + 'conv3', 256, 64, 1); bn(p + 'bn3', 256)
        if i == 0:
            conv(p + 'downsample.0', 256, 64, 1); bn(p + 'downsample.1', 256)
        inpl = 256
    # transitions (model.py:254-287)
    for s, cfg in STAGE_CFG.items():
        ch = cfg['channels']
        nb = len(ch)
        if s == 2:
            conv(b + 'transition1.0.0', 32, 256, 3); bn(b + 'transition1.0.1', 32)
            conv(b + 'transition1.1.0.0', 64, 256, 3); bn(b + 'transition1.1.0.1', 64)
        elif s == 3:
            conv(b + 'transition2.2.0.0', 128, 64, 3); bn(b + 'transition2.2.0.1', 128)
        else:
            conv(b + 'transition3.3.0.0', 256, 128, 3); bn(b + 'transition3.3.0.1', 256)
        for m in range(cfg['modules']):
            p = f'{b}stage{s}.{m}.'
            for br in range(nb):
                for k in range(4):
                    q = f'{p}branches.{br}.{k}.'
                    conv(q + 'conv1', ch[br], ch[br], 3); bn(q + 'bn1', ch[br])
                    conv(q + 'conv2', ch[br], ch[br], 3); bn(q + 'bn2', ch[br])
            n_out = 1 if (s == 4 and m == cfg['modules'] - 1) else nb  # model.py:316-320
            for i in range(n_out):
                for j in range(nb):
                    q = f'{p}fuse_layers.{i}.{j}.'
                    if j > i:
                        conv(q + '0', ch[i], ch[j], 1); bn(q + '1', ch[i])
                    elif j < i:
                        for k in range(i - j):
                            co = ch[i] if k == i - j - 1 else ch[j]
                            conv(f'{q}{k}.0', co, ch[j], 3); bn(f'{q}{k}.1', co)
    for h, co in HEAD_OUT.items():  # model.py:445-468
        p = f'final_layers.{h}.'
        conv(p + '0.0', 64, 34, 3, bias=True); bn(p + '0.1', 64)
        for blk in range(2):
            q = f'{p}1.{blk}.0.'
            conv(q + 'conv1', 64, 64, 3); bn(q + 'bn1', 64)
            conv(q + 'conv2', 64, 64, 3); bn(q + 'bn2', 64)
        conv(p + '2', co, 64, 1, bias=True)
    return sh


def make_romp_state_dict(seed=0, center_bias=0.0):
    """Deterministic synthetic ROMPv1 weights (the licensed ROMP.pkl is unavailable here).
    conv/bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)); every BatchNorm gets non-trivial
    statistics so that BN folding is exercised (SURVEY.md §8d recipe).  `center_bias` is added
    to the center head's output bias so that synthetic center maps become positive and a
    positive threshold selects a few persons per image."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, (shp, kind) in romp_param_spec().items():
        if kind == 'bn_m':
            v = torch.randn(shp, generator=g) * 0.1
        elif kind == 'bn_v':
            v = torch.rand(shp, generator=g) + 0.5
        elif kind == 'bn_w':
            v = torch.rand(shp, generator=g) * 0.4 + 0.8
        elif kind == 'bn_b':
            v = torch.randn(shp, generator=g) * 0.1
        elif kind == 'conv_w':
            bound = 1.0 / math.sqrt(shp[1] * shp[2] * shp[3])
            v = (torch.rand(shp, generator=g) * 2 - 1) * bound
        else:                                               # conv bias (head convs)
            v = (torch.rand(shp, generator=g) * 2 - 1) * 0.05
        sd[k] = v.float().contiguous()
    if center_bias:
        sd['final_layers.2.2.bias'] = sd['final_layers.2.2.bias'] + float(center_bias)
    return sd


def make_images(batch, seed=1):
    """Already-preprocessed synthetic input: float32 (B,512,512,3) in 0..255."""
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (batch, 512, 512, 3), generator=g).float()


# --------------------------------------------------------------------------------------
# network forward (functional restatement)
# --------------------------------------------------------------------------------------
def _bn(x, sd, name):
    """Inference BatchNorm2d, eps=1e-5 (nn.BatchNorm2d default; model.py:60)."""
    w, b = sd[name + '.weight'], sd[name + '.bias']
    m, v = sd[name + '.running_mean'], sd[name + '.running_var']
    return F.batch_norm(x, m, v, w, b, False, 0.0, BN_EPS)


def _conv(x, sd, name, stride=1):
    w = sd[name + '.weight']
    return F.conv2d(x, w, sd.get(name + '.bias'), stride=stride, padding=w.shape[-1] // 2)


def _basic_block(x, sd, p):
    """BasicBlock.forward, model.py:67-83."""
    y = torch.relu(_bn(_conv(x, sd, p + 'conv1'), sd, p + 'bn1'))
    y = _bn(_conv(y, sd, p + 'conv2'), sd, p + 'bn2')
    return torch.relu(y + x)


def _bottleneck(x, sd, p):
    """Bottleneck.forward, model.py:103-123."""
    y = torch.relu(_bn(_conv(x, sd, p + 'conv1'), sd, p + 'bn1'))
    y = torch.relu(_bn(_conv(y, sd, p + 'conv2'), sd, p + 'bn2'))
    y = _bn(_conv(y, sd, p + 'conv3'), sd, p + 'bn3')
    r = x
    if (p + 'downsample.0.weight') in sd:
        r = _bn(_conv(x, sd, p + 'downsample.0'), sd, p + 'downsample.1')
    return torch.relu(y + r)


def _hr_module(xs, sd, p, n_out):
    """HighResolutionModule.forward, model.py:226-244."""
    nb = len(xs)
    xs = list(xs)
    for br in range(nb):
        for k in range(4):
            xs[br] = _basic_block(xs[br], sd, f'{p}branches.{br}.{k}.')
    outs = []
    for i in range(n_out):
        y = None
        for j in range(nb):
            if j == i:
                t = xs[j]
            elif j > i:   # 1x1 conv + BN + nearest upsample (model.py:188-197)
                q = f'{p}fuse_layers.{i}.{j}.'
                t = _bn(_conv(xs[j], sd, q + '0'), sd, q + '1')
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode='nearest')
            else:         # chain of stride-2 3x3 convs (model.py:200-218)
                t = xs[j]
                for k in range(i - j):
                    q = f'{p}fuse_layers.{i}.{j}.{k}.'
                    t = _bn(_conv(t, sd, q + '0', stride=2), sd, q + '1')
                    if k != i - j - 1:
                        t = torch.relu(t)
            y = t if y is None else y + t
        outs.append(torch.relu(y))
    return outs


def coord_maps(size=128):
    """get_coord_maps, model.py:8-37: channel 0 varies along W, channel 1 along H."""
    r = torch.arange(size, dtype=torch.float32) / (size - 1) * 2 - 1
    xx = r.view(1, 1, 1, size).expand(1, 1, size, size)
    yy = r.view(1, 1, size, 1).expand(1, 1, size, size)
    return torch.cat([xx, yy], 1)


@torch.no_grad()
def backbone_forward(sd, image_nhwc):
    """HigherResolutionNet.forward, model.py:382-417.  image: (B,512,512,3) 0..255."""
    b = 'backbone.'
    x = (image_nhwc.permute(0, 3, 1, 2) / 255.) * 2.0 - 1.0
    x = torch.relu(_bn(_conv(x.contiguous(), sd, b + 'conv1', 2), sd, b + 'bn1'))
    x = torch.relu(_bn(_conv(x, sd, b + 'conv2', 2), sd, b + 'bn2'))
    for i in range(4):
        x = _bottleneck(x, sd, f'{b}layer1.{i}.')
    tr = lambda t, n, s: torch.relu(_bn(_conv(t, sd, n + '.0', s), sd, n + '.1'))
    xs = [tr(x, b + 'transition1.0', 1), tr(x, b + 'transition1.1.0', 2)]
    ys = _hr_module(xs, sd, b + 'stage2.0.', 2)
    xs = [ys[0], ys[1], tr(ys[-1], b + 'transition2.2.0', 2)]
    for m in range(4):
        xs = _hr_module(xs, sd, f'{b}stage3.{m}.', 3)
    xs = [xs[0], xs[1], xs[2], tr(xs[-1], b + 'transition3.3.0', 2)]
    for m in range(3):
        xs = _hr_module(xs, sd, f'{b}stage4.{m}.', 1 if m == 2 else 4)
    return xs[0]


@torch.no_grad()
def head_forward(sd, feat):
    """ROMPv1.forward after the backbone, model.py:472-481."""
    x = torch.cat([feat, coord_maps(128).repeat(feat.shape[0], 1, 1, 1)], 1)
    outs = {}
    for h in (1, 2, 3):
        p = f'final_layers.{h}.'
        y = torch.relu(_bn(_conv(x, sd, p + '0.0', 2), sd, p + '0.1'))
        for blk in range(2):
            y = _basic_block(y, sd, f'{p}1.{blk}.0.')
        outs[h] = _conv(y, sd, p + '2')
    center_maps = outs[2]
    params_maps = torch.cat([outs[3], outs[1]], 1)
    return center_maps, params_maps


@torch.no_grad()
def romp_net_forward(sd, image_nhwc):
    """-> center_maps (B,1,64,64), params_maps (B,145,64,64), NCHW like the reference."""
    return head_forward(sd, backbone_forward(sd, image_nhwc))


# --------------------------------------------------------------------------------------
# center-map parsing (post_parser.py:27-64, 128-146; main.py:113)
# --------------------------------------------------------------------------------------
def parse_centermap(center_maps, conf_thresh, max_person=64):
    """CenterMap.parse_centermap (post_parser.py:27-47).  center_maps: np (B,1,64,64).
    Returns batch_ids, flat_inds (int64), scores, ordered batch-major and
    score-descending within an image (ties: lower flat index first -- the reference's
    torch.topk leaves tie order unspecified)."""
    cm = np.asarray(center_maps, dtype=np.float32)
    B, _, H, W = cm.shape
    pad = np.full((B, H + 4, W + 4), -np.inf, np.float32)
    pad[:, 2:-2, 2:-2] = cm[:, 0]
    mx = np.full((B, H, W), -np.inf, np.float32)
    for dy in range(5):
        for dx in range(5):
            mx = np.maximum(mx, pad[:, dy:dy + H, dx:dx + W])
    nmsd = cm[:, 0] * (mx == cm[:, 0]).astype(np.float32)      # post_parser.py:50-54
    bids, finds, scs = [], [], []
    for b in range(B):
        flat = nmsd[b].reshape(-1)
        order = np.lexsort((np.arange(flat.size), -flat))[:max_person]  # top-K, stable
        keep = flat[order] > conf_thresh                                 # :44
        order = order[keep]
        bids.append(np.full(order.size, b, np.int64))
        finds.append(order.astype(np.int64))
        scs.append(flat[order])
    return np.concatenate(bids), np.concatenate(finds), np.concatenate(scs)


def _normalize(v, eps):
    """F.normalize(v, dim=-1, eps): v / max(||v||, eps)."""
    n = np.sqrt((v * v).sum(-1, keepdims=True, dtype=np.float32)).astype(np.float32)
    return v / np.maximum(n, np.float32(eps))


def rot6d_to_rotmat(x6):
    """utils.py:477-491.  x6: (N,6) -> view(-1,3,2): col0=a1, col1=a2 (interleaved)."""
    x = np.asarray(x6, np.float32).reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = _normalize(a1, 1e-6)
    dot = (b1 * a2).sum(-1, keepdims=True, dtype=np.float32)
    b2 = _normalize(a2 - dot * b1, 1e-6)
    b3 = np.cross(b1, b2).astype(np.float32)
    return np.stack([b1, b2, b3], -1)                          # columns b1 b2 b3


def rotmat_to_quat(R, eps=1e-6):
    """rotation_matrix_to_quaternion, utils.py:606-682 (note: works on R^T)."""
    R = np.asarray(R, np.float32)
    t = np.transpose(R, (0, 2, 1))
    m = lambda i, j: t[:, i, j]
    d2 = m(2, 2) < eps
    d01 = m(0, 0) > m(1, 1)
    d0n1 = m(0, 0) < -m(1, 1)
    t0 = 1 + m(0, 0) - m(1, 1) - m(2, 2)
    q0 = np.stack([m(1, 2) - m(2, 1), t0, m(0, 1) + m(1, 0), m(2, 0) + m(0, 2)], -1)
    t1 = 1 - m(0, 0) + m(1, 1) - m(2, 2)
    q1 = np.stack([m(2, 0) - m(0, 2), m(0, 1) + m(1, 0), t1, m(1, 2) + m(2, 1)], -1)
    t2 = 1 - m(0, 0) - m(1, 1) + m(2, 2)
    q2 = np.stack([m(0, 1) - m(1, 0), m(2, 0) + m(0, 2), m(1, 2) + m(2, 1), t2], -1)
    t3 = 1 + m(0, 0) + m(1, 1) + m(2, 2)
    q3 = np.stack([t3, m(1, 2) - m(2, 1), m(2, 0) - m(0, 2), m(0, 1) - m(1, 0)], -1)
    c0 = (d2 & d01).astype(np.float32)[:, None]
    c1 = (d2 & ~d01).astype(np.float32)[:, None]
    c2 = (~d2 & d0n1).astype(np.float32)[:, None]
    c3 = (~d2 & ~d0n1).astype(np.float32)[:, None]
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    den = t0[:, None] * c0 + t1[:, None] * c1 + t2[:, None] * c2 + t3[:, None] * c3
    with np.errstate(invalid='ignore', divide='ignore'):
        q = q / np.sqrt(den).astype(np.float32)
    return (q * np.float32(0.5)).astype(np.float32)


def quat_to_angle_axis(q):
    """quaternion_to_angle_axis, utils.py:554-604."""
    q = np.asarray(q, np.float32)
    q1, q2, q3 = q[:, 1], q[:, 2], q[:, 3]
    s2 = q1 * q1 + q2 * q2 + q3 * q3
    s = np.sqrt(s2).astype(np.float32)
    c = q[:, 0]
    two_theta = np.float32(2.0) * np.where(c < 0, np.arctan2(-s, -c), np.arctan2(s, c)).astype(np.float32)
    with np.errstate(invalid='ignore', divide='ignore'):
        k = np.where(s2 > 0, two_theta / s, np.float32(2.0)).astype(np.float32)
    return np.stack([q1 * k, q2 * k, q3 * k], -1).astype(np.float32)


def rot6d_to_angular(x):
    """rot6D_to_angular, utils.py:471-475 + NaN->0 of utils.py:551.  x: (N, J*6)."""
    N = x.shape[0]
    aa = quat_to_angle_axis(rotmat_to_quat(rot6d_to_rotmat(x.reshape(-1, 6))))
    aa[np.isnan(aa)] = 0.0
    return aa.reshape(N, -1)


def parsing_outputs(center_maps, params_maps, conf_thresh, max_person=64):
    """main.py:113 (1.1**s) + parsing_outputs, post_parser.py:135-146.
    center_maps (B,1,64,64), params_maps (B,145,64,64) numpy NCHW.  Returns dict or None."""
    cm = np.asarray(center_maps, np.float32)
    pm = np.array(params_maps, np.float32, copy=True)
    pm[:, 0] = np.power(np.float32(1.1), pm[:, 0]).astype(np.float32)
    bids, finds, scores = parse_centermap(cm, conf_thresh, max_person)
    if bids.size == 0:
        return None
    B, C = pm.shape[:2]
    params_pred = pm.reshape(B, C, -1).transpose(0, 2, 1)[bids, finds]      # (N,145)
    out = {
        'batch_ids': bids, 'flat_inds': finds, 'scores': scores, 'params_pred': params_pred,
        'cam': params_pred[:, 0:3].copy(),
        'global_orient': rot6d_to_angular(params_pred[:, 3:9]),
        'smpl_betas': params_pred[:, 135:145].copy(),
    }
    body = rot6d_to_angular(params_pred[:, 9:135])
    out['body_pose'] = np.concatenate([body, np.zeros((body.shape[0], 6), np.float32)], 1)
    out['smpl_thetas'] = np.concatenate([out['global_orient'], out['body_pose']], 1)
    out['center_preds'] = np.stack([finds % 64, finds // 64], 1) * 512 // 64
    out['center_confs'] = cm.reshape(B, 1, -1).transpose(0, 2, 1)[bids, finds]
    return out


def convert_cam_to_3d_trans(cams, weight=2.0):
    """utils.py:303-307."""
    s, tx, ty = cams[:, 0], cams[:, 1], cams[:, 2]
    return (np.stack([tx / s, ty / s, 1.0 / s], 1) * weight).astype(np.float32)


def estimate_translation_np(joints_3d, joints_2d, focal_length=443.4, img_size=512.0):
    """estimate_translation (utils.py:391-434) for one person, OpenCV absent: a joint counts when its 2-D row coordinate is
    > -2 pixels (`joints_conf = joints_2d[:, :, -1] > -2.` reads the LAST coordinate of a 2-column array, :405-406) and its
    depth is not the -2 sentinel; fewer than 4 such joints -> INVALID_TRANS (-1,-1,-1), else estimate_translation_np
    (:347-389, unit weights): the t minimising || f (X + t)_xy - (uv - c) (X + t)_z ||, i.e. the least squares  Q t = c  with
    rows (f, 0, cx - u) and (0, f, cy - v).  joints_3d (K,3), joints_2d (K,2) in pixels.  Pinned by
    tests/golden/translation_lsq.npz (oracle/make_golden_translation.py)."""
    X = np.asarray(joints_3d, np.float64)
    uv = np.asarray(joints_2d, np.float64)
    valid = (np.asarray(joints_2d)[:, -1] > -2.) & (np.asarray(joints_3d)[:, -1] != -2.)
    if valid.sum() < 4:
        return -np.ones(3, np.float32)
    X, uv = X[valid], uv[valid]
    K = X.shape[0]
    f = np.full(2 * K, float(focal_length))
    centre = np.tile(np.array([img_size / 2.0, img_size / 2.0]), K)
    Z = np.repeat(X[:, 2], 2)
    Q = np.stack([f * np.tile([1.0, 0.0], K), f * np.tile([0.0, 1.0], K), centre - uv.reshape(-1)], 1)
    c = (uv.reshape(-1) - centre) * Z - f * X[:, :2].reshape(-1)
    return np.linalg.solve(Q.T @ Q, Q.T @ c).astype(np.float32)


# --------------------------------------------------------------------------------------
# SMPL (smpl.py:62-108, 111-290)
# --------------------------------------------------------------------------------------
SMPL_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
                        np.int64)
NV = 6890


def make_synthetic_smpl(seed=0, n_betas=10):
    """Synthetic SMPL model dict with the exact schema of pack_smpl_info.py:70-111
    (the licensed SMPL_NEUTRAL.pth is not available).  Values per SURVEY.md §8d."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    ru = lambda *s: torch.rand(*s, generator=g)

    def sparse_rows(rows, nnz):
        # row-normalised positive regressor with a few non-zeros per row (like real SMPL)
        m = torch.zeros(rows, NV)
        for r in range(rows):
            idx = torch.randperm(NV, generator=g)[:nnz]
            m[r, idx] = ru(nnz) + 0.05
        return m / m.sum(1, keepdim=True)

    w = torch.zeros(NV, 24)
    for v0 in range(0, NV, 1024):
        n = min(1024, NV - v0)
        idx = torch.randint(0, 24, (n, 4), generator=g)
        w[v0:v0 + n].scatter_(1, idx, ru(n, 4) + 0.05)
    w = w / w.sum(1, keepdim=True)
    faces = torch.randint(0, NV, (13776, 3), generator=g).float()
    d = {
        'kintree_table': torch.from_numpy(SMPL_PARENTS.copy()),
        'J_regressor_extra9': sparse_rows(9, 12),
        'J_regressor_h36m17': sparse_rows(17, 40),
        'shapedirs': 0.01 * rn(NV, 3, n_betas),
        'posedirs': 0.001 * rn(207, NV * 3),
        'extra_joints_index': torch.randperm(NV, generator=g)[:21].long(),
        'f': faces,
        'v_template': 0.3 * rn(NV, 3),
        'J_regressor': sparse_rows(24, 30),
        'weights': w,
    }
    if n_betas == 11:
        d['smpla_shapedirs'] = d['shapedirs']
        d['shapedirs'] = d['shapedirs'][:, :, :10].contiguous()
    return {k: (v.float().contiguous() if v.dtype != torch.int64 else v) for k, v in d.items()}


def batch_rodrigues(rv, dtype=np.float32):
    """smpl.py:191-222 (note: +1e-8 on every component before the norm)."""
    rv = np.asarray(rv, dtype)
    angle = np.sqrt(((rv + dtype(1e-8)) ** 2).sum(1, keepdims=True)).astype(dtype)
    d = rv / angle
    c, s = np.cos(angle)[:, :, None], np.sin(angle)[:, :, None]
    rx, ry, rz = d[:, 0], d[:, 1], d[:, 2]
    z = np.zeros_like(rx)
    K = np.stack([z, -rz, ry, rz, z, -rx, -ry, rx, z], 1).reshape(-1, 3, 3)
    return (np.eye(3, dtype=dtype)[None] + s * K + (1 - c) * (K @ K)).astype(dtype)


def smpl_forward(model, betas, poses, root_align=False, dtype=np.float32):
    """SMPL.forward + lbs + VertexJointSelector (smpl.py:24-35, 62-108, 111-188, 236-290).
    model: dict as loaded from the .pth; betas (N,10|11), poses (N,72).
    Returns verts (N,6890,3), joints (N,71,3), J_transformed (N,24,3)."""
    g = lambda k: np.asarray(model[k].numpy() if torch.is_tensor(model[k]) else model[k])
    betas, poses = np.asarray(betas, dtype), np.asarray(poses, dtype)
    N = betas.shape[0]
    sdirs = g('smpla_shapedirs' if betas.shape[1] == 11 else 'shapedirs').astype(dtype)
    v_t, pdirs = g('v_template').astype(dtype), g('posedirs').astype(dtype)
    Jreg, W = g('J_regressor').astype(dtype), g('weights').astype(dtype)
    parents = g('kintree_table').astype(np.int64)
    v_shaped = v_t[None] + np.einsum('bl,mkl->bmk', betas, sdirs)                  # :153
    J = np.einsum('bik,ji->bjk', v_shaped, Jreg)                                    # :156
    R = batch_rodrigues(poses.reshape(-1, 3), dtype).reshape(N, 24, 3, 3)           # :163
    pose_feat = (R[:, 1:] - np.eye(3, dtype=dtype)).reshape(N, 207)                 # :165
    v_posed = (pose_feat @ pdirs).reshape(N, NV, 3) + v_shaped                      # :167-170
    # batch_rigid_transform (:236-290)
    rel = J.copy()
    rel[:, 1:] -= J[:, parents[1:]]
    T = np.zeros((N, 24, 4, 4), dtype)
    T[:, :, :3, :3], T[:, :, :3, 3], T[:, :, 3, 3] = R, rel, 1
    G = [T[:, 0]]
    for i in range(1, 24):
        G.append(G[parents[i]] @ T[:, i])
    G = np.stack(G, 1)
    J_tr = G[:, :, :3, 3].copy()
    Jh = np.concatenate([J, np.zeros((N, 24, 1), dtype)], 2)[..., None]
    A = G.copy()
    A[:, :, :, 3] -= (G @ Jh)[..., 0]
    Tv = (W @ A.reshape(N, 24, 16)).reshape(N, NV, 4, 4)                            # :179
    vh = np.concatenate([v_posed, np.ones((N, NV, 1), dtype)], 2)[..., None]
    verts = (Tv @ vh)[:, :, :3, 0]                                                  # :185-186
    idx = g('extra_joints_index').astype(np.int64)
    j21 = verts[:, idx]                                                             # :25
    j9 = np.einsum('bik,ji->bjk', verts, g('J_regressor_extra9').astype(dtype))     # :26
    j17 = np.einsum('bik,ji->bjk', verts, g('J_regressor_h36m17').astype(dtype))    # :27
    joints = np.concatenate([J_tr, j21, j9, j17], 1)                                # :29
    if root_align:                                                                  # :102-106
        root = joints[:, [45, 46]].mean(1, keepdims=True)
        joints, verts = joints - root, verts - root
    return verts.astype(dtype), joints.astype(dtype), J_tr.astype(dtype)


def batch_orth_proj(X, cam):
    """utils.py:309-315 (mode '2d')."""
    cam = np.asarray(cam, np.float32).reshape(-1, 1, 3)
    return (X[:, :, :2] * cam[:, :, 0:1] + cam[:, :, 1:]).astype(np.float32)


def project_to_org_image(pj2d, pad_info):
    """convert_proejection_from_input_to_orgimg, post_parser.py:81-88."""
    top, bottom, left, right, h, w = [float(v) for v in pad_info]
    s = max(h, w)
    out = np.array(pj2d, np.float32, copy=True)
    out[:, :, 0] = (out[:, :, 0] + 1) * s / 2 - left
    out[:, :, 1] = (out[:, :, 1] + 1) * s / 2 - top
    return out


def romp_param_shapes():
    return OrderedDict((k, v[0]) for k, v in romp_param_spec().items())
