"""Synthetic weights / inputs for benchmarking (the licensed ROMP.pkl and SMPL_NEUTRAL.pth are
not redistributable and there is no network).  Shapes and key names are the reference's
(state_dict layout: SURVEY.md App. C.2; SMPL schema: pack_smpl_info.py:70-111); values are
seeded random with non-trivial BatchNorm statistics so that BN folding is exercised.
"""
import math
from collections import OrderedDict

import torch

_STAGES = {2: (1, [32, 64]), 3: (4, [32, 64, 128]), 4: (3, [32, 64, 128, 256])}


def romp_hrnet32_spec():
    """Ordered {key: (shape, kind)} for ROMPv1 (HRNet-32 + head)."""
    sp = OrderedDict()

    def conv(n, co, ci, k, bias=False):
        sp[n + '.weight'] = ((co, ci, k, k), 'w')
        if bias:
            sp[n + '.bias'] = ((co,), 'b')

    def bn(n, c):
        for suf, kind in (('.weight', 'g'), ('.bias', 'beta'), ('.running_mean', 'm'), ('.running_var', 'v')):
            sp[n + suf] = ((c,), kind)

    conv('backbone.conv1', 64, 3, 3); bn('backbone.bn1', 64)
    conv('backbone.conv2', 64, 64, 3); bn('backbone.bn2', 64)
    for i in range(4):
        p = f'backbone.layer1.{i}.'
        conv(p + 'conv1', 64, 64 if i == 0 else 256, 1); bn(p + 'bn1', 64)
        conv(p + 'conv2', 64, 64, 3); bn(p + 'bn2', 64)
        conv(p + 'conv3', 256, 64, 1); bn(p + 'bn3', 256)
        if i == 0:
            conv(p + 'downsample.0', 256, 64, 1); bn(p + 'downsample.1', 256)
    conv('backbone.transition1.0.0', 32, 256, 3); bn('backbone.transition1.0.1', 32)
    conv('backbone.transition1.1.0.0', 64, 256, 3); bn('backbone.transition1.1.0.1', 64)
    conv('backbone.transition2.2.0.0', 128, 64, 3); bn('backbone.transition2.2.0.1', 128)
    conv('backbone.transition3.3.0.0', 256, 128, 3); bn('backbone.transition3.3.0.1', 256)
    for s, (n_mod, ch) in _STAGES.items():
        for m in range(n_mod):
            p = f'backbone.stage{s}.{m}.'
            for br, c in enumerate(ch):
                for k in range(4):
                    for cv, b in (('conv1', 'bn1'), ('conv2', 'bn2')):
                        conv(f'{p}branches.{br}.{k}.{cv}', c, c, 3); bn(f'{p}branches.{br}.{k}.{b}', c)
            for i in range(1 if (s == 4 and m == n_mod - 1) else len(ch)):
                for j in range(len(ch)):
                    q = f'{p}fuse_layers.{i}.{j}.'
                    if j > i:
                        conv(q + '0', ch[i], ch[j], 1); bn(q + '1', ch[i])
                    for k in range(i - j):
                        co = ch[i] if k == i - j - 1 else ch[j]
                        conv(f'{q}{k}.0', co, ch[j], 3); bn(f'{q}{k}.1', co)
    for h, co in ((1, 142), (2, 1), (3, 3)):
        p = f'final_layers.{h}.'
        conv(p + '0.0', 64, 34, 3, True); bn(p + '0.1', 64)
        for blk in range(2):
            for cv, b in (('conv1', 'bn1'), ('conv2', 'bn2')):
                conv(f'{p}1.{blk}.0.{cv}', 64, 64, 3); bn(f'{p}1.{blk}.0.{b}', 64)
        conv(p + '2', co, 64, 1, True)
    return sp


def make_romp_state_dict(seed=0, center_bias=2.0):
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, (shp, kind) in romp_hrnet32_spec().items():
        if kind == 'w':
            v = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(shp[1] * shp[2] * shp[3])
        elif kind == 'b':
            v = (torch.rand(shp, generator=g) * 2 - 1) * 0.05
        elif kind == 'g':
            v = torch.rand(shp, generator=g) * 0.4 + 0.8
        elif kind == 'v':
            v = torch.rand(shp, generator=g) + 0.5
        else:
            v = torch.randn(shp, generator=g) * 0.1
        sd[k] = v.float()
    sd['final_layers.2.2.bias'] += center_bias      # positive center maps -> a few persons / image
    return sd


def make_images(batch, seed=1, device='cpu'):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (batch, 512, 512, 3), generator=g).float().to(device)


def make_smpl_model(seed=0, n_betas=10):
    """Random SMPL-shaped model dict (dense regressors/weights, row-normalised)."""
    g = torch.Generator().manual_seed(seed)
    NV = 6890
    rn = lambda *s: torch.randn(*s, generator=g)

    def rows(n, nnz):
        m = torch.zeros(n, NV)
        idx = torch.randint(0, NV, (n, nnz), generator=g)
        m.scatter_(1, idx, torch.rand(n, nnz, generator=g) + 0.05)
        return m / m.sum(1, keepdim=True)

    w = torch.zeros(NV, 24)
    w.scatter_(1, torch.randint(0, 24, (NV, 4), generator=g), torch.rand(NV, 4, generator=g) + 0.05)
    d = {
        'kintree_table': torch.tensor([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]),
        'J_regressor_extra9': rows(9, 12), 'J_regressor_h36m17': rows(17, 40),
        'shapedirs': 0.01 * rn(NV, 3, n_betas), 'posedirs': 0.001 * rn(207, NV * 3),
        'extra_joints_index': torch.randperm(NV, generator=g)[:21], 'f': torch.randint(0, NV, (13776, 3), generator=g).float(),
        'v_template': 0.3 * rn(NV, 3), 'J_regressor': rows(24, 30), 'weights': w / w.sum(1, keepdim=True),
    }
    if n_betas == 11:
        d['smpla_shapedirs'] = d['shapedirs']
        d['shapedirs'] = d['shapedirs'][:, :, :10].contiguous()
    return d


# ---------------------------------------------------------------------------------------------- BEV / ResNet-50
def _fill(spec, g):
    """Seeded values for an ordered {key: (shape, kind)} spec; kinds: w (fan-in scaled), b, g/beta/m/v (BatchNorm), emb."""
    sd = OrderedDict()
    for k, (shp, kind) in spec.items():
        if kind == 'w':
            fan = 1
            for d in shp[1:]:
                fan *= d
            v = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(fan)
        elif kind == 'wT':                              # ConvTranspose2d (Cin, Cout, 4, 4): an output sees Cin * 4 of the 16 taps
            v = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(shp[0] * 4)
        elif kind == 'b':
            v = (torch.rand(shp, generator=g) * 2 - 1) * 0.05
        elif kind == 'g':
            v = torch.rand(shp, generator=g) * 0.4 + 0.8
        elif kind == 'v':
            v = torch.rand(shp, generator=g) + 0.5
        else:                                           # beta, m, emb
            v = torch.randn(shp, generator=g) * 0.1
        sd[k] = v.float().contiguous()
    return sd


def _bn(sp, n, c):
    for suf, kind in (('.weight', 'g'), ('.bias', 'beta'), ('.running_mean', 'm'), ('.running_var', 'v')):
        sp[n + suf] = ((c,), kind)


def bev_head_spec():
    """Parameters of the BEVv1 head (simple_romp/bev/model.py:115-186)."""
    sp = OrderedDict()
    sp['position_embeddings.weight'] = ((128, 128), 'emb')
    for i, (co, ci) in zip((0, 3, 6), ((512, 128), (512, 512), (143, 512))):
        sp[f'transformer.{i}.weight'] = ((co, ci), 'w'); sp[f'transformer.{i}.bias'] = ((co,), 'b')
    for head in ('det_head', 'param_head'):
        p = f'{head}.0.0.'
        sp[p + 'conv1.weight'] = ((128, 32, 3, 3), 'w'); _bn(sp, p + 'bn1', 128)
        sp[p + 'conv2.weight'] = ((128, 128, 3, 3), 'w'); _bn(sp, p + 'bn2', 128)
        sp[p + 'downsample.weight'] = ((128, 32, 1, 1), 'w'); sp[p + 'downsample.bias'] = ((128,), 'b')
        if head == 'det_head':
            sp['det_head.1.weight'] = ((4, 128, 1, 1), 'w'); sp['det_head.1.bias'] = ((4,), 'b')
    for i, (k, ci) in zip((0, 3, 6), ((1, 32), (3, 16), (1, 16))):
        sp[f'bv_pre_layers.{i}.weight'] = ((16, ci, k, k), 'w'); sp[f'bv_pre_layers.{i}.bias'] = ((16,), 'b')
        _bn(sp, f'bv_pre_layers.{i + 1}', 16)
    for i, (ci, co) in enumerate(((2560, 512), (512, 512), (512, 128))):
        p = f'bv_out_layers.{i}.'
        sp[p + 'conv1.weight'] = ((co, ci, 3), 'w'); _bn(sp, p + 'bn1', co)
        sp[p + 'conv2.weight'] = ((co, co, 3), 'w'); _bn(sp, p + 'bn2', co)
    for name, c in (('center_map_refiner', 1), ('cam_map_refiner', 3)):
        p = f'{name}.0.'
        sp[p + 'conv1.weight'] = ((c, c, 3, 3, 3), 'w'); _bn(sp, p + 'bn1', c)
        sp[p + 'conv2.weight'] = ((c, c, 3, 3, 3), 'w'); _bn(sp, p + 'bn2', c)
    return sp


def make_bev_state_dict(seed=0, center_bias=1.5):
    """Seeded BEVv1 weights: HRNet-32 backbone of make_romp_state_dict + head; the front-view center output is biased
    positive so that a positive threshold keeps a handful of 3-D centres per image."""
    sd = OrderedDict((k, v) for k, v in make_romp_state_dict(seed, center_bias=0.0).items() if k.startswith('backbone.'))
    sd.update(_fill(bev_head_spec(), torch.Generator().manual_seed(seed + 1000)))
    sd['det_head.1.bias'][0] += center_bias
    return sd


def resnet50_romp_spec():
    """Parameters of ROMP(ResNet_50) of the training tree (romp/lib/models/resnet_50.py:40-120, romp_model.py:53-103)."""
    sp = OrderedDict()

    def conv(n, co, ci, k, bias=False):
        sp[n + '.weight'] = ((co, ci, k, k), 'w')
        if bias:
            sp[n + '.bias'] = ((co,), 'b')

    conv('backbone.conv1', 64, 3, 7); _bn(sp, 'backbone.bn1', 64)
    inpl = 64
    for li, (planes, blocks) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3)), 1):
        for i in range(blocks):
            p = f'backbone.layer{li}.{i}.'
            conv(p + 'conv1', planes, inpl, 1); _bn(sp, p + 'bn1', planes)
            conv(p + 'conv2', planes, planes, 3); _bn(sp, p + 'bn2', planes)
            conv(p + 'conv3', planes * 4, planes, 1); _bn(sp, p + 'bn3', planes * 4)
            if i == 0:
                conv(p + 'downsample.0', planes * 4, inpl, 1); _bn(sp, p + 'downsample.1', planes * 4)
            inpl = planes * 4
    for i, co in enumerate((256, 128, 64)):
        sp[f'backbone.deconv_layers.{3 * i}.weight'] = ((inpl, co, 4, 4), 'wT'); _bn(sp, f'backbone.deconv_layers.{3 * i + 1}', co)
        inpl = co
    for h, co in ((1, 142), (2, 1), (3, 3)):
        p = f'final_layers.{h}.'
        conv(p + '0.0', 64, 66, 3, True); _bn(sp, p + '0.1', 64)
        for blk in range(2):
            for cv, b in (('conv1', 'bn1'), ('conv2', 'bn2')):
                conv(f'{p}1.{blk}.0.{cv}', 64, 64, 3); _bn(sp, f'{p}1.{blk}.0.{b}', 64)
        conv(p + '2', co, 64, 1, True)
    return sp


def make_resnet_state_dict(seed=0, center_bias=2.0):
    sd = _fill(resnet50_romp_spec(), torch.Generator().manual_seed(seed))
    sd['final_layers.2.2.bias'] += center_bias
    return sd
