"""CPU restatement of ROMP with the ResNet-50 backbone (BASELINE configs[0]) -- TEST
INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline; the product path is romp_amd/resnet_plan.py -> libromp_hip.so).

The ResNet-50 variant only exists in the reference's training tree:
  * backbone   romp/lib/models/resnet_50.py  ResNet_50: image_preprocess :32-38 (x/255, ImageNet mean/std),
               make_resnet :40-52 (7x7 s2 stem, MaxPool 3x3 s2, Bottleneck layers [3,4,6,3] -- stride on the 3x3,
               romp/lib/models/basic_modules.py:90-128), three ConvTranspose2d(k4,s2,p1)+BN+ReLU :93-120 -> 64 ch @128^2
  * head       romp/lib/models/romp_model.py  head_forward :35-50, _make_head_layers :78-103 (centermap_size 64,
               head_block_num 2: the same three towers as simple_romp's ROMPv1, on 64+2 CoordConv channels)
The model applies 1.1**scale itself (romp_model.py:47); like simple_romp (main.py:113) the HIP path applies it to the
sampled rows in the parser, so the maps below are the RAW head outputs.

Pinned by tests/golden/resnet50_b1.npz: oracle/make_golden_resnet.py imports the reference's ResNet_50 by file path
(stub modules for torchvision / config / utils, whose only used pieces are restated there) and runs it on this
module's seeded weights; the head is the structure already pinned for ROMPv1 (romp_oracle.head towers).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import romp_oracle as O

LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))     # planes, blocks, stride of the first block
DECONV = (256, 128, 64)
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
HEAD_OUT = {1: 142, 2: 1, 3: 3}


def resnet_param_spec():
    """Ordered {key: (shape, kind)} of the float tensors of ROMP(ResNet_50) (state_dict order of the reference modules)."""
    sh = OrderedDict()

    def conv(name, cout, cin, k, bias=False):
        sh[name + '.weight'] = ((cout, cin, k, k), 'conv_w')
        if bias:
            sh[name + '.bias'] = ((cout,), 'conv_b')

    def bn(name, c):
        for k, kind in zip(O._bn_keys(name), ('bn_w', 'bn_b', 'bn_m', 'bn_v')):
            sh[k] = ((c,), kind)

    b = 'backbone.'
    conv(b + 'conv1', 64, 3, 7); bn(b + 'bn1', 64)
    inpl = 64
    for li, (planes, blocks, stride) in enumerate(LAYERS, 1):
        for i in range(blocks):
            p = f'{b}layer{li}.{i}.'
            conv(p + 'conv1', planes, inpl, 1); bn(p + 'bn1', planes)
            conv(p + 'conv2', planes, planes, 3); bn(p + 'bn2', planes)
            conv(p + 'conv3', planes * 4, planes, 1); bn(p + 'bn3', planes * 4)
            if i == 0:
                conv(p + 'downsample.0', planes * 4, inpl, 1); bn(p + 'downsample.1', planes * 4)
            inpl = planes * 4
    for i, co in enumerate(DECONV):                                  # ConvTranspose2d weight: (Cin, Cout, 4, 4)
        sh[f'{b}deconv_layers.{3 * i}.weight'] = ((inpl, co, 4, 4), 'deconv_w')
        bn(f'{b}deconv_layers.{3 * i + 1}', co)
        inpl = co
    for h, co in HEAD_OUT.items():
        p = f'final_layers.{h}.'
        conv(p + '0.0', 64, 66, 3, bias=True); bn(p + '0.1', 64)
        for blk in range(2):
            q = f'{p}1.{blk}.0.'
            conv(q + 'conv1', 64, 64, 3); bn(q + 'bn1', 64)
            conv(q + 'conv2', 64, 64, 3); bn(q + 'bn2', 64)
        conv(p + '2', co, 64, 1, bias=True)
    return sh


def make_resnet_state_dict(seed=0, center_bias=0.0):
    """Seeded synthetic weights with non-trivial BN statistics (same recipe as romp_oracle.make_romp_state_dict)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, (shp, kind) in resnet_param_spec().items():
        if kind == 'bn_m':
            v = torch.randn(shp, generator=g) * 0.1
        elif kind == 'bn_v':
            v = torch.rand(shp, generator=g) + 0.5
        elif kind == 'bn_w':
            v = torch.rand(shp, generator=g) * 0.4 + 0.8
        elif kind == 'bn_b':
            v = torch.randn(shp, generator=g) * 0.1
        elif kind == 'conv_w':
            v = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(shp[1] * shp[2] * shp[3])
        elif kind == 'deconv_w':                                    # each output sees Cin * 4 taps of the 16
            v = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(shp[0] * 4)
        else:
            v = (torch.rand(shp, generator=g) * 2 - 1) * 0.05
        sd[k] = v.float().contiguous()
    if center_bias:
        sd['final_layers.2.2.bias'] = sd['final_layers.2.2.bias'] + float(center_bias)
    return sd


def _bottleneck(x, sd, p, stride):
    """Bottleneck.forward, basic_modules.py:108-128 (stride on conv2 and on the 1x1 downsample)."""
    y = torch.relu(O._bn(O._conv(x, sd, p + 'conv1'), sd, p + 'bn1'))
    y = torch.relu(O._bn(O._conv(y, sd, p + 'conv2', stride), sd, p + 'bn2'))
    y = O._bn(O._conv(y, sd, p + 'conv3'), sd, p + 'bn3')
    r = x
    if (p + 'downsample.0.weight') in sd:
        r = O._bn(O._conv(x, sd, p + 'downsample.0', stride), sd, p + 'downsample.1')
    return torch.relu(y + r)


@torch.no_grad()
def backbone_forward(sd, image_nhwc):
    """ResNet_50.forward, resnet_50.py:54-62.  image (B,512,512,3) 0..255 -> (B,64,128,128)."""
    b = 'backbone.'
    x = image_nhwc.permute(0, 3, 1, 2) / 255.
    x = (x - torch.tensor(MEAN).view(1, 3, 1, 1)) / torch.tensor(STD).view(1, 3, 1, 1)
    x = torch.relu(O._bn(O._conv(x.contiguous(), sd, b + 'conv1', 2), sd, b + 'bn1'))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (planes, blocks, stride) in enumerate(LAYERS, 1):
        for i in range(blocks):
            x = _bottleneck(x, sd, f'{b}layer{li}.{i}.', stride if i == 0 else 1)
    for i in range(3):
        x = F.conv_transpose2d(x, sd[f'{b}deconv_layers.{3 * i}.weight'], None, stride=2, padding=1)
        x = torch.relu(O._bn(x, sd, f'{b}deconv_layers.{3 * i + 1}'))
    return x


@torch.no_grad()
def resnet_romp_forward(sd, image_nhwc):
    """-> center_maps (B,1,64,64), params_maps (B,145,64,64) [cam 3 | params 142], raw scale channel."""
    return O.head_forward(sd, backbone_forward(sd, image_nhwc))
