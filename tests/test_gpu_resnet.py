"""GPU parity for the ResNet-50 variant of ROMP (BASELINE configs[0]): new layer kinds through the C ABI
(7x7 stem + max-pool, strided 1x1, transposed conv as four 2x2 parity convs) against torch, and the whole
network against the CPU oracle (oracle/resnet_oracle.py) and the fixture produced by the reference's ResNet_50
(tests/golden/resnet50_b1.npz).  Tolerance 1e-4 max-abs on the maps (float32 network, ~60 sequential layers)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import resnet_oracle as RO
from oracle import romp_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from romp_amd import lib
    lib.load()
    return torch.device('cuda:0')


def _variants(lib, op, B):
    buf = C.create_string_buffer(128)
    return [(0, v) for v in range(lib.romp_conv_num_variants()) if lib.romp_conv_describe(C.byref(op), B, v, buf, 128) == 0]


@pytest.mark.parametrize('cin,cout,H', [(2048, 256, 16), (256, 128, 32), (128, 64, 64)])
def test_transposed_conv_as_parity_convs(dev, cin, cout, H):
    """ConvTranspose2d(k4, s2, p1) + BN + ReLU (resnet_50.py:93-120) == four 2x2 convs writing interleaved."""
    from romp_amd import lib as L
    from romp_amd.plan import Program, Act
    B = 2
    g = torch.Generator().manual_seed(cin)
    x = torch.randn(B, cin, H, H, generator=g)
    w = torch.randn(cin, cout, 4, 4, generator=g) / (cin * 4) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ref = torch.relu(F.conv_transpose2d(x, w, None, stride=2, padding=1) * scale[None, :, None, None] + shift[None, :, None, None])
    ref = ref.permute(0, 2, 3, 1).contiguous()
    lib = L.load()
    KY = ((3, 1), (2, 0))
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    H2 = 2 * H
    worst = {}
    for bx3 in (False, True):
        P = Program(dev)
        P.bf16x3 = bx3
        P.buf_floats += [cin * H * H, cout * H2 * H2]
        for a in range(2):
            for b in range(2):
                w2 = torch.stack([torch.stack([w[:, :, KY[a][dy], KY[b][dx]] for dx in range(2)], -1) for dy in range(2)], -2).permute(1, 0, 2, 3).contiguous()
                P.conv(f'p{a}{b}', Act(0, cin, H, H, cin), [w2], [scale], [shift], 2, 1, True, out_buf_special=1, out_cstride=2 * cout,
                       out_coff=a * H2 * cout + b * cout, pad=(1 - a, 1 - b), out_rstride=2 * H2 * cout, out_bstride=H2 * H2 * cout)
        runs = [(1, -1), (0, -1)] + _variants(lib, P.ops[0], B)
        for mode, variant in runs:
            out = torch.full((B, H2, H2, cout), float('nan'), device=dev)
            for op in P.ops:
                L.check(lib.romp_conv_forward(C.byref(op), L.ptr(xd), None, L.ptr(out), B, mode, variant, L.stream_ptr(dev)))
            torch.cuda.synchronize()
            err = (out.cpu() - ref).abs().max().item()
            worst[(bx3, mode, variant)] = err
            assert err < 5e-5, (bx3, mode, variant, err)
    print('deconv %d->%d @%d: %d kernel variants, worst max-abs %.2e' % (cin, cout, H, len(worst), max(worst.values())))


@pytest.mark.parametrize('cin,cout,H', [(256, 512, 128), (1024, 2048, 32)])
def test_conv1x1_stride2(dev, cin, cout, H):
    from romp_amd import lib as L
    from romp_amd.plan import Program, Act
    B = 2
    g = torch.Generator().manual_seed(cout)
    x = torch.randn(B, cin, H, H, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ref = (F.conv2d(x, w, None, stride=2) * scale[None, :, None, None] + shift[None, :, None, None]).permute(0, 2, 3, 1)
    P = Program(dev)
    P.buf_floats.append(cin * H * H)
    P.conv('ds', Act(0, cin, H, H, cin), [w], [scale], [shift], 1, 2, False)
    lib = L.load()
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    for mode, variant in [(1, -1), (0, -1)] + _variants(lib, P.ops[0], B):
        out = torch.full((B, H // 2, H // 2, cout), float('nan'), device=dev)
        L.check(lib.romp_conv_forward(C.byref(P.ops[0]), L.ptr(xd), None, L.ptr(out), B, mode, variant, L.stream_ptr(dev)))
        torch.cuda.synchronize()
        assert (out.cpu() - ref).abs().max().item() < 5e-5, (mode, variant)


@pytest.fixture(scope='module')
def rnet(dev):
    from romp_amd.net import RompNet
    from romp_amd.resnet_plan import build_romp_resnet50
    return RompNet(RO.make_resnet_state_dict(0), dev, max_batch=2, builder=build_romp_resnet50)


def test_resnet_stem_and_backbone_vs_reference_fixture(dev, golden_dir, rnet):
    g = np.load(os.path.join(golden_dir, 'resnet50_b1.npz'))
    img = O.make_images(1, seed=7)
    cm, pm = rnet(img.to(dev))
    feat = rnet.read_buffer(rnet.program.head_in_buf, 1).reshape(128, 128, -1)[:, :, :64].permute(2, 0, 1).cpu().numpy()
    e1 = np.abs(feat.reshape(64, -1)[:, g['sample_pos']] - g['feat_samples']).max()
    e2 = np.abs(feat.astype(np.float64).sum((1, 2)) - g['feat_chan_sum']).max()
    ec = np.abs(cm.cpu().numpy() - g['center_maps']).max()
    es = np.abs(pm[0].cpu().numpy().astype(np.float64).sum((1, 2)) - g['params_chan_sum']).max()
    print(f'ResNet-50 backbone vs reference fixture: samples {e1:.3e} channel sums {e2:.3e}; center maps {ec:.3e} params channel sums {es:.3e}')
    assert e1 < 1e-4 and e2 < 5e-2 and ec < 1e-4 and es < 5e-2


@pytest.mark.parametrize('bf16x3', [False, True])
def test_resnet_net_vs_oracle(dev, bf16x3):
    from romp_amd.net import RompNet
    from romp_amd.resnet_plan import build_romp_resnet50
    sd = RO.make_resnet_state_dict(0)
    net = RompNet(sd, dev, max_batch=2, builder=build_romp_resnet50, bf16x3=bf16x3)
    img = O.make_images(2, seed=3)
    cm_o, pm_o = RO.resnet_romp_forward(sd, img)
    outs = {}
    for mode in ((0,) if bf16x3 else (1, 0)):
        net.set_mode(mode)
        cm, pm = net(img.to(dev))
        ec, ep = (cm.cpu() - cm_o).abs().max().item(), (pm.cpu() - pm_o).abs().max().item()
        print(f'ResNet-50 ROMP bf16x3={bf16x3} mode {mode}: center {ec:.3e} params {ep:.3e}')
        assert ec < 1e-4 and ep < 1e-4
    if bf16x3:
        assert sum('bx' in n for n in net.variant_names(2)) > 0
