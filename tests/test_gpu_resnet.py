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


@pytest.mark.parametrize('B,S,fmt', [(1, 64, 'h2'), (3, 96, 'h2'), (2, 128, 'f32'), (2, 512, 'h2'), (5, 160, 'f32')])
def test_stem7p_vs_torch(dev, B, S, fmt, monkeypatch):
    """csrc/stem7p.hip (ROMP_OP_STEM7P): ImageNet normalisation + conv7x7 s2 p3 + BN + ReLU + MaxPool2d(3, 2, 1) as ONE MFMA kernel
    (romp/lib/models/resnet_50.py:32-45,56), lowered by plan.fuse_stem7p from the [STEM7, MAXPOOL] pair, against torch on the CPU:
    image borders on every side (the conv's zero padding after normalisation, the pool's -inf padding), partial last workgroup
    rounds, both output formats."""
    from romp_amd import lib as L
    from romp_amd.plan import Program, set_conv_math, decode_h2
    from romp_amd.resnet_plan import _stem7, _maxpool
    monkeypatch.delenv('ROMP_STEM', raising=False)
    g = torch.Generator().manual_seed(10 * S + B)
    img = torch.randint(0, 256, (B, S, S, 3), generator=g).float()
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.08
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    x = ((img / 255.0 - mean) / std).permute(0, 3, 1, 2)
    m = torch.relu(F.conv2d(x, w, None, stride=2, padding=3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    ref = F.max_pool2d(m, 3, 2, 1).permute(0, 2, 3, 1)
    P = Program(dev)
    set_conv_math(P, 'f16x2')
    y = _maxpool(P, 'pool', _stem7(P, 'stem', w, sc, sh, S, S))
    if fmt == 'f32':
        P.exported_bufs.add(y.buf)                           # a tensor the host reads stays float32
    ops = P.op_array()
    assert P.fused_stem7p == 1 and [o.kind for o in P.ops] == [L.OP_NOP, L.OP_STEM7P]
    assert P.ops[1].out_fmt == (L.FMT_H2 if fmt == 'h2' else L.FMT_F32)
    lib = L.load()
    h = C.c_void_p()
    sizes = (C.c_int64 * len(P.buf_floats))(*P.buf_floats)
    L.check(lib.romp_net_create(C.byref(h), ops, len(P.ops), sizes, len(P.buf_floats), B))
    try:
        xd = img.to(dev).contiguous()
        dummy = torch.empty(16, device=dev)
        n = P.buf_floats[y.buf] * B
        out = torch.empty(n, device=dev)
        for rep in range(2):
            L.check(lib.romp_net_forward(h, L.ptr(xd), B, L.ptr(dummy), L.ptr(dummy), L.stream_ptr(dev)))
            L.check(lib.romp_net_read_buffer(h, y.buf, B, L.ptr(out), n, L.stream_ptr(dev)))
            torch.cuda.synchronize()
            got = out.cpu().reshape(B, S // 4, S // 4, 64)
            if fmt == 'h2':
                got = decode_h2(got)
            err = (got - ref).abs().max().item() / ref.abs().max().item()
            print(f'stem7p B={B} {S}x{S} {fmt} run {rep}: relative err {err:.3e}')
            assert err < 2e-5, err
    finally:
        lib.romp_net_destroy(h)


def test_stem7p_falls_back_to_the_float32_pair(dev, monkeypatch):
    """ROMP_STEM=valu (and weights beyond the fp16 pieces): the plan keeps ROMP_OP_STEM7 + ROMP_OP_MAXPOOL."""
    from romp_amd.plan import Program, set_conv_math
    from romp_amd.resnet_plan import _stem7, _maxpool, OP_STEM7, OP_MAXPOOL
    g = torch.Generator().manual_seed(1)
    w = torch.randn(64, 3, 7, 7, generator=g)
    for env, scale in (('valu', 0.05), ('', 300.0)):
        monkeypatch.setenv('ROMP_STEM', env)
        P = Program(dev)
        set_conv_math(P, 'f16x2')
        _maxpool(P, 'pool', _stem7(P, 'stem', w * scale, torch.ones(64), torch.zeros(64), 64, 64))
        P.op_array()
        assert P.fused_stem7p == 0 and [o.kind for o in P.ops] == [OP_STEM7, OP_MAXPOOL]


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
    for bx3 in ((False, True) if L.has_bf16x3() else (False,)):      # (the bf16x3 family is optional since round 6)
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


@pytest.mark.parametrize('cin,cout,H', [(2048, 256, 16), (256, 128, 32), (128, 64, 64)])
def test_transposed_conv_as_parity_convs_h2(dev, cin, cout, H):
    """The same four parity convs on pre-split H2 tensors (what the f16x2 program runs): every kernel variant that can take the layer
    -- the generic f16x2 kernels and csrc/conv_h2g.hip's K = 4 taps x Cin form -- writes its parity of the interleaved H2 output
    through the sparse output strides; the decoded tensor against torch's ConvTranspose2d."""
    from romp_amd import lib as L
    from romp_amd.plan import Program, Act, set_conv_math, encode_h2, decode_h2, ACT_SHIFT
    B = 2
    g = torch.Generator().manual_seed(cin + 1)
    x = torch.randn(B, cin, H, H, generator=g)
    w = torch.randn(cin, cout, 4, 4, generator=g) / (cin * 4) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ref = torch.relu(F.conv_transpose2d(x, w, None, stride=2, padding=1) * scale[None, :, None, None] + shift[None, :, None, None])
    ref = ref.permute(0, 2, 3, 1).contiguous()
    lib = L.load()
    KY = ((3, 1), (2, 0))
    xd = encode_h2(x.permute(0, 2, 3, 1).contiguous()).to(dev)
    H2 = 2 * H
    P = Program(dev)
    set_conv_math(P, 'f16x2')
    P.buf_floats += [cin * H * H, cout * H2 * H2]
    for a in range(2):
        for b in range(2):
            w2 = torch.stack([torch.stack([w[:, :, KY[a][dy], KY[b][dx]] for dx in range(2)], -1) for dy in range(2)], -2).permute(1, 0, 2, 3).contiguous()
            P.conv(f'p{a}{b}', Act(0, cin, H, H, cin), [w2], [scale], [shift], 2, 1, True, out_buf_special=1, out_cstride=2 * cout,
                   out_coff=a * H2 * cout + b * cout, pad=(1 - a, 1 - b), out_rstride=2 * H2 * cout, out_bstride=H2 * H2 * cout)
    for op in P.ops:
        assert op.weight_h2
        op.act_shift, op.in_fmt, op.out_fmt = ACT_SHIFT, L.FMT_H2, L.FMT_H2
    buf = C.create_string_buffer(128)
    names = {}
    for mode, variant in [(0, -1)] + _variants(lib, P.ops[0], B):
        out = torch.full((B, H2, H2, cout), float('nan'), device=dev)
        for op in P.ops:
            L.check(lib.romp_conv_forward(C.byref(op), L.ptr(xd), None, L.ptr(out), B, mode, variant, L.stream_ptr(dev)))
        torch.cuda.synchronize()
        L.check(lib.romp_conv_describe(C.byref(P.ops[0]), B, variant, buf, 128))
        err = (decode_h2(out.cpu()) - ref).abs().max().item()
        names[buf.value.decode()] = err
        assert err < 5e-5, (buf.value, variant, err)
    assert any(n.startswith('conv_h2g_k2s1') for n in names), names
    print('deconv (H2) %d->%d @%d: %s' % (cin, cout, H, ', '.join('%s %.1e' % kv for kv in sorted(names.items()))))


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


@pytest.mark.parametrize('bf16x3', [False, 'f16x2', True])
def test_resnet_net_vs_oracle(dev, bf16x3):
    from romp_amd import lib as _L
    if bf16x3 is True and not _L.has_bf16x3():
        pytest.skip('library built without the bf16x3 family (ROMP_WITH_BX3=1 python -m romp_amd.build)')
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
    if bf16x3 is True:
        assert sum('bx' in n for n in net.variant_names(2)) > 0
    if bf16x3 == 'f16x2':                                  # (round 6: the f16x2 ResNet-50 -- the bench line's arithmetic -- at the same gate)
        assert sum('conv_h2g' in n for n in net.variant_names(2)) >= 0


@pytest.mark.parametrize('size', [(1920, 1080), (1280, 720)], ids=['1080p', '720p'])
def test_configs0_demo_shaped_jpeg_through_the_api(dev, size, tmp_path):
    """BASELINE configs[0] as written: `ROMP(--backbone resnet50)(image)` on a demo-image-shaped input.  /root/reference/demo/images
    does not exist on the GPU box, so the frame is a synthetic JPEG of the demo images' sizes (1920x1080, 1280x720) written and
    decoded with PIL, fed as the BGR uint8 array cv2.imread would return.  The drop-in's result dict (reference keys, dtypes,
    shapes) must equal the oracle pipeline -- ResNet-50 network, parse, SMPL -- on the oracle's own pre-processing of the same
    frame (pad to square + INTER_CUBIC resize, oracle/cv_resize_oracle.py)."""
    from PIL import Image
    import romp_amd
    from oracle import cv_resize_oracle as CV
    W, H = size
    rs = np.random.RandomState(W)
    yy, xx = np.mgrid[0:H, 0:W]
    rgb = np.stack([(xx * 255 // W), (yy * 255 // H), ((xx + yy) % 256)], -1).astype(np.uint8)
    rgb[H // 4: H // 4 * 3, W // 3: W // 3 * 2] = rs.randint(0, 256, (H // 4 * 3 - H // 4, W // 3 * 2 - W // 3, 3), dtype=np.uint8)
    path = str(tmp_path / 'demo.jpg')
    Image.fromarray(rgb).save(path, quality=92)
    bgr = np.ascontiguousarray(np.asarray(Image.open(path).convert('RGB'))[:, :, ::-1])      # what cv2.imread(path) yields
    assert bgr.shape == (H, W, 3) and bgr.dtype == np.uint8
    sd = RO.make_resnet_state_dict(0, center_bias=2.0)
    smpl = O.make_synthetic_smpl(0)
    s = romp_amd.romp_settings(['--backbone', 'resnet50'])
    s.GPU, s.max_batch = 0, 1
    model = romp_amd.ROMP(s, state_dict=sd, smpl_model=smpl)
    # a threshold that keeps a handful of persons with these random weights (outside what is compared)
    x = torch.from_numpy(CV.img_preprocess(bgr)[0]).float()
    cm_o, pm_o = RO.resnet_romp_forward(sd, x)
    peaks = np.sort(cm_o.numpy().reshape(-1))[::-1]
    model.centermap_parser.conf_thresh = thresh = float(0.5 * (peaks[40] + peaks[41]))
    out = model(bgr)
    ref = O.parsing_outputs(cm_o.numpy(), pm_o.numpy(), thresh)
    assert out is not None and ref is not None
    for k in ('cam', 'global_orient', 'body_pose', 'smpl_betas', 'smpl_thetas', 'center_preds', 'center_confs', 'cam_trans', 'verts', 'joints', 'pj2d_org'):
        assert k in out and isinstance(out[k], np.ndarray), k
    N = len(ref['batch_ids'])
    assert out['verts'].shape == (N, 6890, 3) and out['joints'].shape == (N, 71, 3) and out['pj2d_org'].shape == (N, 71, 2)
    assert np.array_equal(out['center_preds'], ref['center_preds'])
    assert np.abs(out['smpl_thetas'] - ref['smpl_thetas']).max() < 1e-3
    vo, jo, _ = O.smpl_forward(smpl, out['smpl_betas'], out['smpl_thetas'])
    ev = float(np.abs(out['verts'] - vo).max())
    print('%dx%d JPEG through ROMP(resnet50): %d persons, verts vs oracle on identical theta/beta %.2e' % (W, H, N, ev))
    assert ev < 1e-4
