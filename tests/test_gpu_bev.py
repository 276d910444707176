"""GPU parity tests for the BEV path (BASELINE config 4): HIP kernels through the C ABI vs the CPU
oracle (oracle/bev_oracle.py) and the reference-generated fixtures (bev_b1.npz, smpla_parser_n5.npz).

Tolerances: 3-D center / camera maps 1e-4 max-abs (float32 conv stack + Conv1d K=7680 + two 3-D
convs); detections exact as a set on fixture inputs; params_pred 2e-4; SMPL-A meshes 1e-4.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import bev_oracle as BO
from oracle import romp_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from romp_amd import lib
    lib.load()
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def bev_model(dev):
    from romp_amd.bev import BEVv1
    return BEVv1(BO.make_bev_state_dict(0), dev, center_thresh=0.1, max_batch=2)


@pytest.mark.parametrize('B', [1, 5, 32])
@pytest.mark.parametrize('cin,cout', [(2560, 512), (512, 128)])
def test_conv1d_layer(dev, B, cin, cout):
    """Conv1d(k=3)+BN+ReLU of the bird's-eye-view head on the MFMA conv kernel (ksize code 13)."""
    from romp_amd import lib as L
    from romp_amd.plan import Program, Act
    g = torch.Generator().manual_seed(cin + cout + B)
    x = torch.randn(B, cin, 128, generator=g)                     # (B, C, L) like the reference
    w = torch.randn(cout, cin, 3, generator=g) / (cin * 3) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ref = torch.relu(F.conv1d(x, w, None, padding=1) * scale[None, :, None] + shift[None, :, None]).permute(0, 2, 1)
    P = Program(dev)
    P.buf_floats.append(cin * 128)
    P.conv('t', Act(0, cin, 1, 128, cin), [w], [scale], [shift], 13, 1, True)
    op = P.ops[0]
    op.H = B                                                       # rows of the "image" = batch items
    xd = x.permute(0, 2, 1).contiguous().to(dev)                   # (B, L, C)
    lib = L.load()
    buf = C.create_string_buffer(128)
    runs = [(1, -1), (0, -1)] + [(0, v) for v in range(lib.romp_conv_num_variants())
                                 if lib.romp_conv_describe(C.byref(op), 1, v, buf, 128) == 0]
    for mode, variant in runs:
        out = torch.full((B, 128, cout), float('nan'), device=dev)
        L.check(lib.romp_conv_forward(C.byref(op), L.ptr(xd), None, L.ptr(out), 1, mode, variant, L.stream_ptr(dev)))
        torch.cuda.synchronize()
        err = (out.cpu() - ref).abs().max().item()
        print(f'mode {mode} variant {variant}: max-abs err {err:.3e}')
        assert err < 5e-5


def test_bev_localization_vs_oracle_and_golden(dev, golden_dir, bev_model):
    g = np.load(os.path.join(golden_dir, 'bev_b1.npz'))
    sd = BO.make_bev_state_dict(0)
    img = O.make_images(1, seed=4)
    c3d, cam3d = bev_model.localization(img.to(dev))
    assert c3d.shape == (1, 64, 128, 128) and cam3d.shape == (1, 3, 64, 128, 128)
    c, m = c3d.cpu().numpy(), cam3d.cpu().numpy()
    e1 = np.abs(c[0].reshape(-1)[g['sample_pos']] - g['center3d_samples']).max()
    e2 = np.abs(m[0].reshape(3, -1)[:, g['sample_pos']] - g['cam3d_samples']).max()
    print(f'vs reference samples: center3d {e1:.3e} cam3d {e2:.3e}')
    assert e1 < 1e-4 and e2 < 1e-4
    x = O.backbone_forward(sd, img)
    co, mo, _ = BO.coarse2fine_localization(sd, x)
    e3, e4 = np.abs(c - co.numpy()).max(), np.abs(m - mo.numpy()).max()
    print(f'vs oracle full maps: center3d {e3:.3e} cam3d {e4:.3e}')
    assert e3 < 1e-4 and e4 < 1e-4


@pytest.mark.parametrize('B,thresh', [(1, 0.999), (3, 0.9995), (2, 0.99)])
def test_bev_parse_vs_oracle(dev, B, thresh):
    """MaxPool3d(5) NMS + ordered top-K on random volumes incl. a plateau and the saturated case."""
    from romp_amd.bev import CenterMap3D
    gen = torch.Generator().manual_seed(B)
    cm = torch.rand(B, 64, 128, 128, generator=gen)
    cm[0, 10:12, 20:22, 30:32] = 2.0                                  # 2x2x2 plateau: 8 maxima (exact equality)
    bo, zo, so = BO.parse_3dcentermap(cm, thresh)
    bids, czyx, conf = CenterMap3D(thresh).parse_3dcentermap(cm.to(dev))
    assert np.array_equal(bids.cpu().numpy(), bo)
    assert np.array_equal(czyx.cpu().numpy(), zo)
    assert np.array_equal(conf.cpu().numpy(), so)
    assert czyx.dtype == torch.int64
    e = CenterMap3D(5.0).parse_3dcentermap(cm.to(dev))
    assert e[0].numel() == 0


def test_bev_forward_vs_golden(dev, golden_dir, bev_model):
    g = np.load(os.path.join(golden_dir, 'bev_b1.npz'))
    bev_model.centermap_parser.conf_thresh = float(g['thresh'])
    out = bev_model(O.make_images(1, seed=4).to(dev))
    assert out is not None

    def canon(b, zyx, conf):
        flat = (zyx[:, 0] * 128 + zyx[:, 1]) * 128 + zyx[:, 2]
        return np.lexsort((flat, -conf, b))
    zyx, conf = out['pred_czyxs'].cpu().numpy(), out['center_confs'].cpu().numpy()
    ko = canon(out['pred_batch_ids'].cpu().numpy(), zyx, conf)
    kg = canon(g['pred_batch_ids'], g['pred_czyxs'], g['center_confs'])
    assert np.array_equal(zyx[ko], g['pred_czyxs'][kg])
    np.testing.assert_allclose(conf[ko], g['center_confs'][kg], atol=1e-4)
    pp = out['params_pred'].cpu().numpy()[ko]
    e = np.abs(pp - g['params_pred'][kg]).max()
    print('params_pred max-abs vs reference', e, 'detections', len(ko))
    assert e < 2e-4
    assert np.array_equal(out['cam_czyx'].cpu().numpy()[ko], g['cam_czyx'][kg])
    np.testing.assert_allclose(out['smpl_thetas'].cpu().numpy()[ko], g['smpl_thetas'][kg], atol=5e-4)
    np.testing.assert_allclose(out['smpl_betas'].cpu().numpy()[ko], g['smpl_betas'][kg], atol=2e-4)
    np.testing.assert_allclose(out['cam_trans'].cpu().numpy()[ko], g['cam_trans'][kg], rtol=1e-4, atol=1e-4)
    bev_model.centermap_parser.conf_thresh = 1e3
    assert bev_model(O.make_images(1, seed=4).to(dev)) is None


def test_smpla_parser_golden(dev, golden_dir):
    from romp_amd.bev import SMPLA_parser
    g = np.load(os.path.join(golden_dir, 'smpla_parser_n5.npz'))
    parser = SMPLA_parser(O.make_synthetic_smpl(seed=0, n_betas=11), O.make_synthetic_smpl(seed=5, n_betas=10)).to(dev)
    v, j, _ = parser(torch.from_numpy(g['betas']).to(dev), torch.from_numpy(g['thetas']).to(dev))
    ev, ej = np.abs(v.cpu().numpy() - g['verts']).max(), np.abs(j.cpu().numpy() - g['joints']).max()
    print(f'SMPLA parser: verts {ev:.3e} joints {ej:.3e}')
    assert ev < 1e-4 and ej < 1e-4


def test_bev_api(dev, golden_dir):
    from romp_amd import bev
    g = np.load(os.path.join(golden_dir, 'bev_b1.npz'))
    s = bev.bev_settings([])
    s.GPU, s.center_thresh, s.max_batch = 0, float(g['thresh']), 2
    model = bev.BEV(s, state_dict=BO.make_bev_state_dict(0), smpla_model=O.make_synthetic_smpl(0, 11),
                    smil_model=O.make_synthetic_smpl(5, 10))
    res = model.forward_batch(O.make_images(2, seed=4).to(dev))
    N = res['cam'].shape[0]
    assert N >= len(g['pred_batch_ids'])
    assert res['verts'].shape == (N, 6890, 3) and res['joints'].shape == (N, 71, 3) and res['smpl_betas'].shape == (N, 11)
    th, be = res['smpl_thetas'].cpu().numpy(), res['smpl_betas'].cpu().numpy()
    vo, jo = BO.smpla_forward(O.make_synthetic_smpl(0, 11), O.make_synthetic_smpl(5, 10), be, th)
    assert np.abs(res['verts'].cpu().numpy() - vo).max() < 1e-4
    rs = np.random.RandomState(0)
    out = model(rs.randint(0, 256, (300, 500, 3)).astype(np.uint8))
    assert out is None or isinstance(out['verts'], np.ndarray)
