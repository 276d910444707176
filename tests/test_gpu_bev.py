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


def test_bev_benchmark_batch_vs_oracle(dev):
    """BASELINE configs[3] at the size bench.py times it (B=32, conv_math f16x2, variants autotuned AT 32): images 0 and 19 of the
    batch -- 3-D centre maps and camera maps against the oracle (1e-4), and the detections / regressed parameters of those two
    images against the oracle's BEV forward (same set, params 2e-4)."""
    from romp_amd.bev import BEVv1
    sd = BO.make_bev_state_dict(0)
    model = BEVv1(sd, dev, center_thresh=0.1, max_batch=32, bf16x3='f16x2')
    model.net.autotune(32, iters=1)
    assert sum('conv_h2' in n for n in model.net.variant_names(32)) > 0
    img = O.make_images(32, seed=4)
    x = img.to(dev)
    c3d, cam3d = model.localization(x)
    pick = [0, 19]
    xo = O.backbone_forward(sd, img[pick])
    co, mo, _ = BO.coarse2fine_localization(sd, xo)
    e3 = np.abs(c3d[pick].cpu().numpy() - co.numpy()).max()
    e4 = np.abs(cam3d[pick].cpu().numpy() - mo.numpy()).max()
    print(f'BEV B=32 f16x2, images {pick} vs oracle: center3d {e3:.3e} cam3d {e4:.3e}')
    assert e3 < 1e-4 and e4 < 1e-4
    # detections + regression of the two images: threshold half-way between the 12th and 13th strongest oracle maximum of image 0
    bo, zo, so = BO.parse_3dcentermap(co, 0.0 + 1e-6)
    s0 = np.sort(so[bo == 0])[::-1]
    thresh = float(0.5 * (s0[min(11, len(s0) - 2)] + s0[min(12, len(s0) - 1)]))
    model.centermap_parser.conf_thresh = thresh
    out = model(x)
    ref = BO.bev_forward(sd, img[pick], thresh)
    assert out is not None and ref is not None
    b = out['pred_batch_ids'].cpu().numpy()
    rows = np.concatenate([np.nonzero(b == p)[0] for p in pick])

    def canon(bb, zyx):
        return np.lexsort(((zyx[:, 0] * 128 + zyx[:, 1]) * 128 + zyx[:, 2], bb))
    zyx = out['pred_czyxs'].cpu().numpy()[rows]
    ko = canon(np.searchsorted(pick, b[rows]), zyx)
    kr = canon(ref['pred_batch_ids'], ref['pred_czyxs'])
    assert np.array_equal(zyx[ko], ref['pred_czyxs'][kr]), 'detections differ'
    e = np.abs(out['params_pred'].cpu().numpy()[rows][ko] - ref['params_pred'][kr]).max()
    print(f'BEV B=32: {len(rows)} detections in the 2 images, params_pred max-abs vs oracle {e:.3e}')
    assert e < 2e-4


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


def test_bev_temporal_and_render(dev):
    """BEV(settings: -t --render_mesh) over a short clip (bev/main.py:162-166,260-287,147-150): every reported person carries a
    track id; thetas / betas / cam are the OneEuro-filtered values of that track (oracle filters fed with the raw per-frame
    estimates); meshes come from the smoothed parameters; 'rendered_image' = [frame | Sim3DR rendering] equals the oracle
    renderer on the oracle's perspective projection of the returned meshes, bit for bit."""
    from oracle import sim3dr_oracle as SO
    from oracle import temporal_oracle as TO
    from romp_amd import bev, tracker
    from romp_amd.vis import mesh_color_left2right
    tracker.Track.last_id = 0
    s = bev.bev_settings(['-t', '--render_mesh'])
    s.GPU, s.center_thresh, s.max_batch = 0, 0.9995, 2
    smpla, smil = O.make_synthetic_smpl(0, 11), O.make_synthetic_smpl(5, 10)
    _, base_tri = SO.ellipsoid_mesh(84, 82, [0, 0, 0], [1, 1, 1])
    faces = np.zeros((13776, 3), np.int64)
    faces[:len(base_tri)] = base_tri
    smpla, smil = dict(smpla, f=torch.from_numpy(faces).float()), dict(smil, f=torch.from_numpy(faces).float())
    model = bev.BEV(s, state_dict=BO.make_bev_state_dict(0), smpla_model=smpla, smil_model=smil)
    seen = []
    inner = model.temporal_optimization

    def spy(outputs, signal_ID):
        raw = {k: outputs[k].clone() for k in ('smpl_thetas', 'smpl_betas', 'cam', 'params_pred')}
        res = inner(outputs, signal_ID)
        seen.append((raw, None if res is None else {k: (res[k].clone() if torch.is_tensor(res[k]) else res[k].copy())
                                                   for k in ('smpl_thetas', 'smpl_betas', 'cam', 'params_pred', 'track_ids')}))
        return res

    model.temporal_optimization = spy
    rs = np.random.RandomState(2)
    frame0 = rs.randint(0, 256, (360, 640, 3)).astype(np.uint8)
    from romp_amd.utils import img_preprocess_device
    for thresh in (0.9995, 0.999, 0.99, 0.9, 0.5, 0.1):                     # the highest threshold that still sees a few persons
        model.model.centermap_parser.conf_thresh = thresh
        probe = model.model(img_preprocess_device(frame0, dev)[0])
        if probe is not None and probe['cam'].shape[0] >= 3:
            break
    print('center threshold', thresh, 'persons on the first frame', probe['cam'].shape[0])
    filters, all_ids = {}, set()
    for f in range(4):
        frame = np.clip(frame0.astype(np.int32) + rs.randint(-5, 6, frame0.shape), 0, 255).astype(np.uint8)
        out = model(frame)
        raw, sm = seen[-1]
        assert sm is not None and out is not None
        ids = sm['track_ids'].tolist()
        assert len(set(ids)) == len(ids)
        if f == 0:
            assert sorted(ids) == list(range(1, len(ids) + 1))            # every first-frame detection starts a confirmed track
        all_ids |= set(ids)
        worst = 0.
        for r, tid in enumerate(ids):
            src = torch.where((raw['params_pred'] == sm['params_pred'][r]).all(1))[0]
            assert len(src) >= 1
            k = int(src[0])
            flt = filters.setdefault(tid, TO.make_filters(s.smooth_coeff))
            t, b, c = TO.smooth(flt, raw['smpl_thetas'][k].cpu(), raw['smpl_betas'][k].cpu(), raw['cam'][k].cpu())
            worst = max(worst, float((sm['smpl_thetas'][r].cpu() - t).abs().max()), float((sm['smpl_betas'][r].cpu() - b).abs().max()),
                        float((sm['cam'][r].cpu() - c).abs().max()))
        print('frame %d: %d tracked persons, smoothed vs oracle filters max-abs %.3e' % (f, len(ids), worst))
        assert worst < 5e-5
        # what is returned: the survivors of duplicate suppression / outlier removal, ids kept in step with the rows
        n = len(out['track_ids'])
        assert out['cam'].shape == (n, 3) and out['verts'].shape == (n, 6890, 3) and set(out['track_ids'].tolist()) <= set(ids)
        vo, _ = BO.smpla_forward(smpla, smil, out['smpl_betas'], out['smpl_thetas'])
        assert np.abs(out['verts'] - vo).max() < 1e-4
        ct = BO.cam_to_trans(out['cam'])
        assert np.abs(out['cam_trans'] - ct).max() < 1e-4 * max(1., np.abs(ct).max())
        # rendering
        assert out['rendered_image'].shape == (360, 1280, 3) and np.array_equal(out['rendered_image'][:, :640], frame)
        F32 = np.float32
        verts, tr = out['verts'].astype(F32), out['cam_trans'].astype(F32)
        p = verts + tr[:, None]
        z = p[..., 2] + F32(1e-6)
        px, py = p[..., 0] / z * F32(443.4) / F32(256), p[..., 1] / z * F32(443.4) / F32(256)
        pad = F32(640)
        top, left = F32((640 - 360) // 2), F32(0)
        vorg = np.stack([(px + F32(1)) * pad / F32(2) - left, (py + F32(1)) * pad / F32(2) - top, (verts[..., 2] + F32(1)) * pad / F32(2)], -1).astype(F32)
        order = torch.sort(torch.from_numpy(tr[:, 2]), descending=True).indices.numpy()
        vorg = vorg[order]
        vorg[:, :, 2] *= -1
        colors = mesh_color_left2right(torch.from_numpy(tr))[order]
        ref = SO.render_meshes(vorg, faces.astype(np.int32), frame, colors, use_ref=SO.load_ref() is not None)
        nd = int((out['rendered_image'][:, 640:] != ref).sum())
        print('   rendered_image vs oracle pipeline: differing bytes %d, painted px %d' % (nd, int((ref != frame).any(2).sum())))
        assert nd <= 3 * 8                                                  # a vertex within 1 ulp of a pixel centre may flip a pixel
    assert len(all_ids) >= 1


def test_bev_plan_file_roundtrip(dev, bev_model, tmp_path):
    """The BEV program carries HOST-side constant tables (scale anchors, the Conv3d refiners' weights) next to its device
    constants: both blobs of the plan file; romp_net_load gives bit-identical 3-D maps."""
    from romp_amd.export import save_plan
    from romp_amd.net import RompNet
    img = O.make_images(2, seed=4).to(dev)
    c0, m0 = bev_model.localization(img)
    path = str(tmp_path / 'bev.plan')
    save_plan(bev_model.net, path)
    net2 = RompNet.from_plan(path, dev, max_batch=2, out_shapes=bev_model.net.out_shapes)
    c1, m1 = net2.forward_nhwc(img)
    assert c1.shape == c0.shape and torch.equal(c1, c0) and torch.equal(m1, m0)
