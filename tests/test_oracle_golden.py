"""Pin the CPU oracle (oracle/romp_oracle.py) to outputs of the reference itself.

The fixtures in tests/golden/ were produced by oracle/make_golden.py, which runs the
reference's own modules on the same seeded inputs.  Tolerances: the conv network is the
same ATen arithmetic (bit-exact expected, 1e-6 allowed for thread-count dependent
summation order); numpy restatements of SMPL / rotation maths differ from torch only by
float32 summation order (tolerances stated per test).
"""
import os

import numpy as np
import pytest
import torch

from oracle import romp_oracle as O


def _ld(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_param_spec_matches_reference_count():
    spec = O.romp_param_spec()
    assert len(spec) == 1544            # 1851 state-dict entries - 307 num_batches_tracked
    assert sum(int(np.prod(s)) for s, _ in spec.values()) == 29104530   # 29.1 M params
    n_bn = sum(1 for _, k in spec.values() if k == 'bn_w')
    assert n_bn == 307
    n_conv = sum(1 for _, k in spec.values() if k == 'conv_w')
    assert n_conv == 310


def test_rot6d_cases(golden_dir):
    g = _ld(golden_dir, 'rot6d_cases.npz')
    R = O.rot6d_to_rotmat(g['x'])
    ok = np.isfinite(g['rotmat']).all(axis=(1, 2))
    np.testing.assert_allclose(R[ok], g['rotmat'][ok], atol=2e-6, rtol=0)
    aa = O.rot6d_to_angular(g['x'])
    assert np.isfinite(aa).all()
    # near-pi rotations are ill-conditioned: compare via the rotation they encode
    err = np.abs(aa - g['aa']).max(1)
    loose = err > 1e-5
    assert loose.sum() <= 12
    Ra = O.batch_rodrigues(aa[loose])
    Rb = O.batch_rodrigues(g['aa'][loose])
    np.testing.assert_allclose(Ra, Rb, atol=2e-3)
    np.testing.assert_allclose(aa[~loose], g['aa'][~loose], atol=1e-5)


def test_parse_against_reference(golden_dir):
    g = _ld(golden_dir, 'parse_b3.npz')
    gen = torch.Generator().manual_seed(11)
    cm = torch.rand(3, 1, 64, 64, generator=gen)
    pm = torch.randn(3, 145, 64, 64, generator=gen)
    cm[2] *= 0.2
    out = O.parsing_outputs(cm.numpy(), pm.numpy(), float(g['thresh']))
    assert np.array_equal(out['batch_ids'], g['batch_ids'])
    assert np.array_equal(out['flat_inds'], g['flat_inds'])
    assert np.array_equal(out['scores'], g['scores'])
    assert np.array_equal(out['center_preds'], g['center_preds'])
    assert np.array_equal(out['center_confs'], g['center_confs'])
    np.testing.assert_allclose(out['cam'], g['cam'], atol=1e-6, rtol=1e-6)
    assert np.array_equal(out['smpl_betas'], g['smpl_betas'])
    np.testing.assert_allclose(out['smpl_thetas'], g['smpl_thetas'], atol=2e-5)
    np.testing.assert_allclose(out['body_pose'], g['body_pose'], atol=2e-5)
    assert out['body_pose'].shape[1] == 69 and out['smpl_thetas'].shape[1] == 72
    # no detection -> None (post_parser.py:138-140)
    assert O.parsing_outputs(cm.numpy() * 0.01, pm.numpy(), float(g['thresh'])) is None


@pytest.mark.parametrize('tag,nb', [('smpl', 10), ('smpla', 11)])
def test_smpl_against_reference(golden_dir, tag, nb):
    g = _ld(golden_dir, f'{tag}_n4.npz')
    model = O.make_synthetic_smpl(seed=0, n_betas=nb)
    for ra in (0, 1):
        v, j, _ = O.smpl_forward(model, g['betas'], g['poses'], root_align=bool(ra))
        assert v.shape == (4, 6890, 3) and j.shape == (4, 71, 3)
        # float32 summation-order noise floor is ~3e-7; gate 5e-6
        np.testing.assert_allclose(v, g[f'verts_ra{ra}'], atol=5e-6, rtol=0)
        np.testing.assert_allclose(j, g[f'joints_ra{ra}'], atol=5e-6, rtol=0)


def test_projection_against_reference(golden_dir):
    g = _ld(golden_dir, 'projection.npz')
    pj = O.batch_orth_proj(g['joints'], g['cam'])
    np.testing.assert_allclose(pj, g['pj2d'], atol=1e-6)
    np.testing.assert_allclose(O.project_to_org_image(pj, g['pad']), g['pj2d_org'], atol=1e-3, rtol=1e-6)
    np.testing.assert_allclose(O.convert_cam_to_3d_trans(g['cam']), g['cam_trans'], rtol=1e-6)


def test_net_against_reference(golden_dir):
    g = _ld(golden_dir, 'romp_net_b1.npz')
    sd = O.make_romp_state_dict(0)
    img = O.make_images(1, seed=1)
    feat = O.backbone_forward(sd, img)
    cm, pm = O.head_forward(sd, feat)
    assert cm.shape == (1, 1, 64, 64) and pm.shape == (1, 145, 64, 64)
    f = feat[0].reshape(32, -1).numpy()
    np.testing.assert_allclose(f[:, g['feat_pos']], g['feat_samples'], atol=2e-5)
    np.testing.assert_allclose(cm.numpy(), g['center_maps'], atol=2e-5)
    p = pm[0].reshape(145, -1)
    np.testing.assert_allclose(p.numpy()[:, g['sample_pos']], g['params_samples'], atol=2e-5)
    np.testing.assert_allclose(p.double().sum(1).numpy(), g['params_chan_sum'], atol=2e-2)
