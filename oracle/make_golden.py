"""Generate the golden fixtures under tests/golden/ by running the REFERENCE ITSELF.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference, which does
not exist on the GPU box).  The reference's own modules are imported read-only by file
path: ``model.py`` and ``smpl.py`` import cleanly; ``post_parser.py`` /
``utils.py`` need ``import cv2`` to succeed, so an empty stub module is registered (cv2 is
only used by I/O + PnP functions that are not on the gated path).

Inputs are the seeded synthetic tensors of ``oracle/romp_oracle.py`` (identical generators
are re-run by the tests), outputs are what the reference computes on them.

    python oracle/make_golden.py          # rewrites tests/golden/*.npz
"""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import romp_oracle as O  # noqa: E402

REF = '/root/reference/simple_romp'
GOLD = os.path.join(ROOT, 'tests', 'golden')


def _load_reference():
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    pkg = types.ModuleType('romp')
    pkg.__path__ = [os.path.join(REF, 'romp')]
    sys.modules['romp'] = pkg
    mods = {}
    for name in ('model', 'smpl', 'utils', 'post_parser'):
        spec = importlib.util.spec_from_file_location(f'romp.{name}', os.path.join(REF, 'romp', f'{name}.py'))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f'romp.{name}'] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


def golden_net(ref):
    sd = O.make_romp_state_dict(0)
    net = ref['model'].ROMPv1().eval()
    ref_sd = net.state_dict()
    ref_keys = [k for k in ref_sd if not k.endswith('num_batches_tracked')]
    assert ref_keys == list(sd.keys()), 'oracle param spec != reference state_dict keys'
    for k in ref_keys:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    net.load_state_dict(sd, strict=False)   # only num_batches_tracked is absent
    img = O.make_images(1, seed=1)
    feats = {}
    net.backbone.register_forward_hook(lambda m, i, o: feats.__setitem__('backbone', o.detach().clone()))
    with torch.no_grad():
        cm, pm = net(img)
    rs = np.random.RandomState(7)
    pos = np.sort(rs.choice(64 * 64, 256, replace=False))
    fpos = np.sort(rs.choice(128 * 128, 256, replace=False))
    feat = feats['backbone'][0].reshape(32, -1).numpy()
    np.savez_compressed(
        os.path.join(GOLD, 'romp_net_b1.npz'),
        center_maps=cm.numpy(), sample_pos=pos,
        params_samples=pm[0].reshape(145, -1).numpy()[:, pos],
        params_chan_sum=pm[0].reshape(145, -1).double().sum(1).numpy(),
        params_chan_abs=pm[0].reshape(145, -1).double().abs().sum(1).numpy(),
        feat_pos=fpos, feat_samples=feat[:, fpos],
        feat_chan_sum=feats['backbone'][0].reshape(32, -1).double().sum(1).numpy())
    print('net: center range', float(cm.min()), float(cm.max()))
    return cm, pm


def golden_parse(ref):
    g = torch.Generator().manual_seed(11)
    cm = torch.rand(3, 1, 64, 64, generator=g)
    pm = torch.randn(3, 145, 64, 64, generator=g)
    cm[2] *= 0.2      # image 2: nothing above threshold
    thresh = 0.985
    pm2 = pm.clone()
    pm2[:, 0] = torch.pow(1.1, pm2[:, 0])                        # main.py:113
    parser = ref['post_parser'].CenterMap(conf_thresh=thresh)
    out = ref['post_parser'].parsing_outputs(cm, pm2, parser)
    bids, finds, _, scores = parser.parse_centermap(cm)
    # canonical order: batch-major, score desc, flat index asc (tie order is unspecified upstream)
    key = np.lexsort((finds.numpy(), -scores.numpy(), bids.numpy()))
    sel = lambda t: t.numpy()[key]
    np.savez_compressed(
        os.path.join(GOLD, 'parse_b3.npz'), thresh=np.float32(thresh),
        batch_ids=sel(bids), flat_inds=sel(finds), scores=sel(scores),
        cam=sel(out['cam']), global_orient=sel(out['global_orient']), body_pose=sel(out['body_pose']),
        smpl_betas=sel(out['smpl_betas']), smpl_thetas=sel(out['smpl_thetas']),
        center_preds=sel(out['center_preds']), center_confs=sel(out['center_confs']))
    print('parse: detections', len(bids), 'per image', np.bincount(bids.numpy(), minlength=3))
    # empty case must return None (post_parser.py:138-140)
    assert ref['post_parser'].parsing_outputs(cm * 0.01, pm2, parser) is None


def rot6d_cases():
    """6D inputs that reach every branch of rotation_matrix_to_quaternion / quaternion_to_angle_axis."""
    g = torch.Generator().manual_seed(5)
    x = [torch.randn(64, 6, generator=g)]
    def from_R(R):   # interleaved 6D of the first two columns (x.view(-1,3,2))
        return torch.stack([R[:, 0], R[:, 1]], -1).reshape(1, 6)
    eye = torch.eye(3)
    x.append(from_R(eye))                                            # identity: sin^2 == 0 branch
    for ax in range(3):                                              # 180deg about each axis
        R = -torch.eye(3); R[ax, ax] = 1
        x.append(from_R(R))
    for ang in (1e-4, 3.1, 3.14159, 2.0, -2.5):
        for ax in range(3):
            c, s = np.cos(ang), np.sin(ang)
            R = torch.eye(3)
            i, j = [(1, 2), (0, 2), (0, 1)][ax]
            R[i, i], R[i, j], R[j, i], R[j, j] = c, -s, s, c
            x.append(from_R(R))
    x.append(torch.zeros(1, 6))                                      # degenerate -> NaN -> 0
    x.append(torch.tensor([[1., 1., 0., 0., 0., 0.]]))               # a2 parallel a1
    return torch.cat(x, 0).float()


def golden_rot6d(ref):
    x = rot6d_cases()
    aa = ref['utils'].rot6D_to_angular(x.clone())
    R = ref['utils'].rot6d_to_rotmat(x.clone())
    np.savez_compressed(os.path.join(GOLD, 'rot6d_cases.npz'), x=x.numpy(), aa=aa.numpy(), rotmat=R.numpy())
    print('rot6d cases', x.shape[0])


def golden_smpl(ref):
    for nb, tag in ((10, 'smpl'), (11, 'smpla')):
        model = O.make_synthetic_smpl(seed=0, n_betas=nb)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, 'm.pth')
            torch.save(model, path)
            smpl = ref['smpl'].SMPL(path, model_type=tag)
        g = torch.Generator().manual_seed(3)
        betas = torch.randn(4, nb, generator=g)
        poses = 0.3 * torch.randn(4, 72, generator=g)
        poses[3, 6:9] = 0.0                                          # exact zero rotation vector
        res = {}
        for ra in (False, True):
            v, j, _ = smpl(betas, poses, root_align=ra)
            res[f'verts_ra{int(ra)}'] = v.numpy()
            res[f'joints_ra{int(ra)}'] = j.numpy()
        np.savez_compressed(os.path.join(GOLD, f'{tag}_n4.npz'), betas=betas.numpy(), poses=poses.numpy(), **res)
        print(tag, 'verts absmax', float(np.abs(res['verts_ra0']).max()))


def golden_projection(ref):
    g = torch.Generator().manual_seed(9)
    j = torch.randn(3, 71, 3, generator=g)
    cam = torch.rand(3, 3, generator=g) + 0.2
    pad = torch.Tensor([280, 1000, 0, 1280, 720, 1280])
    pj = ref['utils'].batch_orth_proj(j, cam, mode='2d')
    org = ref['post_parser'].convert_proejection_from_input_to_orgimg(pj.clone(), pad)
    tr = ref['utils'].convert_cam_to_3d_trans(cam)
    np.savez_compressed(os.path.join(GOLD, 'projection.npz'), joints=j.numpy(), cam=cam.numpy(),
                        pad=pad.numpy(), pj2d=pj.numpy(), pj2d_org=org.numpy(), cam_trans=tr.numpy())


if __name__ == '__main__':
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    ref = _load_reference()
    golden_rot6d(ref)
    golden_parse(ref)
    golden_smpl(ref)
    golden_projection(ref)
    golden_net(ref)
    print('golden fixtures written to', GOLD)


# ------------------------------------------------------------------------------------------- BEV
def _load_reference_bev():
    pkg = types.ModuleType('bev')
    pkg.__path__ = [os.path.join(REF, 'bev')]
    sys.modules['bev'] = pkg
    mods = {}
    for name in ('post_parser', 'model'):
        spec = importlib.util.spec_from_file_location(f'bev.{name}', os.path.join(REF, 'bev', f'{name}.py'))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f'bev.{name}'] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


def golden_bev():
    """Reference BEVv1 (bev/model.py:104-250) on the oracle's synthetic BEV weights."""
    from oracle import bev_oracle as BO
    bev = _load_reference_bev()
    sd = BO.make_bev_state_dict(0)
    img = O.make_images(1, seed=4)
    net = bev['model'].BEVv1(center_thresh=0.1).eval()
    ref_keys = [k for k in net.state_dict() if not k.endswith('num_batches_tracked')]
    assert sorted(ref_keys) == sorted(sd.keys()), set(ref_keys) ^ set(sd.keys())
    for k in ref_keys:
        assert tuple(net.state_dict()[k].shape) == tuple(sd[k].shape), k
    net.load_state_dict(sd, strict=False)
    with torch.no_grad():
        x = net.backbone(img)
        c3d, cam3d, c_fv = net.coarse2fine_localization(x)
    # threshold with ~12 detections on this input
    mx = torch.nn.functional.max_pool3d(c3d.unsqueeze(1), 5, 1, 2).squeeze(1)
    peaks = torch.sort(c3d[(mx == c3d)].flatten(), descending=True)[0]
    gaps = (peaks[7:24] - peaks[8:25]).numpy()                  # cut at the widest gap among the top peaks
    n_det = 8 + int(np.argmax(gaps))
    thresh = float((peaks[n_det - 1] + peaks[n_det]) / 2)
    assert thresh > 0 and gaps.max() > 1e-4
    net.centermap_parser.conf_thresh = thresh
    with torch.no_grad():
        out = net(img)
    pk = bev['post_parser'].pack_params_dict(out['params_pred'])
    trans = bev['post_parser'].denormalize_cam_params_to_trans(pk['cam'])
    rs = np.random.RandomState(3)
    pos = np.sort(rs.choice(64 * 128 * 128, 4096, replace=False))
    np.savez_compressed(
        os.path.join(GOLD, 'bev_b1.npz'), thresh=np.float32(thresh), sample_pos=pos,
        center3d_samples=c3d[0].reshape(-1).numpy()[pos], cam3d_samples=cam3d[0].reshape(3, -1).numpy()[:, pos],
        center3d_sum=np.float64(c3d.double().sum()), cam3d_sum=cam3d[0].reshape(3, -1).double().sum(1).numpy(),
        center_fv=c_fv[0, 0].numpy(),
        pred_batch_ids=out['pred_batch_ids'].numpy(), pred_czyxs=out['pred_czyxs'].numpy(),
        center_confs=out['center_confs'].numpy(), params_pred=out['params_pred'].numpy(), cam_czyx=out['cam_czyx'].numpy(),
        cam=pk['cam'].numpy(), smpl_thetas=pk['smpl_thetas'].numpy(), smpl_betas=pk['smpl_betas'].numpy(), cam_trans=trans.numpy())
    print('bev: thresh %.4f detections %d' % (thresh, len(out['pred_batch_ids'])))
    # SMPLA_parser with one "baby" (betas[:,10] > 0.8): bev/post_parser.py:255-278
    smpla = O.make_synthetic_smpl(seed=0, n_betas=11)
    smil = O.make_synthetic_smpl(seed=5, n_betas=10)
    with tempfile.TemporaryDirectory() as td:
        pa, pi = os.path.join(td, 'a.pth'), os.path.join(td, 'i.pth')
        torch.save(smpla, pa); torch.save(smil, pi)
        parser = bev['post_parser'].SMPLA_parser(pa, pi)
    g = torch.Generator().manual_seed(6)
    betas = torch.randn(5, 11, generator=g) * 0.5
    betas[:, 10] = torch.tensor([0.1, 0.95, 0.3, 0.85, -0.2])
    thetas = 0.3 * torch.randn(5, 72, generator=g)
    v, j, _ = parser(betas, thetas)
    np.savez_compressed(os.path.join(GOLD, 'smpla_parser_n5.npz'), betas=betas.numpy(), thetas=thetas.numpy(),
                        verts=v.numpy(), joints=j.numpy())


if __name__ == '__main__' and os.environ.get('GOLDEN_ONLY', '') in ('', 'bev'):
    golden_bev()
