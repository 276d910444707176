"""Golden fixture for BEV post-processing, produced by the reference's own functions
(bev/post_parser.py).  Build container only.   python oracle/make_golden_bev_post.py"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location('mg', os.path.join(ROOT, 'oracle', 'make_golden.py'))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)


def make_inputs(seed=0):
    g = torch.Generator().manual_seed(seed)
    N = 9
    joints = 0.25 * torch.randn(N, 71, 3, generator=g)
    cam = torch.zeros(N, 3)
    cam[:, 0] = torch.tensor([0.9, 0.88, 0.5, 0.45, 0.7, 0.12, 0.6, 0.55, 0.8])       # scale
    cam[:, 1:] = torch.rand(N, 2, generator=g) * 1.2 - 0.6
    joints[1] = joints[0] + 0.002 * torch.randn(71, 3, generator=g)                   # near-duplicate of person 0
    cam[1, 1:] = cam[0, 1:] + 0.003
    joints[7] = joints[6] + 0.002 * torch.randn(71, 3, generator=g)                   # near-duplicate of person 6
    cam[7, 1:] = cam[6, 1:] - 0.002
    cam[5, 1:] = torch.tensor([0.95, -0.9])                                           # tiny + remote: outlier candidate
    pad = torch.Tensor([280, 1000, 0, 1280, 720, 1280])
    return joints, cam, pad


if __name__ == '__main__':
    mg._load_reference()
    bev = mg._load_reference_bev()
    pp = bev['post_parser']
    joints, cam, pad = make_inputs()
    N = cam.shape[0]
    outputs = {'joints': joints.clone(), 'cam': cam.clone(), 'params_pred': torch.zeros(N, 146), 'idx': torch.arange(N)}
    outputs.update(pp.body_mesh_projection2image(outputs['joints'], outputs['cam'], input2org_offsets=pad))
    pj_org_all, trans_all = outputs['pj2d_org'].clone(), outputs['cam_trans'].clone()
    outputs = pp.suppressing_redundant_prediction_via_projection(outputs, (720, 1280, 3), thresh=20)
    after_nms = outputs['idx'].clone()
    outputs = pp.remove_outlier(outputs, relative_scale_thresh=1.6)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'bev_post.npz'), joints=joints.numpy(), cam=cam.numpy(),
                        pad=pad.numpy(), pj2d_org=pj_org_all.numpy(), cam_trans=trans_all.numpy(),
                        kept_after_nms=after_nms.numpy(), kept_final=outputs['idx'].numpy())
    print('kept after nms', after_nms.tolist(), 'final', outputs['idx'].tolist())
