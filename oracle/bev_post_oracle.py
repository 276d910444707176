"""CPU oracle for BEV's per-image post-processing -- TEST INFRASTRUCTURE ONLY.

Restates simple_romp/bev/post_parser.py: denormalize_cam_params_to_trans :114-128,
perspective_projection :68-107, convert_proejection_from_input_to_orgimg :129-136,
suppressing_redundant_prediction_via_projection :167-198, remove_outlier :200-222 (numpy float32).
Pinned by tests/golden/bev_post.npz (oracle/make_golden_bev_post.py runs the reference itself).
"""
import numpy as np

TAN_FOV = np.float32(np.tan(np.radians(60 / 2.)))


def cam_to_trans(cam):
    cam = np.asarray(cam, np.float32)
    depth = (np.float32(1) / (cam[:, 0] * TAN_FOV + np.float32(1e-3)))[:, None]
    return np.concatenate([cam[:, 1:][:, ::-1] * depth * TAN_FOV, depth], 1).astype(np.float32)


def project(joints, cam, pad_info):
    """-> pj2d (normalised), pj2d_org (original-image pixels), cam_trans."""
    t = cam_to_trans(cam)
    p = np.asarray(joints, np.float32) + t[:, None]
    pj = p[:, :, :2] / (p[:, :, 2:3] + np.float32(1e-6)) * np.float32(443.4) / np.float32(256.0)
    top, bottom, left, right, h, w = [np.float32(v) for v in pad_info]
    s = max(h, w)
    org = np.stack([(pj[:, :, 0] + 1) * s / 2 - left, (pj[:, :, 1] + 1) * s / 2 - top], -1).astype(np.float32)
    return pj.astype(np.float32), org, t


def postprocess(joints, cam, pad_info, nms_thresh=20.0, relative_scale_thresh=1.6, scale_thresh=0.25):
    """-> dict(pj2d, pj2d_org, cam_trans, keep (bool mask))."""
    cam = np.asarray(cam, np.float32)
    pj, org, t = project(joints, cam, pad_info)
    N = cam.shape[0]
    removed = np.zeros(N, bool)
    if N > 1:
        d = np.sqrt(((org[:, None] - org[None]) ** 2).sum(-1)).mean(-1)
        sc = cam[:, 0] * 2
        d = d / np.maximum(sc[:, None], sc[None])
        thr = nms_thresh * max(float(pad_info[4]), float(pad_info[5])) / 640.
        for a in range(N):
            for b in range(a + 1, N):
                if d[a, b] < thr:
                    removed[a if sc[a] < sc[b] else b] = True
    alive = np.nonzero(~removed)[0]
    m = alive.size
    if m >= 3:
        tt = t[alive]
        dm = np.sqrt(((tt[:, None] - tt[None]) ** 2).sum(-1))
        mean = np.sort(dm, 1)[:, 1:-1].mean(1)
        rel = mean / ((mean.sum() - mean) / (m - 1))
        out = (rel > relative_scale_thresh) & (cam[alive, 0] < scale_thresh)
        removed[alive[out]] = True
    return {'pj2d': pj, 'pj2d_org': org, 'cam_trans': t, 'keep': ~removed}
