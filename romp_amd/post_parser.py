"""Host-side mirror of simple_romp/romp/post_parser.py for the HIP path.

Same names and argument meaning as the reference (``CenterMap``, ``parsing_outputs``,
``SMPL_parser``, ``body_mesh_projection2image``), but every arithmetic step is one call
into libromp_hip.so (csrc/parse.hip, csrc/smpl.hip).
"""
import ctypes as C

import numpy as np
import torch
from torch import nn

from . import lib as L
from .smpl import SMPL


class CenterMap(object):
    """post_parser.py:8-25 -- holds the parse configuration (size 64, K = 64 persons)."""

    def __init__(self, conf_thresh):
        self.size = 64
        self.max_person = 64
        self.sigma = 1
        self.conf_thresh = conf_thresh

    def parse_centermap(self, center_maps, params_maps_nhwc=None):
        """post_parser.py:27-47.  Returns batch_ids, flat_inds, center_yxs, scores (device tensors).
        Order: batch-major, score-descending inside an image (ties: lower flat index)."""
        r, _ = _parse(center_maps, params_maps_nhwc, self.conf_thresh, self.max_person)
        if r is None:
            e = torch.empty(0, device=center_maps.device)
            return e.long(), e.long(), e.reshape(0, 2), e
        yx = torch.stack([(r['flat_inds'] // self.size).float(), (r['flat_inds'] % self.size).float()], 1)
        return r['batch_ids'], r['flat_inds'], yx, r['scores']


def _parse(center_maps, params_maps_nhwc, conf_thresh, max_person, watch=None):
    """-> (dict of row tensors or None, watched word or None).  `watch`: device address of an int32 that rides back with the
    counts (romp_parse_watch; RompNet.sat_counter for the API's range guard)."""
    lib = L.load()
    dev = center_maps.device
    if dev.type != 'cuda':
        raise L.RompHipError('parsing runs on the HIP device only (no CPU fallback)')
    cm = center_maps.reshape(center_maps.shape[0], 64, 64).contiguous().float()
    B = cm.shape[0]
    if params_maps_nhwc is None:
        params_maps_nhwc = torch.zeros(B, 64, 64, 145, device=dev)
    pm = params_maps_nhwc.contiguous()
    assert pm.shape == (B, 64, 64, 145) and pm.dtype == torch.float32
    cap = B * max_person
    # one allocation per dtype, sliced: the single-image path pays for every torch.empty
    ibuf = torch.empty(cap * 4 + B * (2 * max_person + 2) + 2, device=dev, dtype=torch.int32)
    fbuf = torch.empty(cap * (1 + 145 + 3 + 72 + 10), device=dev, dtype=torch.float32)
    out, at = {}, 0
    for key, w in (('scores', 1), ('params_pred', 145), ('cam', 3), ('smpl_thetas', 72), ('smpl_betas', 10)):
        out[key] = fbuf[at:at + cap * w].view(cap, w) if w > 1 else fbuf[at:at + cap]
        at += cap * w
    out['batch_ids'], out['flat_inds'] = ibuf[:cap], ibuf[cap:2 * cap]
    out['center_preds'] = ibuf[2 * cap:4 * cap].view(cap, 2)
    ws = ibuf[4 * cap:]
    n, w = C.c_int32(0), C.c_int32(0)
    with torch.cuda.device(dev):
        L.check(lib.romp_parse_watch(L.ptr(cm), L.ptr(pm), B, float(conf_thresh), int(max_person), C.byref(n),
                                     L.ptr(out['batch_ids']), L.ptr(out['flat_inds']), L.ptr(out['scores']),
                                     L.ptr(out['params_pred']), L.ptr(out['cam']), L.ptr(out['smpl_thetas']),
                                     L.ptr(out['smpl_betas']), L.ptr(out['center_preds']), L.ptr(ws), L.stream_ptr(dev),
                                     C.c_void_p(watch or 0), C.byref(w)))
    N = n.value
    seen = w.value if watch else None
    if N == 0:
        return None, seen
    out = {k: v[:N] for k, v in out.items()}
    il = ibuf[:4 * cap].long()                                # the reference's index tensors are int64: one cast for the three
    out['batch_ids'], out['flat_inds'], out['center_preds'] = il[:N], il[cap:cap + N], il[2 * cap:4 * cap].view(cap, 2)[:N]
    return out, seen


def parsing_outputs(center_maps, params_maps, centermap_parser, return_batch_ids=False, guard=None, quiet=False):
    """post_parser.py:135-146 (+ the 1.1**scale of main.py:113, which the kernel applies to the
    sampled rows only).  center_maps (B,1,64,64); params_maps either the NHWC tensor
    (B,64,64,145) produced by RompNet.forward_nhwc or a (B,145,64,64) NCHW(-view) tensor.
    Returns the reference's dict (device tensors) or None when nobody is detected.
    `guard` (net.RangeGuard of the net that produced the maps): its saturation counter comes back with the detection count and
    the result has a third / second element: True when the caller must re-run this call in float32 (`quiet`: no 'None person
    detected' line for a result that is about to be replaced)."""
    if params_maps.dim() == 4 and params_maps.shape[1] == 145 and params_maps.shape[-1] != 145:
        params_maps = params_maps.permute(0, 2, 3, 1)          # free for RompNet's NCHW view
    r, seen = _parse(center_maps, params_maps, centermap_parser.conf_thresh, centermap_parser.max_person,
                     watch=guard.watch if guard is not None else None)
    if guard is not None:
        rerun = guard.check(seen, next_in_flight=getattr(guard, 'pipelined', False))
        if r is None:
            if not (rerun and quiet):
                print('None person detected')
            return (None, None, rerun) if return_batch_ids else (None, rerun)
    elif r is None:
        print('None person detected')
        return (None, None) if return_batch_ids else None
    thetas = r['smpl_thetas']
    res = {
        'cam': r['cam'], 'global_orient': thetas[:, :3].contiguous(), 'body_pose': thetas[:, 3:].contiguous(),
        'smpl_betas': r['smpl_betas'], 'smpl_thetas': thetas,
        'center_preds': r['center_preds'], 'center_confs': r['scores'].unsqueeze(1),
    }
    if guard is not None:
        return (res, r['batch_ids'], rerun) if return_batch_ids else (res, rerun)
    return (res, r['batch_ids']) if return_batch_ids else res


def rot6D_to_angular(rot6D):
    """utils.py:471-475 on device.  (N, J*6) -> (N, J*3)."""
    lib = L.load()
    x = rot6D.contiguous().float()
    n = x.numel() // 6
    out = torch.empty(n, 3, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        L.check(lib.romp_rot6d_to_aa(L.ptr(x), n, L.ptr(out), L.stream_ptr(x.device)))
    return out.reshape(rot6D.shape[0], -1)


def convert_cam_to_3d_trans(cams, weight=2.):
    """utils.py:303-307 (one kernel, csrc/parse.hip cam_to_trans_kernel)."""
    cams = cams.contiguous().float()
    out = torch.empty(cams.shape[0], 3, device=cams.device)
    with torch.cuda.device(cams.device):
        L.check(L.load().romp_cam_to_trans(L.ptr(cams), cams.shape[0], float(weight), L.ptr(out), L.stream_ptr(cams.device)))
    return out


try:
    import cv2 as _cv2  # noqa: F401
    _HAVE_CV2 = True
except Exception:
    _HAVE_CV2 = False


def estimate_translation_lsq(joints_3d, joints_2d, focal_length=443.4, img_size=(512., 512.)):
    """Camera translation by linear least squares -- the reference's estimate_translation without OpenCV (utils.py:391-434 ->
    estimate_translation_np :347-389): host statement of csrc/parse.hip translation_lsq_kernel.  A joint counts when its
    pixel row coordinate is > -2 (the reference's `joints_2d[:, :, -1] > -2.`); fewer than 4 -> (-1,-1,-1).
    joints_3d (N,K,3), joints_2d (N,K,2) numpy."""
    X = np.asarray(joints_3d, np.float64)
    uv = np.asarray(joints_2d, np.float64)
    N = X.shape[0]
    f = float(focal_length)
    c = np.asarray(img_size, np.float64) / 2.
    w = ((np.asarray(joints_2d)[:, :, -1] > -2.) & (np.asarray(joints_3d)[:, :, -1] != -2.)).astype(np.float64)
    qx, qy = (c[0] - uv[:, :, 0]) * w, (c[1] - uv[:, :, 1]) * w
    rx = ((uv[:, :, 0] - c[0]) * X[:, :, 2] - f * X[:, :, 0]) * w
    ry = ((uv[:, :, 1] - c[1]) * X[:, :, 2] - f * X[:, :, 1]) * w
    out = np.full((N, 3), -1.0, np.float32)
    for i in range(N):
        d = w[i].sum() * f * f
        A = np.array([[d, 0, f * qx[i].sum()], [0, d, f * qy[i].sum()], [f * qx[i].sum(), f * qy[i].sum(), (qx[i] ** 2 + qy[i] ** 2).sum()]])
        b = np.array([f * rx[i].sum(), f * ry[i].sum(), (qx[i] * rx[i] + qy[i] * ry[i]).sum()])
        if w[i].sum() >= 4:
            try:
                out[i] = np.linalg.solve(A, b)
            except np.linalg.LinAlgError:
                pass
    return out


def pnp_translation(j24, p24):
    """The reference's first-choice camera translation (post_parser.py:96-101 -> utils.py:391-434): cv2.solvePnPRansac per person on
    the host.  j24 (N,24,3), p24 (N,24,2) numpy in the 512-pixel input frame -> (N,3) float32, or None without OpenCV / on failure."""
    if not _HAVE_CV2:
        return None
    import cv2
    camK = np.eye(3)
    camK[0, 0] = camK[1, 1] = 443.4
    camK[:2, 2] = 256
    try:
        t = np.zeros((len(j24), 3), np.float32)
        for i in range(len(j24)):
            ret, rvec, tvec, inl = cv2.solvePnPRansac(j24[i], p24[i], camK, None, flags=cv2.SOLVEPNP_EPNP,
                                                      reprojectionError=20, iterationsCount=100)
            t[i] = -1 if inl is None else tvec[:, 0]
        return t
    except Exception:
        return None


def body_mesh_projection2image(j3d_preds, cam_preds, vertices=None, input2org_offsets=None, host_pnp=True):
    """post_parser.py:104-114.  pj2d / pj2d_org on device (csrc/parse.hip project_kernel);
    `cam_trans` follows the reference's PnP step (post_parser.py:96-101): cv2.solvePnPRansac on the host when OpenCV is
    installed (`pnp_translation`), else the reference's own least-squares fallback (utils.py:347-389) on the device
    (csrc/parse.hip translation_lsq_kernel; `estimate_translation_lsq` below is its host statement)."""
    lib = L.load()
    dev = j3d_preds.device
    j = j3d_preds.contiguous().float()
    cam = cam_preds.contiguous().float()
    N, J = j.shape[:2]
    pad = input2org_offsets if input2org_offsets is not None else torch.tensor([0., 512., 0., 512., 512., 512.])
    pad_c = (C.c_float * 6)(*[float(v) for v in pad])
    pj2d = torch.empty(N, J, 2, device=dev)
    pj2d_org = torch.empty(N, J, 2, device=dev)
    ct = torch.empty(N, 3, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.romp_project(L.ptr(j), N, J, L.ptr(cam), pad_c, L.ptr(pj2d), L.ptr(pj2d_org), L.ptr(ct),
                                 L.stream_ptr(dev)))
    trans = None
    if _HAVE_CV2 and host_pnp:                                         # the reference's first choice: OpenCV PnP on the host
        # (host_pnp=False: the caller runs pnp_translation itself on arrays it downloads anyway -- ROMP._forward_fast)
        t = pnp_translation(j[:, :24].detach().cpu().numpy(), (pj2d[:, :24].detach().cpu().numpy() + 1) * 256)   # post_parser.py:98
        trans = None if t is None else torch.from_numpy(t).float().to(dev)
    if trans is None:                                                  # its fallback, the linear least squares: on the device
        trans = torch.empty(N, 3, device=dev)
        with torch.cuda.device(dev):
            L.check(lib.romp_estimate_translation(L.ptr(j), N, J, 24, L.ptr(pj2d), 443.4, 512., L.ptr(trans), L.stream_ptr(dev)))
    out = {'pj2d': pj2d, 'cam_trans': trans}
    if input2org_offsets is not None:
        out['pj2d_org'] = pj2d_org
    if vertices is not None:                                           # post_parser.py:107-113 (dropped again before return unless rendering)
        v = vertices.contiguous().float()
        camed, org = torch.empty_like(v), torch.empty_like(v)
        with torch.cuda.device(dev):
            L.check(lib.romp_project_verts(L.ptr(v), v.shape[0], v.shape[1], L.ptr(cam), pad_c, L.ptr(camed), L.ptr(org), L.stream_ptr(dev)))
        out['verts_camed'] = camed
        if input2org_offsets is not None:
            out['verts_camed_org'] = org
    return out


class SMPL_parser(nn.Module):
    """post_parser.py:116-125."""

    def __init__(self, model_path):
        super(SMPL_parser, self).__init__()
        self.smpl_model = SMPL(model_path)

    def forward(self, outputs, root_align=False):
        verts, joints, face = self.smpl_model(outputs['smpl_betas'], outputs['smpl_thetas'], root_align=root_align)
        outputs.update({'verts': verts, 'joints': joints, 'smpl_face': face})
        return outputs
