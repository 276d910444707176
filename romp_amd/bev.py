"""BEV on the MI355X -- mirror of ``simple_romp/bev/main.py`` (``bev_settings`` :27-87, ``BEV`` :91-181)
and of the device-side steps of ``BEVv1.forward`` (bev/model.py:232-250) for BASELINE config 4.

    from romp_amd import bev
    model = bev.BEV(bev.bev_settings([]), state_dict=..., smpla_model=..., smil_model=...)
    outputs = model(bgr_image)         # dict of numpy arrays, or None

Network (HRNet-32 + BEV head), 3-D center parsing, per-person regression, SMPL-A / SMIL meshes,
perspective projection, projection-based duplicate suppression and outlier removal
(bev/post_parser.py:68-136,167-222) all run in libromp_hip.so.  Video mode (``-t``): ByteTrack-3D association on the host
(tracker.py) + OneEuro filters on the device (temporal.py); ``--render_mesh``: the Sim3DR rasteriser (renderer.py).
Not built: the long-image "crowd" sliding window (bev/main.py:184-258, CPU orchestration of repeated single forwards).
"""
import argparse
import ctypes as C
import os.path as osp
import sys

import numpy as np
import torch
from torch import nn

from . import lib as L
from .bev_plan import DEPTH, MAP, build_bev_hrnet32, cam3dmap_anchor
from .net import RangeGuard, RompNet
from .smpl import SMPL
from .utils import convert_tensor2numpy, determine_device, img_preprocess_device


def bev_settings(input_args=sys.argv[1:]):
    """bev/main.py:27-87 (model_id 2 defaults; rendering / crowd / temporal flags kept for
    compatibility, the options they enable are outside the MI355X hot path)."""
    p = argparse.ArgumentParser(description='BEV on MI355X')
    p.add_argument('-m', '--mode', type=str, default='image')
    p.add_argument('--model_id', type=int, default=2)
    p.add_argument('-i', '--input', type=str, default=None)
    p.add_argument('-o', '--save_path', type=str, default=osp.join(osp.expanduser('~'), 'BEV_results'))
    p.add_argument('--crowd', action='store_true', help='long-image sliding window (not built)')
    p.add_argument('--GPU', type=int, default=0)
    p.add_argument('--overlap_ratio', type=float, default=0.8)
    p.add_argument('--center_thresh', type=float, default=0.1)
    p.add_argument('--nms_thresh', type=float, default=20)
    p.add_argument('--relative_scale_thresh', type=float, default=1.6)
    p.add_argument('--show_largest', action='store_true')
    p.add_argument('--calc_smpl', action='store_false')
    p.add_argument('--render_mesh', action='store_true',
                   help='[romp_amd] off by default (the reference defaults to on with a bird view that needs its pyrender/cv2 overlays)')
    p.add_argument('--renderer', type=str, default='sim3dr')
    p.add_argument('--show_items', type=str, default='mesh', help="only 'mesh' is rendered on the device path")
    p.add_argument('-sc', '--smooth_coeff', type=float, default=3.)
    p.add_argument('--show', action='store_true')
    p.add_argument('--smpl_path', type=str, default=osp.join(osp.expanduser('~'), '.romp', 'SMPLA_NEUTRAL.pth'))
    p.add_argument('--smil_path', type=str, default=osp.join(osp.expanduser('~'), '.romp', 'smil_packed_info.pth'))
    p.add_argument('--model_path', type=str, default=osp.join(osp.expanduser('~'), '.romp', 'BEV.pth'))
    p.add_argument('-t', '--temporal_optimize', action='store_true')
    p.add_argument('--max_batch', type=int, default=32)
    p.add_argument('--conv_math', type=str, default='f16x2', choices=['f32', 'bf16x3', 'f16x2', 'all'],
                   help='[romp_amd] f32 MFMA only, or also the f32-accurate split-precision kernels (f16x2: 2 fp16 pieces; bf16x3; chosen by autotune)')
    args = p.parse_args(input_args)
    if not torch.cuda.is_available():
        args.GPU = -1
    return args


TAN_FOV = float(np.tan(np.radians(60 / 2.)))


def denormalize_cam_params_to_trans(normed_cams):
    """bev/post_parser.py:114-128 (positive_constrain False): (scale, ty, tx) -> camera-space translation."""
    depth = (1. / (normed_cams[:, 0] * TAN_FOV + 1e-3)).unsqueeze(1)
    return torch.cat([torch.flip(normed_cams[:, 1:], [1]) * depth * TAN_FOV, depth], 1)


class CenterMap3D(object):
    """bev/post_parser.py:19-66 -- parse configuration + the device parse."""

    def __init__(self, conf_thresh):
        self.size, self.max_person, self.conf_thresh = 128, 64, conf_thresh

    def parse_3dcentermap(self, center_maps):
        """center_maps (B,64,128,128) device tensor -> [batch_ids (N), center_zyxs (N,3), scores (N)]
        (int64 / int64 / float32), batch-major and score-descending; empty tensors if nobody."""
        lib = L.load()
        dev = center_maps.device
        if dev.type != 'cuda':
            raise L.RompHipError('3-D center parsing runs on the HIP device only (no CPU fallback)')
        cm = center_maps.contiguous().float()
        B = cm.shape[0]
        cap = B * self.max_person
        bids = torch.empty(cap, device=dev, dtype=torch.int32)
        czyx = torch.empty(cap, 3, device=dev, dtype=torch.int32)
        conf = torch.empty(cap, device=dev, dtype=torch.float32)
        ws = torch.empty(lib.romp_bev_workspace_ints(B, self.max_person), device=dev, dtype=torch.int32)
        n = C.c_int32(0)
        with torch.cuda.device(dev):
            L.check(lib.romp_bev_parse(L.ptr(cm), B, float(self.conf_thresh), self.max_person, C.byref(n), L.ptr(bids),
                                       L.ptr(czyx), L.ptr(conf), L.ptr(ws), L.stream_ptr(dev)))
        N = n.value
        return [bids[:N].long(), czyx[:N].long(), conf[:N]]


class BEVv1(object):
    """Device-side BEVv1 (bev/model.py:104-250): network program + parse + per-person regression."""

    def __init__(self, state_dict, device, center_thresh=0.1, max_batch=32, bf16x3=False):
        self.device = torch.device(device)
        self.net = RompNet(state_dict, self.device, max_batch=max_batch, builder=build_bev_hrnet32,
                           out_shapes=((DEPTH, MAP, MAP), (3, DEPTH, MAP, MAP)), bf16x3=bf16x3)
        self.centermap_parser = CenterMap3D(center_thresh)
        self.range_guard = RangeGuard(self.net)                # default-on, as in ROMP (net.RangeGuard)
        f = lambda k: state_dict[k].detach().float()
        dv = lambda t: t.contiguous().to(self.device)
        self.emb = dv(f('position_embeddings.weight'))
        self.w1t, self.b1 = dv(f('transformer.0.weight').t()), dv(f('transformer.0.bias'))      # [in][out]
        self.w2t, self.b2 = dv(f('transformer.3.weight').t()), dv(f('transformer.3.bias'))
        self.w3t, self.b3 = dv(f('transformer.6.weight').t()), dv(f('transformer.6.bias'))
        self.anchors = (C.c_float * DEPTH)(*cam3dmap_anchor(60, MAP).tolist())

    def localization(self, images):
        """coarse2fine_localization: -> center_maps_3d (B,64,128,128), cam_maps_3d (B,3,64,128,128)."""
        return self.net.forward_nhwc(images)

    @torch.no_grad()
    def __call__(self, images):
        """images (B,512,512,3) float on device -> dict like BEVv1.forward (:247-249) or None."""
        lib = L.load()
        net = self.net
        c3d, cam3d = self.localization(images)
        bids, czyx, confs = self.centermap_parser.parse_3dcentermap(c3d)
        # range guard: the parse has just synchronised on the count, so the network is done -- one 4-byte read of its saturation
        # counter; if it moved, the whole call again on the exact-f32 program (whose front-view features the regression then reads)
        if self.range_guard.enabled and self.range_guard.check(net.saturated):
            net = self.net.f32_twin()
            c3d, cam3d = net.forward_nhwc(images)
            self.range_guard.warn(images)
            bids, czyx, confs = self.centermap_parser.parse_3dcentermap(c3d)
        N = bids.shape[0]
        if N == 0:
            print('No person detected!')
            return None
        dev = self.device
        f32 = dict(device=dev, dtype=torch.float32)
        out = {'params_pred': torch.empty(N, 146, **f32), 'cam': torch.empty(N, 3, **f32),
               'smpl_thetas': torch.empty(N, 72, **f32), 'smpl_betas': torch.empty(N, 11, **f32),
               'cam_trans': torch.empty(N, 3, **f32)}
        cam_czyx = torch.empty(N, 3, device=dev, dtype=torch.int32)
        b32, z32 = bids.int().contiguous(), czyx.int().contiguous()
        P = net.program
        feat_ptr = net.buffer_ptr(P.fv_buf) + 4 * P.fv_coff
        with torch.cuda.device(dev):
            L.check(lib.romp_bev_regress(L.ptr(cam3d), C.c_void_p(feat_ptr), P.fv_cstride, N, L.ptr(b32), L.ptr(z32),
                                         self.anchors, L.ptr(self.emb), L.ptr(self.w1t), L.ptr(self.b1), L.ptr(self.w2t),
                                         L.ptr(self.b2), L.ptr(self.w3t), L.ptr(self.b3), L.ptr(out['params_pred']),
                                         L.ptr(cam_czyx), L.ptr(out['cam']), L.ptr(out['smpl_thetas']),
                                         L.ptr(out['smpl_betas']), L.ptr(out['cam_trans']), L.stream_ptr(dev)))
        out.update({'cam_czyx': cam_czyx.float(), 'center_map_3d': c3d, 'cam_maps_3d': cam3d, 'pred_batch_ids': bids,
                    'pred_czyxs': czyx, 'center_confs': confs})
        return out


class SMPLA_parser(nn.Module):
    """bev/post_parser.py:255-278: SMPL-A for adults, SMIL for betas[:,10] > 0.8, always root-aligned."""

    def __init__(self, smpla_path, smil_path):
        super(SMPLA_parser, self).__init__()
        self.smil_model = SMPL(smil_path, model_type='smpl')
        self.smpl_model = SMPL(smpla_path, model_type='smpla')
        self.baby_thresh = 0.8

    def forward(self, betas=None, thetas=None, root_align=True):
        baby_mask = betas[:, 10] > self.baby_thresh
        if baby_mask.sum() > 0:
            adult_mask = ~baby_mask
            n = len(thetas)
            verts = torch.zeros(n, 6890, 3, device=thetas.device)
            joints = torch.zeros(n, 71, 3, device=thetas.device)
            verts[baby_mask], joints[baby_mask], face = self.smil_model(betas[baby_mask, :10].contiguous(),
                                                                        thetas[baby_mask].contiguous(), root_align=root_align)
            if adult_mask.sum() > 0:
                verts[adult_mask], joints[adult_mask], face = self.smpl_model(betas[adult_mask].contiguous(),
                                                                              thetas[adult_mask].contiguous(),
                                                                              root_align=root_align)
        else:
            verts, joints, face = self.smpl_model(betas, thetas, root_align=root_align)
        return verts, joints, face


class BEV(nn.Module):
    def __init__(self, settings, state_dict=None, smpla_model=None, smil_model=None):
        super(BEV, self).__init__()
        self.settings = settings
        if settings.GPU == -1:
            raise L.RompHipError('romp_amd.bev needs a HIP device; there is no CPU fallback')
        if settings.crowd:
            raise NotImplementedError('crowd mode (long-image sliding window, bev/main.py:184-258) is host orchestration outside the MI355X hot path')
        self.tdevice = determine_device(settings.GPU)
        if state_dict is None:
            state_dict = torch.load(settings.model_path, map_location='cpu')
        self.model = BEVv1(state_dict, self.tdevice, center_thresh=settings.center_thresh,
                           max_batch=getattr(settings, 'max_batch', 32),
                           bf16x3=getattr(settings, 'conv_math', 'f16x2'))
        if settings.calc_smpl:
            self.smpl_parser = SMPLA_parser(smpla_model if smpla_model is not None else settings.smpl_path,
                                            smil_model if smil_model is not None else settings.smil_path).to(self.tdevice)
        self.result_keys = ['smpl_thetas', 'smpl_betas', 'cam', 'cam_trans', 'params_pred', 'center_confs', 'pred_batch_ids']
        if settings.temporal_optimize:                                                      # bev/main.py:117-121
            self.OE_filters = {}
            if not settings.show_largest:
                from .tracker import Tracker
                self.tracker = Tracker(det_thresh=0.12, low_conf_det_thresh=0.05, track_buffer=60, match_thresh=300, frame_rate=30)
        if settings.render_mesh:                                                            # bev/main.py:112-114
            from .vis import setup_renderer
            self.renderer = setup_renderer(name=getattr(settings, 'renderer', 'sim3dr'), device=self.tdevice)
            self.visualize_items = getattr(settings, 'show_items', 'mesh').split(',')

    def temporal_optimization(self, outputs, signal_ID, image_scale=128, depth_scale=30):
        """bev/main.py:260-287: ByteTrack-3D association on the host (tracker.py), OneEuro filters on the device (temporal.py).
        Returns None when no confirmed track is seen in the frame."""
        from .temporal import OneEuroBank
        if signal_ID not in self.OE_filters:                                                # check_filter_state (utils.py:246-255)
            if len(self.OE_filters) > 100:
                self.OE_filters.clear()
            self.OE_filters[signal_ID] = OneEuroBank(self.tdevice, getattr(self.settings, 'smooth_coeff', 3.), outputs['smpl_betas'].shape[1])
        bank = self.OE_filters[signal_ID]
        if self.settings.show_largest:
            max_id = int(torch.argmax(outputs['cam'][:, 0]))
            th, be, ca = (outputs[k][max_id:max_id + 1].contiguous().clone() for k in ('smpl_thetas', 'smpl_betas', 'cam'))
            outputs['smpl_thetas'], outputs['smpl_betas'], outputs['cam'] = bank.smooth([0], th, be, ca)
            return outputs
        cams = outputs['cam'].cpu().numpy()
        cam_trans = outputs['cam_trans'].cpu().numpy()
        det_confs = outputs['center_confs'].cpu().numpy()
        tracking_points = np.concatenate([(cams[:, [2, 1]] + 1) * image_scale, cam_trans[:, [2]] * depth_scale,
                                          cams[:, [0]] * image_scale / 2], 1)
        tracked_ids, results_inds = self.tracker.update(tracking_points, det_confs)
        if len(tracked_ids) == 0:
            return None
        rows = torch.as_tensor(results_inds, dtype=torch.long, device=self.tdevice)
        for key in self.result_keys:
            outputs[key] = outputs[key][rows].contiguous()
        outputs['smpl_thetas'], outputs['smpl_betas'], outputs['cam'] = bank.smooth(tracked_ids, outputs['smpl_thetas'],
                                                                                    outputs['smpl_betas'], outputs['cam'])
        outputs['track_ids'] = np.array(tracked_ids).astype(np.int32)
        return outputs

    @torch.no_grad()
    def forward_batch(self, images, pad_infos=None):
        """[extension] images (B,512,512,3) float on device -> dict of device tensors or None.
        With `pad_infos` (B,6) the per-image post-processing of process_normal_image (bev/main.py:168-180)
        runs too: projection, duplicate suppression, outlier removal (rows of dropped persons removed)."""
        out = self.model(images)
        if out is None:
            return None
        res = {k: out[k] for k in self.result_keys}
        if self.settings.calc_smpl:
            verts, joints, face = self.smpl_parser(res['smpl_betas'], res['smpl_thetas'])
            res.update({'verts': verts, 'joints': joints})
            if pad_infos is not None:
                res = self._postprocess(res, images.shape[0], pad_infos)
        return res

    def _postprocess(self, res, B, pad_infos):
        lib = L.load()
        dev = self.tdevice
        N = res['cam'].shape[0]
        counts = torch.bincount(res['pred_batch_ids'], minlength=B)
        offsets = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        offsets[1:] = torch.cumsum(counts, 0).int()
        pads = torch.as_tensor(pad_infos, dtype=torch.float32).reshape(B, 6).to(dev).contiguous()
        pj = torch.empty(N, 71, 2, device=dev)
        pjo = torch.empty(N, 71, 2, device=dev)
        tr = torch.empty(N, 3, device=dev)
        keep = torch.empty(N, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            L.check(lib.romp_bev_postprocess(L.ptr(res['joints'].contiguous()), L.ptr(res['cam'].contiguous()), L.ptr(offsets),
                                             B, L.ptr(pads), float(self.settings.nms_thresh),
                                             float(self.settings.relative_scale_thresh), L.ptr(pj), L.ptr(pjo), L.ptr(tr),
                                             L.ptr(keep), L.stream_ptr(dev)))
        res.update({'cam_trans': tr, 'pj2d': pjo, 'pj2d_org': pjo})       # the reference aliases pj2d to pj2d_org
        mask = keep.bool()
        mask_np = None
        out = {}
        for k, v in res.items():                        # remove_subjects (bev/post_parser.py:154-165): every per-person entry
            if torch.is_tensor(v) and v.shape[:1] == (N,) and k != 'smpl_face':
                v = v[mask]
            elif isinstance(v, np.ndarray) and v.shape[:1] == (N,):
                mask_np = mask.cpu().numpy() if mask_np is None else mask_np
                v = v[mask_np]
            out[k] = v
        return out

    def forward(self, image, signal_ID=0, **kwargs):
        """bev/main.py:139-181 (normal images): BGR uint8 HxWx3 -> dict of numpy arrays or None."""
        input_image, image_pad_info = img_preprocess_device(image, self.tdevice)
        if not (self.settings.temporal_optimize or self.settings.render_mesh):
            res = self.forward_batch(input_image, image_pad_info.reshape(1, 6))
            return None if res is None else convert_tensor2numpy(res)
        out = self.model(input_image)
        if out is None:
            return None
        res = {k: out[k] for k in self.result_keys}
        if self.settings.temporal_optimize:                                                 # bev/main.py:162-166
            res = self.temporal_optimization(res, signal_ID)
            if res is None:
                return None
            res['cam_trans'] = denormalize_cam_params_to_trans(res['cam'])
        if self.settings.calc_smpl:
            verts, joints, face = self.smpl_parser(res['smpl_betas'], res['smpl_thetas'])
            res.update({'verts': verts, 'joints': joints, 'smpl_face': face})
            res = self._postprocess(res, 1, image_pad_info.reshape(1, 6))
            if self.settings.render_mesh:                                                   # bev/main.py:147-150
                res['verts_camed_org'] = self._verts_camed_org(res['verts'], res['cam_trans'], image_pad_info)
                from .vis import rendering_romp_bev_results
                cfgs = {'mesh_color': 'identity', 'items': self.visualize_items, 'renderer': getattr(self.settings, 'renderer', 'sim3dr')}
                res = rendering_romp_bev_results(self.renderer, res, image, cfgs)
        return convert_tensor2numpy(res)

    def _verts_camed_org(self, verts, cam_trans, pad_info):
        lib = L.load()
        v = verts.contiguous().float()
        org = torch.empty_like(v)
        pad_c = (C.c_float * 6)(*[float(x) for x in pad_info])
        with torch.cuda.device(self.tdevice):
            L.check(lib.romp_bev_project_verts(L.ptr(v), v.shape[0], v.shape[1], L.ptr(cam_trans.contiguous().float()), pad_c,
                                               L.ptr(org), L.stream_ptr(self.tdevice)))
        return org
