"""Drop-in for ``simple_romp/romp/main.py``: ``romp_settings`` (:17-60), ``ROMP`` (:64-176), ``main`` (:178-204).

    import romp_amd as romp
    model = romp.ROMP(romp.main.default_settings)
    outputs = model(cv2.imread(path))          # dict of numpy arrays, or None

Same settings Namespace, same dict keys/dtypes/shapes, ``None`` when nobody is detected.
Every arithmetic stage (network, parsing, SMPL, projection) runs in libromp_hip.so on the
MI355X; there is NO CPU path here -- ``--GPU -1`` (or a missing HIP device/extension) raises.
``forward_batch`` is an extension of the API for batched throughput (the reference has none).
"""
import argparse
import os
import os.path as osp
import sys

import numpy as np
import torch
from torch import nn

from . import lib as L
from .net import RangeGuard, RompNet
from .post_parser import (_HAVE_CV2, CenterMap, SMPL_parser, body_mesh_projection2image, convert_cam_to_3d_trans, pnp_translation,
                          parsing_outputs)
from .vis import rendering_romp_bev_results, setup_renderer
from .utils import ResultSaver, convert_tensor2numpy, determine_device, img_preprocess, img_preprocess_device


def romp_settings(input_args=sys.argv[1:]):
    """main.py:17-60 -- identical flags and defaults (auto-download is dropped: no network)."""
    parser = argparse.ArgumentParser(description='ROMP: Monocular, One-stage, Regression of Multiple 3D People')
    parser.add_argument('-m', '--mode', type=str, default='image', help='Inferece mode, including image, video, webcam')
    parser.add_argument('-i', '--input', type=str, default=None, help='Path to the input image / video')
    parser.add_argument('-o', '--save_path', type=str, default=osp.join(osp.expanduser("~"), 'ROMP_results'), help='Path to save the results')
    parser.add_argument('--GPU', type=int, default=0, help='The gpu device number to run the inference on.')
    parser.add_argument('--onnx', action='store_true', help='(reference flag; the HIP path replaces the ONNX session)')
    parser.add_argument('-t', '--temporal_optimize', action='store_true', help='Whether to use OneEuro filter to smooth the results')
    parser.add_argument('--center_thresh', type=float, default=0.25, help='The confidence threshold of positive detection in 2D human body center heatmap.')
    parser.add_argument('--show_largest', action='store_true', help='Whether to show the largest person only')
    parser.add_argument('-sc', '--smooth_coeff', type=float, default=3., help='The smoothness coeff of OneEuro filter, the smaller, the smoother.')
    parser.add_argument('--calc_smpl', action='store_false', help='Whether to calculate the smpl mesh from estimated SMPL parameters')
    parser.add_argument('--render_mesh', action='store_true', help='Whether to render the estimated 3D mesh mesh to image')
    parser.add_argument('--renderer', type=str, default='sim3dr', help='Choose the renderer for visualizaiton')
    parser.add_argument('--show', action='store_true', help='Whether to show the rendered results')
    parser.add_argument('--show_items', type=str, default='mesh', help='The items to visualized')
    parser.add_argument('--save_video', action='store_true', help='Whether to save the video results')
    parser.add_argument('--frame_rate', type=int, default=24, help='The frame_rate of saved video results')
    parser.add_argument('--smpl_path', type=str, default=osp.join(osp.expanduser("~"), '.romp', 'SMPL_NEUTRAL.pth'), help='The path of smpl model file')
    parser.add_argument('--model_path', type=str, default=osp.join(osp.expanduser("~"), '.romp', 'ROMP.pkl'), help='The path of ROMP checkpoint')
    parser.add_argument('--model_onnx_path', type=str, default=osp.join(osp.expanduser("~"), '.romp', 'ROMP.onnx'), help='The path of ROMP onnx checkpoint')
    parser.add_argument('--root_align', type=bool, default=False, help='Please set this config as True to use the ROMP checkpoints trained by yourself.')
    parser.add_argument('--webcam_id', type=int, default=0, help='The Webcam ID.')
    parser.add_argument('--max_batch', type=int, default=32, help='[romp_amd] largest batch forward_batch will be called with')
    parser.add_argument('--conv_math', type=str, default='f16x2', choices=['f32', 'bf16x3', 'f16x2', 'all'],
                        help='[romp_amd] f32: exact-f32 MFMA kernels only; f16x2 (default) / bf16x3 (needs a library built with ROMP_WITH_BX3=1): also offer the f32-accurate split-precision kernels '
                             '(2 fp16 pieces, 3 products / 3 bf16 pieces, 6 products) on the 16-bit matrix pipe, chosen per layer by measurement the first '
                             'time a batch size is seen; all: both families')
    parser.add_argument('--backbone', type=str, default='hrnet32', choices=['hrnet32', 'resnet50'],
                        help='[romp_amd] hrnet32: the simple_romp model (ROMP.pkl); resnet50: the training tree\'s ResNet-50 variant '
                             '(romp/lib/models/resnet_50.py + romp_model.py state_dict)')
    parser.add_argument('--plan_path', type=str, default=None,
                        help='[romp_amd] start from a plan file (python -m romp_amd.export: the lowered network with its packed constants '
                             'and measured kernel tables) instead of --model_path: the counterpart of the reference\'s --onnx / --model_onnx_path')
    parser.add_argument('--no_calibrate', action='store_true',
                        help='[romp_amd] skip the start-up range calibration of the f16x2 kernels (one float32 forward; tensors whose range does '
                             'not fit fp16 pieces stay float32): ranges are then trusted, out-of-range values saturate (RompNet.saturated counts them)')
    parser.add_argument('--calib_dir', type=str, default=None,
                        help='[romp_amd] calibrate on up to 4 images of this directory (pre-processed like inputs) instead of synthetic frames')
    parser.add_argument('--host_preprocess', action='store_true', help='[romp_amd] pad/resize on the host (cv2 / numpy) instead of the device kernel')
    args = parser.parse_args(input_args)
    if not torch.cuda.is_available():
        args.GPU = -1
        args.temporal_optimize = False
    if args.show:
        args.render_mesh = True
    if args.render_mesh or args.show_largest:
        args.calc_smpl = True
    if not os.path.exists(args.smpl_path):
        alt = args.smpl_path.replace('SMPL_NEUTRAL.pth', 'smpl_packed_info.pth')
        if os.path.exists(alt):
            args.smpl_path = alt
    return args


default_settings = romp_settings(input_args=[])


class ROMP(nn.Module):
    def __init__(self, romp_settings, state_dict=None, smpl_model=None):
        """`state_dict` / `smpl_model` optionally supply already-loaded weights (dicts with the
        reference's schemas) instead of settings.model_path / settings.smpl_path."""
        super(ROMP, self).__init__()
        self.settings = romp_settings
        if self.settings.GPU == -1:
            raise L.RompHipError('romp_amd is the MI355X path of ROMP: it needs a HIP device (GPU=%d); '
                                 'there is no CPU fallback' % self.settings.GPU)
        self.tdevice = determine_device(self.settings.GPU)
        self._build_model_(state_dict)
        self.range_guard = RangeGuard(self.model)          # default-on: no call returns clamped f16x2 maps (net.RangeGuard)
        self._initilization_(smpl_model)

    def _build_model_(self, state_dict=None):
        """main.py:72-77: load the state_dict and bind it into the HIP network context (or, like the reference's ONNX branch
        main.py:86-89, start from the exported plan file)."""
        if state_dict is None and getattr(self.settings, 'plan_path', None):
            self.model = RompNet.from_plan(self.settings.plan_path, self.tdevice, max_batch=getattr(self.settings, 'max_batch', 32))
            return
        if state_dict is None:
            state_dict = torch.load(self.settings.model_path, map_location='cpu')
        builder = None
        if getattr(self.settings, 'backbone', 'hrnet32') == 'resnet50':
            from .resnet_plan import build_romp_resnet50 as builder
        self.model = RompNet(state_dict, self.tdevice, max_batch=getattr(self.settings, 'max_batch', 32), builder=builder,
                             bf16x3=getattr(self.settings, 'conv_math', 'f16x2'),
                             calibrate=False if getattr(self.settings, 'no_calibrate', False) else None,
                             calib_images=self._calibration_frames())

    def _calibration_frames(self):
        """--calib_dir: up to 4 real frames, pre-processed exactly like inputs, for RompNet's range calibration (the
        synthetic default frames are not photos).  None: the synthetic set."""
        d = getattr(self.settings, 'calib_dir', None)
        if not d:
            return None
        paths = sorted(osp.join(d, f) for f in os.listdir(d) if f.lower().endswith(('.jpg', '.jpeg', '.png', '.bmp'))) if osp.isdir(d) else [d]
        frames = []
        for path in paths[:4]:
            img = _imread_bgr(path)
            if img is not None:
                frames.append(img_preprocess_device(img, self.tdevice)[0])
        if not frames:
            raise L.RompHipError('--calib_dir %s holds no readable image' % d)
        return torch.cat(frames, 0)

    def _initilization_(self, smpl_model=None):
        self.centermap_parser = CenterMap(conf_thresh=self.settings.center_thresh)
        if self.settings.calc_smpl:
            self.smpl_parser = SMPL_parser(smpl_model if smpl_model is not None else self.settings.smpl_path).to(self.tdevice)
        if self.settings.temporal_optimize:                                                 # main.py:98-99, :105-116
            self.OE_filters = {}
            if not self.settings.show_largest:
                try:
                    from norfair import Tracker
                except ImportError:
                    raise NotImplementedError('multi-person temporal smoothing associates persons with the third-party norfair '
                                              'tracker (main.py:108-116), which is not installed; use --show_largest')
                from .utils import euclidean_distance
                self.tracker = Tracker(distance_function=euclidean_distance, distance_threshold=200)
                self.tracker_initialized = False
        if self.settings.render_mesh:                                                       # main.py:101-103
            self.visualize_items = self.settings.show_items.split(',')
            self.renderer = setup_renderer(name=self.settings.renderer, device=self.tdevice)

    def single_image_forward(self, image):
        """main.py:106-115."""
        if getattr(self.settings, 'host_preprocess', False):
            input_image, image_pad_info = img_preprocess(image)            # reference path (cv2 / numpy on the host)
            input_image = input_image.to(self.tdevice)
        else:                                                              # uint8 upload + pad/resize on the device
            input_image, image_pad_info = img_preprocess_device(image, self.tdevice)
        center_maps, params_maps = self.model(input_image)
        parsed_results, rerun = parsing_outputs(center_maps, params_maps, self.centermap_parser, guard=self.range_guard, quiet=True)
        if rerun:                                                          # a value left the f16x2 range: the float32 program's answer
            center_maps, params_maps = self._rerun_f32(input_image)
            parsed_results = parsing_outputs(center_maps, params_maps, self.centermap_parser)
        return parsed_results, image_pad_info

    def _rerun_f32(self, images):
        """The call again on the exact-f32 program of the same weights (RompNet.f32_twin) after the range guard tripped; warns
        once, naming the ops that clamped.  -> (center_maps (B,1,64,64), params_maps (B,64,64,145) NHWC)."""
        net32 = self.model.f32_twin()
        center, params = net32.forward_nhwc(images)
        self.range_guard.warn(images)
        return center.unsqueeze(1), params

    def temporal_optimization(self, outputs, signal_ID):
        """main.py:117-157: OneEuro smoothing of thetas / betas / cam, per tracked person (or of the largest
        person with --show_largest).  Filters on the device (temporal.py / csrc/temporal.hip)."""
        from .temporal import OneEuroBank
        if signal_ID not in self.OE_filters:                                                # check_filter_state (utils.py:246-255)
            if len(self.OE_filters) > 100:
                self.OE_filters.clear()
            self.OE_filters[signal_ID] = OneEuroBank(self.tdevice, self.settings.smooth_coeff, outputs['smpl_betas'].shape[1])
        bank = self.OE_filters[signal_ID]
        if self.settings.show_largest:
            max_id = int(torch.argmax(outputs['cam'][:, 0]))
            th, be, ca = (outputs[k][max_id:max_id + 1].contiguous().clone() for k in ('smpl_thetas', 'smpl_betas', 'cam'))
            outputs['smpl_thetas'], outputs['smpl_betas'], outputs['cam'] = bank.smooth([0], th, be, ca)
            return outputs
        import numpy as np
        from norfair import Detection
        from .utils import get_tracked_ids
        detections = [Detection(points=cam[[2, 1]] * 512) for cam in outputs['cam'].cpu().numpy()]
        if not self.tracker_initialized:                  # the reference never sets the flag (main.py:141-143): the eight warm-up
            for _ in range(8):                            # updates run on EVERY frame; kept, the track ids depend on it
                self.tracker.update(detections=detections)
        tracked_objects = self.tracker.update(detections=detections)
        if len(tracked_objects) == 0:
            return outputs
        tracked_ids = get_tracked_ids(detections, tracked_objects)
        th, be, ca = (outputs[k].contiguous() for k in ('smpl_thetas', 'smpl_betas', 'cam'))
        outputs['smpl_thetas'], outputs['smpl_betas'], outputs['cam'] = bank.smooth(tracked_ids, th, be, ca)
        outputs['track_ids'] = np.array(tracked_ids).astype(np.int32)
        return outputs

    def _finish(self, outputs, image_pad_info):
        outputs['cam_trans'] = convert_cam_to_3d_trans(outputs['cam'])                      # main.py:166
        if self.settings.calc_smpl:
            outputs = self.smpl_parser(outputs, root_align=self.settings.root_align)        # main.py:168
            outputs.update(body_mesh_projection2image(outputs['joints'], outputs['cam'],
                                                      vertices=outputs['verts'] if self.settings.render_mesh else None,
                                                      input2org_offsets=image_pad_info))   # main.py:169
        return outputs

    def _forward_fast(self, image):
        """forward() for the plain case (meshes, no smoothing, no rendering), arranged for latency: the reference's flow has two
        host round trips per frame (the detection count after the parse, the results at the end) with every launch of the
        post-processing issued between them while the GPU idles.  Here everything is enqueued for all `max_person` candidate
        rows while the network is still running -- parse without its count read-back, SMPL, projection, camera translation --
        then ONE small download (count + per-person rows) and one for the N meshes.  Same kernels on the same inputs as the
        standard path: the same bytes (tests/test_gpu_parity.py::test_romp_api_fast_path).  The saturation counter of the network
        rides in that same download (romp_parse_watch); if it moved, the frame is done again from the float32 program's maps."""
        x, pad = img_preprocess_device(image, self.tdevice)
        center, params = self.model.forward_nhwc(x)
        result, seen = self._fast_from_maps(center, params, pad, self.range_guard.watch)
        if self.range_guard.check(seen):
            c32, p32 = self._rerun_f32(x)
            result, _ = self._fast_from_maps(c32.squeeze(1), p32, pad, None)
        if result is None:
            print('None person detected')
        return result

    def _fast_from_maps(self, center, params, pad, watch):
        """The latency-arranged post-processing of one frame's maps -> (result dict or None, the watched word or None)."""
        import ctypes as C
        import numpy as np
        from . import lib as L
        lib, dev, cap = L.load(), self.tdevice, self.centermap_parser.max_person
        st = getattr(self, '_fast', None)
        if st is None:
            st = self._fast = dict(ibuf=torch.zeros(cap * 4 + 2 * cap + 2 + 2, device=dev, dtype=torch.int32),
                                   fbuf=torch.zeros(cap * (1 + 145 + 3 + 72 + 10), device=dev, dtype=torch.float32))
        ibuf, fbuf = st['ibuf'], st['fbuf']
        views, at = {}, 0
        for key, w in (('scores', 1), ('params_pred', 145), ('cam', 3), ('smpl_thetas', 72), ('smpl_betas', 10)):
            views[key] = fbuf[at:at + cap * w].view(cap, w)
            at += cap * w
        with torch.cuda.device(dev):
            L.check(lib.romp_parse_watch(L.ptr(center), L.ptr(params), 1, float(self.centermap_parser.conf_thresh), cap, None,
                                         L.ptr(ibuf), L.ptr(ibuf[cap:]), L.ptr(views['scores']), L.ptr(views['params_pred']), L.ptr(views['cam']),
                                         L.ptr(views['smpl_thetas']), L.ptr(views['smpl_betas']), L.ptr(ibuf[2 * cap:]), L.ptr(ibuf[4 * cap:]),
                                         L.stream_ptr(dev), C.c_void_p(watch or 0), None))
        verts, joints, _ = self.smpl_parser.smpl_model(views['smpl_betas'], views['smpl_thetas'], root_align=self.settings.root_align)
        proj = body_mesh_projection2image(joints, views['cam'], input2org_offsets=pad, host_pnp=False)
        small = torch.cat([ibuf.view(torch.float32), fbuf, proj['cam_trans'].reshape(-1), joints.reshape(-1), proj['pj2d_org'].reshape(-1),
                           proj['pj2d'].reshape(-1)])
        host = small.cpu().numpy()                                   # the one synchronisation point of the frame
        hi = host[:ibuf.numel()].view(np.int32)
        N = int(hi[4 * cap + 2 * cap])
        seen = int(hi[4 * cap + 2 * cap + 2]) if watch else None     # workspace[B * (2 * cap + 2)] with B = 1: the watched word
        if N == 0:
            return None, seen
        hf = host[ibuf.numel():]
        out, at = {}, 0
        for key, w in (('scores', 1), ('params_pred', 145), ('cam', 3), ('smpl_thetas', 72), ('smpl_betas', 10), ('cam_trans', 3),
                       ('joints', 213), ('pj2d_org', 142), ('pj2d', 142)):
            out[key] = hf[at:at + cap * w].reshape(cap, w)[:N]
            at += cap * w
        if _HAVE_CV2:                                                # OpenCV installed: the reference's PnP translation, on the rows just downloaded
            t = pnp_translation(out['joints'].reshape(N, 71, 3)[:, :24], (out['pj2d'].reshape(N, 71, 2)[:, :24] + 1) * 256)
            if t is not None:
                out['cam_trans'] = t
        th = out['smpl_thetas']
        return {'cam': out['cam'], 'global_orient': np.ascontiguousarray(th[:, :3]), 'body_pose': np.ascontiguousarray(th[:, 3:]),
                'smpl_betas': out['smpl_betas'], 'smpl_thetas': th, 'center_preds': hi[2 * cap:4 * cap].reshape(cap, 2)[:N].astype(np.int64),
                'center_confs': out['scores'].reshape(N, 1), 'cam_trans': out['cam_trans'], 'verts': verts[:N].cpu().numpy(),
                'joints': out['joints'].reshape(N, 71, 3), 'pj2d_org': out['pj2d_org'].reshape(N, 71, 2)}, seen

    def forward(self, image, signal_ID=0, **kwargs):
        """main.py:160-176: BGR uint8 HxWx3 numpy -> dict of numpy arrays, or None."""
        s = self.settings
        if (getattr(self, 'fast_single', True) and s.calc_smpl and not s.temporal_optimize and not s.render_mesh
                and not getattr(s, 'host_preprocess', False)):
            return self._forward_fast(image)
        outputs, image_pad_info = self.single_image_forward(image)
        if outputs is None:
            return None
        if self.settings.temporal_optimize:                                                 # main.py:164-165
            outputs = self.temporal_optimization(outputs, signal_ID)
        outputs = self._finish(outputs, image_pad_info)
        if self.settings.render_mesh:                                                       # main.py:170-172
            rendering_cfgs = {'mesh_color': 'identity', 'items': self.visualize_items, 'renderer': self.settings.renderer}
            outputs = rendering_romp_bev_results(self.renderer, outputs, image, rendering_cfgs)
        return convert_tensor2numpy(outputs)

    @torch.no_grad()
    def forward_batch(self, images, return_tensors=True):
        """[extension] images: float32 (B,512,512,3) 0..255 already pre-processed, on the device.
        Runs net -> parse -> SMPL for the whole batch; returns (outputs dict, batch_ids) with device
        tensors (verts (N,6890,3), joints (N,71,3), cam, smpl_thetas, smpl_betas, ...), or
        (None, None) if nobody is detected."""
        center, params = self.model.forward_nhwc(images)
        outputs, batch_ids, rerun = parsing_outputs(center.unsqueeze(1), params, self.centermap_parser, return_batch_ids=True,
                                                    guard=self.range_guard, quiet=True)
        if rerun:                                                          # (the range guard: this call again on the float32 program)
            c32, p32 = self._rerun_f32(images)
            outputs, batch_ids = parsing_outputs(c32, p32, self.centermap_parser, return_batch_ids=True)
        if outputs is None:
            return None, None
        outputs['cam_trans'] = convert_cam_to_3d_trans(outputs['cam'])
        if self.settings.calc_smpl:
            outputs = self.smpl_parser(outputs, root_align=self.settings.root_align)
        if not return_tensors:
            outputs = convert_tensor2numpy(outputs)
        return outputs, batch_ids


    @torch.no_grad()
    def forward_chunks(self, images, chunk, next_images=None):
        """[extension] forward_batch over `images` (n,512,512,3) in chunks of `chunk` images, software-pipelined: the network of
        chunk i+1 (its own HIP stream and its own pair of output maps) runs while chunk i is parsed and meshed on the caller's
        stream, so the host sync of the parse (the detection count) and the parse / SMPL kernels hide under the next network
        forward.  Yields (outputs dict or None, batch_ids or None, first image index of the chunk).
        `next_images` (round 6): the tensor the NEXT call of forward_chunks will walk (same chunk size; it must stay alive and
        unchanged until then).  The pipeline then stays primed across calls: the network of the next call's first chunk is
        launched under this call's last parse + SMPL (and whatever the caller does between the calls -- packing records, the
        all-gather), and the next call picks it up instead of starting with an exposed network.  A job of few chunks per call (the
        8-GPU shard: 4 calls of 32 per step) otherwise fills and drains the pipeline every step."""
        dev = self.tdevice
        n = images.shape[0]
        starts = list(range(0, n, chunk))
        if not hasattr(self, '_pipe'):
            # ROMP_PIPE_NETS=2 (round 4 experiment, off by default): TWO networks in flight -- even chunks on the net, odd chunks on
            # its twin (RompNet.twin: own arena and graphs, shared weights), each on its own stream, half a period apart, so that one
            # forward's HBM-bound single-kernel phases (stem, layer1, head: 3.9 of a forward's 10.9 ms, scripts/timeline.py) could run
            # beside the other's matrix-bound HRNet modules.  Measured: 2 964 -> 2 977 images/s (+0.4 %) on one box, 2 847 = 2 847 on
            # another; with every kernel capped at one workgroup per CU 2 765: the persistent kernels fill every CU
            # slot, so two forwards time-slice at kernel granularity instead of overlapping.  Not worth a second arena.
            two = os.environ.get('ROMP_PIPE_NETS', '1') == '2' and len(starts) > 1 and self.model.max_batch > 2
            nets = [self.model, self.model.twin()] if two else [self.model, self.model]
            streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)] if two else [torch.cuda.Stream(dev)] * 2
            self._pipe = dict(nets=nets, streams=streams, bufs={}, ev_net=[torch.cuda.Event(), torch.cuda.Event()],
                              ev_free=[torch.cuda.Event(), torch.cuda.Event()], primed=None)
        P = self._pipe
        cur = torch.cuda.current_stream(dev)

        def bufs(B, par):
            key = (B, par)
            if key not in P['bufs']:
                P['bufs'][key] = (torch.empty((B,) + tuple(self.model.out_shapes[0]), device=dev),
                                  torch.empty((B,) + tuple(self.model.out_shapes[1]), device=dev))
            return P['bufs'][key]

        def launch(x, par):
            c, p_ = bufs(x.shape[0], par)
            st = P['streams'][par]
            st.wait_event(P['ev_free'][par])                      # the chunk that used this pair of maps has been parsed
            with torch.cuda.stream(st):
                P['nets'][par].forward_nhwc(x, c, p_)
                P['ev_net'][par].record(st)
            return c, p_

        # a first chunk launched by the previous call (its `next_images` were these images)?  Its parity decides this call's
        primed, P['primed'] = P['primed'], None
        if primed is not None and (primed['ptr'], primed['shape'], primed['chunk']) == (images.data_ptr(), tuple(images.shape), chunk):
            off, pending = primed['par'], primed['bufs']
        else:
            off = 0
            P['ev_free'][0].record(cur)
            P['ev_free'][1].record(cur)
            for st in set(P['streams']):
                st.wait_stream(cur)                               # the images are ready
            pending = launch(images[starts[0]:starts[0] + chunk], off)
        for i, c0 in enumerate(starts):
            center, params = pending
            par = (i + off) & 1
            more = i + 1 < len(starts)
            if more:
                pending = launch(images[starts[i + 1]:starts[i + 1] + chunk], par ^ 1)
            elif next_images is not None and next_images.shape[0] > 0:
                P['primed'] = dict(ptr=next_images.data_ptr(), shape=tuple(next_images.shape), chunk=chunk, par=par ^ 1,
                                   bufs=launch(next_images[:chunk], par ^ 1))
            cur.wait_event(P['ev_net'][par])
            # (range guard: the next chunk's network is already running and bumps the same counter -- a change is charged to this
            # chunk AND the next, RangeGuard.check(next_in_flight=True))
            self.range_guard.pipelined = more or P['primed'] is not None
            outputs, batch_ids, rerun = parsing_outputs(center.unsqueeze(1), params, self.centermap_parser, return_batch_ids=True,
                                                        guard=self.range_guard, quiet=True)
            self.range_guard.pipelined = False
            if rerun:
                c32, p32 = self._rerun_f32(images[c0:c0 + chunk])
                outputs, batch_ids = parsing_outputs(c32, p32, self.centermap_parser, return_batch_ids=True)
            if outputs is not None:
                outputs['cam_trans'] = convert_cam_to_3d_trans(outputs['cam'])
                if self.settings.calc_smpl:
                    outputs = self.smpl_parser(outputs, root_align=self.settings.root_align)
            P['ev_free'][par].record(cur)
            yield outputs, batch_ids, c0
        if P['primed'] is None:                                   # (primed: the caller's stream must NOT wait for the next call's network)
            for st in set(P['streams']):
                cur.wait_stream(st)


def _imread_bgr(path):
    """cv2.imread(path) (BGR uint8), or the same array through PIL when OpenCV is not installed; None if unreadable."""
    try:
        import cv2
        return cv2.imread(path)
    except ImportError:
        from PIL import Image
        try:
            return np.ascontiguousarray(np.asarray(Image.open(path).convert('RGB'))[:, :, ::-1])
        except OSError:
            return None


def main():
    """main.py:178-204 (image mode; video/webcam need OpenCV)."""
    args = romp_settings()
    romp = ROMP(args)
    if args.mode == 'image':
        saver = ResultSaver(args.mode, args.save_path)
        image = _imread_bgr(args.input)
        outputs = romp(image)
        saver(outputs, args.input)
    else:
        raise NotImplementedError('mode %s needs OpenCV capture; only --mode image is wired here' % args.mode)


if __name__ == '__main__':
    main()
