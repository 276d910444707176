"""RompNet -- device-side ROMPv1 (HRNet-32 + head) behind the reference's network seam.

Replaces ``ROMPv1.forward`` (simple_romp/romp/model.py:470-481) / the ONNX session call at
``main.py:109-112``: ``(B,512,512,3) float 0..255 -> center_maps (B,1,64,64),
params_maps (B,145,64,64)``.  PyTorch-ROCm owns the I/O tensors; all arithmetic is in
libromp_hip.so (csrc/conv_mfma.hip, stem_fuse.hip, net.hip).
"""
import ctypes as C
import os

import torch

from . import lib as L
from .plan import Program, build_romp_hrnet32, coord_channels, decode_h2, encode_h2


_CHECK_FINITE = os.environ.get('ROMP_CHECK_FINITE', '0') not in ('', '0')
# The range guard (RangeGuard below) is ON by default.  ROMP_RANGE_GUARD=0 exists for measuring what it costs, not as a product
# option: no guard (the round-5 behaviour: clamped maps pass silently).  Every kernel that forms fp16 pieces counts its clamps in
# every build (the two fused BasicBlock kernels since round 6: one v_pk_maximum3_f16 per four values), so there is nothing else to
# switch: the guard's own cost is the read-back, measured inside the run-to-run spread (profiles/r06_guard_cost_packed.txt).
_RANGE_GUARD = os.environ.get('ROMP_RANGE_GUARD', '1')


class RangeGuard:
    """Default-on range safety of the f16x2 arithmetic.  The reference's network is float32 and has no range to leave
    (simple_romp/romp/main.py:106-115); the f16x2 kernels clamp a value beyond 65504 / 2^act_shift while splitting it into fp16
    pieces -- finite, wrong -- and calibration (RompNet._measure_ranges) can only vouch for the frames it saw.  Every clamp bumps
    the net's device counter (conv_common.h sat_report / sat_report_pk: every kernel, every build).  The API reads that counter back WITH the detection count of every call (romp_parse_watch: the same D2H, the
    same synchronisation -- no extra round trip) and hands it to `check`; a change means this call clamped somewhere, and the
    caller re-runs it on the exact-f32 program of the same weights (`RompNet.f32_twin`, built on first need) -- so no call of the
    API returns clamped maps.  Pipelined callers (ROMP.forward_chunks) have the NEXT network already in flight when they read the
    counter: a change is then charged to this call AND the next (`carry`), which is conservative and cannot miss one."""

    def __init__(self, net):
        self.net = net
        self.enabled = bool(net.bf16x3) and _RANGE_GUARD != '0'
        self.seen, self.carry, self.reruns, self.warned = 0, False, 0, False
        if self.enabled:
            self.seen = net.saturated & 0xffffffff

    @property
    def watch(self):
        """Device address of the counter for romp_parse_watch (None: guard off)."""
        return self.net.sat_counter if self.enabled else None

    def check(self, value, next_in_flight=False):
        """`value`: the counter as it came back with this call's counts (None: guard off).  True: re-run this call in float32."""
        if not self.enabled or value is None:
            return False
        value &= 0xffffffff
        changed = value != self.seen
        hit = changed or self.carry
        self.carry = changed and next_in_flight
        self.seen = value
        if hit:
            self.reruns += 1
        return hit

    def resync(self):
        """After anything else bumped the counter (a range scan): take its present value as seen."""
        if self.enabled:
            self.seen = self.net.saturated & 0xffffffff

    def warn(self, images=None):
        """Once per net: say that a forward left the calibrated range, and (given the frames) which ops clamped."""
        if self.warned:
            return
        self.warned = True
        import warnings
        ops = ''
        if images is not None:
            try:
                torch.cuda.synchronize(self.net.device)        # (a pipelined caller has another forward of this net in flight: the scan uses the same arena)
                scan = self.net.range_scan(images[:min(int(images.shape[0]), 2)])
                names = [nm for nm, _, _, sat in scan if sat > 0]
                ops = '; clamping ops: ' + (', '.join(names[:8]) + (' ... (%d in all)' % len(names) if len(names) > 8 else '') if names else 'none on the first frames')
                self.resync()
            except L.RompHipError:
                pass
        warnings.warn('romp_amd: activations left the calibrated range of the f16x2 kernels (values beyond 65504 / 2^act_shift were clamped); '
                      'the call was re-run on the exact-f32 program and its result is the float32 one.  Calibrate on representative '
                      'frames (--calib_dir / calib_images=) or use --conv_math f32 to avoid the second forward' + ops)


class RompNet:
    def __init__(self, state_dict, device='cuda:0', max_batch=32, input_size=512, use_graph=False, builder=None,
                 out_shapes=None, bf16x3=False, split_k=None, calibrate=None, calib_images=None, calib_margin=None):
        """`builder(state_dict, device, input_size, bf16x3=) -> Program` (default: ROMP HRNet-32 + head);
        `out_shapes`: per-image shapes of the two output tensors of the program.  `bf16x3` is the conv_math
        setting: False / 'f32', True / 'bf16x3', 'f16x2' or 'all' (plan.set_conv_math).  `split_k`: lower the layers with few
        pixels and many input channels as split-K convs (plan.Program.conv); default: only for single-image nets
        (max_batch <= 2), where those layers are a handful of work items with a long serial channel loop; an int sets the
        work-item target (default 128).  `calibrate` (default: whenever the f16x2 kernels are on offer): measure every
        tensor's max|x| with a float32 forward of `calib_images` ((B,S,S,3) float 0..255 on the device; default: synthetic
        frames) and keep tensors / layers whose range does not fit the fp16 pieces on the float32 path (plan.assign_formats,
        `self.range_fallback` lists them); False skips it (ranges are then trusted to fit, the kernels only saturate).
        `calib_margin`: head-room factor between the largest calibrated value of a tensor and the fp16 limit (default
        plan.CALIB_MARGIN = 4; the synthetic default frames are not photos -- pass real frames through `calib_images`, e.g.
        ROMP(--calib_dir), for a production checkpoint).  What calibration cannot foresee stays OBSERVABLE: `self.saturated`
        counts the clamps the kernels report (0 for a healthy net), `range_scan(images)` says which op clamped."""
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise L.RompHipError('RompNet needs a HIP device (the HIP path has no CPU fallback)')
        self.lib = L.load()
        self.max_batch = int(max_batch)
        self.bf16x3 = bf16x3 not in (False, None, 'f32')      # split-precision kernels on offer: pick by measurement
        self._tuned = set()
        self.split = 1
        self.input_size = input_size
        with torch.cuda.device(self.device):
            self.split_k = (128 if self.max_batch <= 2 else 0) if split_k is None else (128 if split_k is True else int(split_k))
            kw = dict(split_k_items=self.split_k) if self.split_k else {}
            if builder is None:
                builder = build_romp_hrnet32
            self.program: Program = builder(state_dict, self.device, input_size, bf16x3=bf16x3, **kw)
            self._src = (state_dict, builder, kw, out_shapes)        # what f32_twin() lowers again (references, no copies)
            if calibrate is None:
                calibrate = bool(getattr(self.program, 'f16x2', False))
            self.op_maxabs = None
            if calibrate:
                ms = input_size // 8
                self.op_maxabs = self._measure_ranges(state_dict, builder, kw, input_size, out_shapes or ((ms, ms), (ms, ms, 145)),
                                                      calib_images)
                self.program.op_maxabs = self.op_maxabs
                if calib_margin is not None:
                    self.program.calib_margin = float(calib_margin)
            ops = self.program.op_array()
            self.range_fallback = list(getattr(self.program, 'range_fallback', []))
            sizes = (C.c_int64 * len(self.program.buf_floats))(*self.program.buf_floats)
            h = C.c_void_p()
            L.check(self.lib.romp_net_create(C.byref(h), ops, len(self.program.ops), sizes,
                                             len(self.program.buf_floats), self.max_batch))
            self._h = h
            ms = input_size // 8
            self.out_shapes = out_shapes or ((ms, ms), (ms, ms, 145))
            if self.program.coord_off is not None:      # ROMP head: constant CoordConv channels of the head input (model.py:473)
                fs = input_size // 4
                coords = coord_channels(self.max_batch, fs, self.device, self.program.head_in_ch, self.program.coord_off)
                if self.program.buf_fmt.get(self.program.head_in_buf) == L.FMT_H2:     # the head input lives pre-split: constants too
                    coords = encode_h2(coords.cpu()).to(self.device)
                L.check(self.lib.romp_net_write_buffer(self._h, self.program.head_in_buf, L.ptr(coords),
                                                       coords.numel(), L.stream_ptr(self.device)))
            torch.cuda.synchronize(self.device)
        if use_graph:
            self.set_graph(True)

    def _measure_ranges(self, state_dict, builder, kw, input_size, out_shapes, calib_images):
        """max|x| per op output of the same program lowered with conv_math='f32' (every tensor float32, exact-f32 kernels), over
        a few calibration frames -> list aligned with self.program.ops (None where an op writes no arena buffer)."""
        import math
        P32 = builder(state_dict, self.device, input_size, bf16x3='f32', **kw)
        if calib_images is None:
            g = torch.Generator(device='cpu').manual_seed(20240924)
            lin = torch.linspace(0, 255, input_size)
            calib_images = torch.stack([
                torch.rand(input_size, input_size, 3, generator=g) * 255.0,                      # white noise
                (lin[:, None, None] * 0.5 + lin[None, :, None] * 0.5).expand(-1, -1, 3).clone(),   # smooth ramp
                (torch.rand(input_size // 16, input_size // 16, 3, generator=g) * 255.0).repeat_interleave(16, 0).repeat_interleave(16, 1),
            ]).to(self.device)
        calib_images = calib_images.to(self.device, torch.float32).contiguous()
        B = min(int(calib_images.shape[0]), 4)
        ops32 = P32.op_array()
        sizes = (C.c_int64 * len(P32.buf_floats))(*P32.buf_floats)
        h = C.c_void_p()
        L.check(self.lib.romp_net_create(C.byref(h), ops32, len(P32.ops), sizes, len(P32.buf_floats), B))
        try:
            if P32.coord_off is not None:
                coords = coord_channels(B, input_size // 4, self.device, P32.head_in_ch, P32.coord_off)
                L.check(self.lib.romp_net_write_buffer(h, P32.head_in_buf, L.ptr(coords), coords.numel(), L.stream_ptr(self.device)))
            c = torch.empty((B,) + tuple(out_shapes[0]), device=self.device)
            q = torch.empty((B,) + tuple(out_shapes[1]), device=self.device)
            mx = (C.c_float * len(P32.ops))()
            bad = (C.c_int32 * len(P32.ops))()
            L.check(self.lib.romp_net_range_scan(h, L.ptr(calib_images[:B]), B, L.ptr(c), L.ptr(q), L.stream_ptr(self.device), mx, bad, None))
        finally:
            self.lib.romp_net_destroy(h)
        # The two lowerings name their layers alike but need not have the same op list: a single-image float32 program splits the
        # input channels of its deep 3x3 layers into `<name>.splitk` + `<name>.ksum`, the f16x2 program runs them as ONE conv on
        # csrc/conv_h2k.hip.  A layer's output tensor is what its last op wrote: matched by layer name.
        by_name = {}
        for i, op in enumerate(P32.ops):
            if op.out_buf < 0 or op.kind in (L.OP_FORK, L.OP_JOIN) or P32.names[i].endswith('.splitk'):
                continue
            name = P32.names[i][:-len('.ksum')] if P32.names[i].endswith('.ksum') else P32.names[i]
            by_name[name] = math.inf if bad[i] else float(mx[i])
        out = []
        for i, op in enumerate(self.program.ops):
            name = self.program.names[i]
            if op.out_buf < 0 or op.kind in (L.OP_FORK, L.OP_JOIN) or name.endswith('.splitk'):
                out.append(None)
                continue
            key = name[:-len('.ksum')] if name.endswith('.ksum') else name
            assert key in by_name, 'calibration: layer %s of the program has no counterpart in the float32 lowering' % key
            out.append(by_name[key])
        return out

    def twin(self):
        """A second executor of the SAME lowered program on the same device: its own activation arena, work queues, side
        streams and hipGraphs; the packed constants (weights) are shared.  `ROMP.forward_chunks` alternates the chunks of a job
        between a net and its twin on two HIP streams: the single-kernel phases of one forward (the stem, layer1's HBM-bound
        bottlenecks, the head: a third of a forward's wall time, scripts/timeline.py) then run beside the other forward's
        matrix-bound HRNet modules instead of alone.  Variant tables installed so far are copied."""
        if getattr(self, '_plan_path', None) is not None:
            t = RompNet.from_plan(self._plan_path, self.device, max_batch=self.max_batch, out_shapes=self.out_shapes)
            t.set_graph(getattr(self, '_use_graph', False))
            return t
        t = RompNet.__new__(RompNet)
        for k in ('device', 'lib', 'max_batch', 'bf16x3', 'split', 'split_k', 'input_size', 'program', 'op_maxabs', 'range_fallback', 'out_shapes', '_src'):
            setattr(t, k, getattr(self, k, None))
        t._tuned = set()
        with torch.cuda.device(self.device):
            ops = self.program.op_array()
            sizes = (C.c_int64 * len(self.program.buf_floats))(*self.program.buf_floats)
            h = C.c_void_p()
            L.check(self.lib.romp_net_create(C.byref(h), ops, len(self.program.ops), sizes, len(self.program.buf_floats), self.max_batch))
            t._h = h
            if self.program.coord_off is not None:
                fs = self.input_size // 4
                coords = coord_channels(self.max_batch, fs, self.device, self.program.head_in_ch, self.program.coord_off)
                if self.program.buf_fmt.get(self.program.head_in_buf) == L.FMT_H2:
                    coords = encode_h2(coords.cpu()).to(self.device)
                L.check(self.lib.romp_net_write_buffer(t._h, self.program.head_in_buf, L.ptr(coords), coords.numel(), L.stream_ptr(self.device)))
            torch.cuda.synchronize(self.device)
        for B in sorted(self._tuned):
            t.set_tuned(B, self.tuned_variants(B))
        t.set_graph(getattr(self, '_use_graph', False))
        if not getattr(self, '_use_streams', True):
            t.set_streams(False)
        return t

    def f32_twin(self):
        """The exact-f32 lowering of the same weights (conv_math='f32': v_mfma_f32_32x32x2_f32 products, float32 tensors -- no
        fp16 range anywhere), built on first need and kept: what RangeGuard re-runs a call on whose activations left the
        calibrated range.  A net loaded from a plan file holds only its own packed constants and cannot build one."""
        t = getattr(self, '_f32', None)
        if t is not None:
            return t
        src = getattr(self, '_src', None)
        if src is None:
            raise L.RompHipError('activations left the calibrated range of the f16x2 kernels and this net was loaded from a plan file, which '
                                 'holds no float32 program to fall back to: export the plan with conv_math=f32 or calibrated on '
                                 'representative frames (python -m romp_amd.export --calib_dir), or start from --model_path')
        state_dict, builder, kw, out_shapes = src
        t = RompNet(state_dict, self.device, max_batch=self.max_batch, input_size=self.input_size, builder=builder, out_shapes=out_shapes,
                    bf16x3='f32', split_k=self.split_k, calibrate=False)
        if not getattr(self, '_use_streams', True):
            t.set_streams(False)
        t.set_graph(getattr(self, '_use_graph', False))
        self._f32 = t
        return t

    @property
    def sat_counter(self):
        """Device address of the saturation counter (romp_net_sat_counter): the `watch` argument of romp_parse_watch."""
        return int(self.lib.romp_net_sat_counter(self._h) or 0)

    @classmethod
    def from_plan(cls, path, device, max_batch=32, use_graph=False, out_shapes=None):
        """A net from a plan file (export.save_plan): no state_dict, no lowering -- libromp_hip.so's romp_net_load does it all.
        Variant tables measured before the export are installed when the file was written by the same build."""
        from types import SimpleNamespace
        from .export import read_plan
        self = cls.__new__(cls)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise L.RompHipError('RompNet needs a HIP device (the HIP path has no CPU fallback)')
        self.lib = L.load()
        self.max_batch = int(max_batch)
        plan = read_plan(path)
        self._plan_path, self._src = path, None
        ops = plan['ops']
        self.bf16x3 = any(o.weight_h2 or o.weight_aux for o in ops)
        self.split, self.split_k, self.input_size = 1, plan['split_k_items'], plan['input_size']     # (the plan KIND is in the header)
        buf_fmt = {}
        for o in ops:
            if o.out_buf >= 0:
                buf_fmt[o.out_buf] = o.out_fmt
        self.program = SimpleNamespace(ops=list(ops), names=['op%d' % i for i in range(len(ops))], buf_floats=plan['buf_floats'],
                                       buf_fmt=buf_fmt, flops=[0.0] * len(ops), bytes=[0.0] * len(ops), coord_off=None,
                                       split_k_items=plan['split_k_items'])
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.romp_net_load(C.byref(h), str(path).encode(), self.max_batch))
        self._h = h
        kind = C.c_int32(-1)
        L.check(self.lib.romp_net_plan_kind(h, C.byref(kind)))
        assert kind.value == self.split_k, (kind.value, self.split_k)     # the C reader and read_plan parse one header
        n_variants = self.lib.romp_conv_num_variants()
        self._tuned = {B for B, (nv, _) in plan['tuned'].items() if nv == n_variants and B <= self.max_batch}
        ms = self.input_size // 8
        cf, pf = plan['center_floats'], plan['params_floats']
        self.out_shapes = out_shapes or (((ms, ms), (ms, ms, pf // (ms * ms))) if cf == ms * ms and pf % (ms * ms) == 0 else ((cf,), (pf,)))
        if use_graph:
            self.set_graph(True)
        return self

    # -- configuration -------------------------------------------------------------------
    def set_mode(self, mode):
        """0 = MFMA kernels (default), 1 = naive direct-conv cross-check kernels."""
        L.check(self.lib.romp_net_set_mode(self._h, int(mode)))

    def set_streams(self, enable):
        """Run independent HRNet branches on side HIP streams (default on)."""
        self._use_streams = bool(enable)
        L.check(self.lib.romp_net_set_streams(self._h, int(bool(enable))))

    def set_split(self, lanes, wg_cap=1):
        """lanes=2: run even batches as two half-batch lanes on two streams (convs capped at `wg_cap`
        workgroups per CU so the lanes' kernels co-reside); kernel variants are then tuned for B/2."""
        import math
        L.check(self.lib.romp_net_set_split(self._h, int(lanes), int(wg_cap), self.input_size * self.input_size * 3,
                                            math.prod(self.out_shapes[0]), math.prod(self.out_shapes[1])))
        self.split = int(lanes)

    def set_graph(self, enable):
        self._use_graph = bool(enable)
        L.check(self.lib.romp_net_set_graph(self._h, int(bool(enable))))

    # -- forward -------------------------------------------------------------------------
    def forward_nhwc(self, image, center_out=None, params_out=None):
        """image (B,H,W,3) float32 on device -> center (B,64,64), params (B,64,64,145) NHWC."""
        assert image.dtype == torch.float32 and image.is_cuda and image.dim() == 4 and image.shape[-1] == 3
        image = image.contiguous()
        B = image.shape[0]
        if tuple(image.shape[1:3]) != (self.input_size, self.input_size):
            raise L.RompHipError('RompNet was lowered for %dx%d inputs, got %dx%d' % (self.input_size, self.input_size, image.shape[1], image.shape[2]))
        if B > self.max_batch:
            raise L.RompHipError('batch %d > max_batch %d' % (B, self.max_batch))
        Bt = B // 2 if (self.split == 2 and B >= 2 and B % 2 == 0) else B     # batch the kernels really see
        if self.bf16x3 and Bt not in self._tuned:     # the bf16x3 kernels are only ever picked by measurement
            self.autotune(Bt)
        if center_out is None:
            center_out = torch.empty((B,) + tuple(self.out_shapes[0]), device=self.device, dtype=torch.float32)
        if params_out is None:
            params_out = torch.empty((B,) + tuple(self.out_shapes[1]), device=self.device, dtype=torch.float32)
        L.check(self.lib.romp_net_forward(self._h, L.ptr(image), B, L.ptr(center_out), L.ptr(params_out),
                                          L.stream_ptr(self.device)))
        if _CHECK_FINITE:                                     # debug mode (env ROMP_CHECK_FINITE=1): one reduction + sync per forward
            if not (bool(torch.isfinite(center_out).all()) and bool(torch.isfinite(params_out).all())):
                raise L.RompHipError('non-finite values in the network outputs (activation range outside the f16x2 kernels\' '
                                     'fp16 pieces? build the net with calibrate=True / conv_math=\'f32\')')
            n_sat = self.saturated                            # (the fused-block kernels run their counting builds in this mode)
            if n_sat and not getattr(self, '_sat_warned', False):
                import warnings
                self._sat_warned = True
                warnings.warn('RompNet: %d saturation events -- activations beyond the calibrated range of the f16x2 kernels were '
                              'clamped at 65504 / 2^act_shift (finite but wrong there); calibrate on representative frames '
                              '(calib_images= / --calib_dir) or use conv_math=\'f32\'; range_scan(images) names the ops' % n_sat)
        return center_out, params_out

    @property
    def saturated(self):
        """Saturation events since the net was built (or `reset_saturated()`): how many (wave, work item) times a kernel clamped a
        value at +-65504 while splitting it into the fp16 pieces of the H2 format -- 0 for a net inside its calibrated range.
        Synchronises the current stream."""
        v = C.c_int64(0)
        L.check(self.lib.romp_net_saturated(self._h, C.byref(v), 0, L.stream_ptr(self.device)))
        return int(v.value)

    def reset_saturated(self):
        v = C.c_int64(0)
        L.check(self.lib.romp_net_saturated(self._h, C.byref(v), 1, L.stream_ptr(self.device)))
        return int(v.value)

    def set_sat_check(self, enable):
        """(Rounds 4-5: run the counting builds of the fused BasicBlock kernels.  Since round 6 those kernels always count; the call is
        kept for hosts written against ABI 5 and changes nothing.)"""
        L.check(self.lib.romp_net_set_sat_check(self._h, int(bool(enable))))

    def range_scan(self, image):
        """Diagnosis on THIS net's program (not the float32 lowering calibration uses): runs it op by op on `image` and returns
        per op (name, max|x| of the region it wrote, non-finite count, saturation events it reported)."""
        B = image.shape[0]
        Bt = B // 2 if (self.split == 2 and B >= 2 and B % 2 == 0) else B
        if self.bf16x3 and Bt not in self._tuned:
            self.autotune(Bt)
        c = torch.empty((B,) + tuple(self.out_shapes[0]), device=self.device)
        q = torch.empty((B,) + tuple(self.out_shapes[1]), device=self.device)
        n = len(self.program.ops)
        mx, bad, sat = (C.c_float * n)(), (C.c_int32 * n)(), (C.c_int32 * n)()
        L.check(self.lib.romp_net_range_scan(self._h, L.ptr(image.contiguous()), B, L.ptr(c), L.ptr(q), L.stream_ptr(self.device), mx, bad, sat))
        return [(self.program.names[i], float(mx[i]), int(bad[i]), int(sat[i])) for i in range(n)]

    def __call__(self, image):
        """Reference layout: center_maps (B,1,64,64), params_maps (B,145,64,64) (views, no copy)."""
        c, p = self.forward_nhwc(image)
        return c.unsqueeze(1), p.permute(0, 3, 1, 2)

    def autotune(self, B, iters=5):
        """Pick the fastest conv kernel variant per layer for batch size B (measured on device)."""
        with torch.cuda.device(self.device):
            L.check(self.lib.romp_net_autotune(self._h, int(B), int(iters), L.stream_ptr(self.device)))
        self._tuned.add(int(B))

    def tuned_variants(self, B):
        """Variant index per op for batch B (-1 = heuristic): what autotune(B) chose."""
        return [self.lib.romp_net_tuned_variant(self._h, int(B), i) for i in range(len(self.program.ops))]

    def set_tuned(self, B, variants):
        """Install a saved variant table (from tuned_variants) instead of measuring."""
        arr = (C.c_int32 * len(variants))(*[int(v) for v in variants])
        L.check(self.lib.romp_net_set_tuned(self._h, int(B), arr, len(variants)))
        self._tuned.add(int(B))

    def variant_names(self, B):
        """Kernel variant name per op (tuned choice if autotune(B) ran, else the heuristic)."""
        buf = C.create_string_buffer(128)
        names = []
        for i, op in enumerate(self.program.ops):
            v = self.lib.romp_net_tuned_variant(self._h, int(B), i)
            L.check(self.lib.romp_conv_describe(C.byref(op), int(B), v, buf, 128))
            names.append(buf.value.decode())
        return names

    def buffer_ptr(self, buf):
        """Device address of arena buffer `buf`."""
        return self.lib.romp_net_buffer_ptr(self._h, int(buf))

    def read_buffer(self, buf, B, channels=None):
        """Debug / test helper: arena buffer `buf` of B images as float32 values.  A tensor stored in the H2 format is decoded
        (`channels` = its channel stride, needed to find the octets)."""
        n = self.program.buf_floats[buf] * B
        out = torch.empty(n, device=self.device, dtype=torch.float32)
        L.check(self.lib.romp_net_read_buffer(self._h, buf, B, L.ptr(out), n, L.stream_ptr(self.device)))
        if self.program.buf_fmt.get(buf) == L.FMT_H2:
            assert channels is not None and channels % 8 == 0, 'H2 buffer: pass its channel stride'
            torch.cuda.synchronize(self.device)
            out = decode_h2(out.reshape(-1, channels)).reshape(-1)
        return out

    def profile(self, image, iters=3):
        """Per-op mean milliseconds (HIP events on the current stream)."""
        B = image.shape[0]
        c = torch.empty((B,) + tuple(self.out_shapes[0]), device=self.device)
        p = torch.empty((B,) + tuple(self.out_shapes[1]), device=self.device)
        out = (C.c_float * len(self.program.ops))()
        L.check(self.lib.romp_net_profile(self._h, L.ptr(image.contiguous()), B, L.ptr(c), L.ptr(p),
                                          L.stream_ptr(self.device), out, iters))
        return list(out)

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and getattr(self, 'lib', None) is not None:
            self.lib.romp_net_destroy(h)
            self._h = None
