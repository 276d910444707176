"""SMPL on the HIP device -- mirror of simple_romp/romp/smpl.py:37-108 (class ``SMPL``).

Same constructor and call signature: ``SMPL(model_path, model_type='smpl')(betas, poses,
root_align=False) -> (verts (N,6890,3), joints (N,71,3), faces)``.  The model file schema is
the reference's packed ``.pth`` (pack_smpl_info.py:70-111).  PyTorch holds the buffers; the
forward is ``smpl_forward`` of libromp_hip.so (csrc/smpl.hip).
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import lib as L


class SMPL(nn.Module):
    def __init__(self, model_path, model_type='smpl', dtype=torch.float32):
        super(SMPL, self).__init__()
        self.dtype = dtype
        model_info = model_path if isinstance(model_path, dict) else torch.load(model_path)
        self.register_buffer('extra_joints_idxs', model_info['extra_joints_index'].long())
        self.register_buffer('J_regressor_extra9', model_info['J_regressor_extra9'].float())
        self.register_buffer('J_regressor_h36m17', model_info['J_regressor_h36m17'].float())
        self.register_buffer('faces_tensor', model_info['f'])
        self.register_buffer('v_template', model_info['v_template'].float())
        key = 'shapedirs' if model_type == 'smpl' else 'smpla_shapedirs'     # smpl.py:49-52
        self.register_buffer('shapedirs', model_info[key].float())
        self.register_buffer('J_regressor', model_info['J_regressor'].float())
        self.register_buffer('posedirs', model_info['posedirs'].float())
        self.register_buffer('parents', model_info['kintree_table'].long())
        self.register_buffer('lbs_weights', model_info['weights'].float())
        self._ctx = None
        self._ctx_device = None

    def _context(self):
        dev = self.shapedirs.device
        if dev.type != 'cuda':
            raise L.RompHipError('SMPL runs on the HIP device only (no CPU fallback); call .to("cuda:N")')
        if self._ctx is not None and self._ctx_device == dev:
            return self._ctx
        self._release()
        lib = L.load()
        parents = (C.c_int64 * 24)(*[int(v) for v in self.parents.cpu().tolist()])
        extra = (C.c_int64 * 21)(*[int(v) for v in self.extra_joints_idxs.cpu().tolist()])
        h = C.c_void_p()
        c = lambda t: L.ptr(t.contiguous())
        with torch.cuda.device(dev):
            L.check(lib.smpl_ctx_create(C.byref(h), c(self.v_template), c(self.shapedirs), int(self.shapedirs.shape[-1]),
                                        c(self.posedirs), c(self.J_regressor), c(self.lbs_weights), parents,
                                        c(self.J_regressor_extra9), c(self.J_regressor_h36m17), extra, 64,
                                        L.stream_ptr(dev)))
        self._ctx, self._ctx_device = h, dev
        return h

    def _release(self):
        if getattr(self, '_ctx', None) is not None:
            L.load().smpl_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def forward(self, betas=None, poses=None, root_align=False):
        if isinstance(betas, np.ndarray):
            betas = torch.from_numpy(betas).type(self.dtype)
        if isinstance(poses, np.ndarray):
            poses = torch.from_numpy(poses).type(self.dtype)
        dev = self.shapedirs.device
        betas = betas.to(dev).float().contiguous()
        poses = poses.to(dev).float().contiguous()
        ctx = self._context()
        N = betas.shape[0]
        verts = torch.empty(N, 6890, 3, device=dev, dtype=torch.float32)
        joints = torch.empty(N, 71, 3, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            L.check(L.load().smpl_forward(ctx, L.ptr(betas), int(betas.shape[1]), L.ptr(poses), N, int(bool(root_align)),
                                          L.ptr(verts), L.ptr(joints), L.stream_ptr(dev)))
        return verts, joints, self.faces_tensor
