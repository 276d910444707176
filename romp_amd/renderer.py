"""Sim3DR mesh renderer on the HIP device -- drop-in for ``simple_romp/vis_human/sim3drender``
(``renderer.py``: ``Sim3DR``, ``rasterize``, ``get_normal``; the Cython extension ``Sim3DR_Cython`` is
replaced by ``romp_sim3dr_*`` in libromp_hip.so, csrc/render.hip).  Images are bit-identical to the
reference's (tests/test_render.py).  No CPU path: a missing HIP device / extension raises.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L


def _to_ctype(arr):
    return arr if arr.flags.c_contiguous else arr.copy(order='C')


def convert_type(obj):
    """renderer.py:27-30."""
    if isinstance(obj, (tuple, list)):
        return np.array(obj, dtype=np.float32)[None, :]
    return obj


class _Topology(object):
    """Triangles on the device + the vertex -> (triangle, corner) incidence lists in ascending order."""

    def __init__(self, triangles, nver, device):
        tri = np.ascontiguousarray(triangles, np.int32)
        flat = tri.reshape(-1).astype(np.int64)
        order = np.argsort(flat, kind='stable').astype(np.int32)           # stable: ascending corner index per vertex
        counts = np.bincount(flat, minlength=nver)
        off = np.zeros(nver + 1, np.int32)
        off[1:] = np.cumsum(counts)
        self.ntri, self.nver = tri.shape[0], nver
        self.tri = torch.from_numpy(tri).to(device)
        self.adj_off = torch.from_numpy(off).to(device)
        self.adj_ent = torch.from_numpy(order).to(device)


_topologies = {}


def _topology(triangles, nver, device):
    tri = np.ascontiguousarray(triangles, np.int32)
    key = (tri.shape, nver, str(device), hash(tri.tobytes()))
    if key not in _topologies:
        if len(_topologies) > 8:
            _topologies.clear()
        _topologies[key] = _Topology(tri, nver, device)
    return _topologies[key]


def _device(device=None):
    if not torch.cuda.is_available():
        raise L.RompHipError('romp_amd.renderer needs a HIP device; there is no CPU fallback')
    return torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())


def get_normal(vertices, triangles, device=None):
    """renderer.py:32-37: per-vertex normals (numpy in, numpy out)."""
    dev = _device(device)
    lib = L.load()
    v = torch.from_numpy(np.ascontiguousarray(vertices, np.float32)).to(dev)
    topo = _topology(triangles, v.shape[0], dev)
    out = torch.empty_like(v)
    L.check(lib.romp_sim3dr_normals(L.ptr(v), L.ptr(topo.tri), L.ptr(topo.adj_off), L.ptr(topo.adj_ent), v.shape[0], L.ptr(out),
                                    L.stream_ptr(dev)))
    return out.cpu().numpy()


def _rasterize_dev(image_dev, v_dev, topo, colors_dev, reverse, keys):
    h, w, c = image_dev.shape
    L.check(L.load().romp_sim3dr_rasterize(L.ptr(image_dev), L.ptr(v_dev), L.ptr(topo.tri), L.ptr(colors_dev), topo.ntri, h, w, c,
                                           int(bool(reverse)), L.ptr(keys), L.stream_ptr(image_dev.device)))


def rasterize(vertices, triangles, colors, bg=None, height=None, width=None, channel=None, reverse=False, device=None):
    """renderer.py:39-62: z-buffer rasterization of per-vertex colours onto `bg` (modified in place and
    returned, like the reference)."""
    if bg is None:
        assert height is not None and width is not None and channel is not None
        bg = np.zeros((height, width, channel), dtype=np.uint8)
    dev = _device(device)
    img = torch.from_numpy(np.ascontiguousarray(bg)).to(dev)
    v = torch.from_numpy(np.ascontiguousarray(vertices, np.float32)).to(dev)
    col = torch.from_numpy(np.ascontiguousarray(colors, np.float32)).to(dev)
    keys = torch.empty(img.shape[0] * img.shape[1], dtype=torch.int64, device=dev)
    _rasterize_dev(img, v, _topology(triangles, v.shape[0], dev), col, reverse, keys)
    bg[...] = img.cpu().numpy()
    return bg


class Sim3DR(object):
    """renderer.py:64-133.  `__call__(verts_list, triangles, bg, mesh_colors)` paints the meshes one after the
    other (each with a fresh z-buffer) onto a copy of `bg` and returns the uint8 image."""

    def __init__(self, **kwargs):
        self.intensity_ambient = convert_type(kwargs.get('intensity_ambient', 0.66))
        self.intensity_directional = convert_type(kwargs.get('intensity_directional', 0.36))
        self.intensity_specular = convert_type(kwargs.get('intensity_specular', 0.1))
        self.specular_exp = kwargs.get('specular_exp', 1)
        self.color_directional = convert_type(kwargs.get('color_directional', (1, 1, 1)))
        self.light_pos = convert_type(kwargs.get('light_pos', (0, 0, -5)))
        self.view_pos = convert_type(kwargs.get('view_pos', (0, 0, 5)))
        self.device = kwargs.get('device', None)
        if self.specular_exp != 1:
            raise NotImplementedError('specular_exp != 1 is not on the device path (the reference default is 1)')

    def update_light_pos(self, light_pos):
        self.light_pos = convert_type(light_pos)

    def _light_cfg(self, color):
        """The 14 floats of romp_sim3dr_light.  The ambient term is formed exactly as renderer.py:83 does: a
        float64 product added into a float32 zero buffer."""
        amb = np.zeros((1, 3), np.float32)
        if self.intensity_ambient > 0:
            amb += self.intensity_ambient * np.array(color)
        i_dir = np.float32(self.intensity_directional) if self.intensity_directional > 0 else np.float32(0)
        i_spec = np.float32(self.intensity_specular) if self.intensity_specular > 0 else np.float32(0)
        cfg = np.concatenate([amb.reshape(3), [i_dir, i_spec], np.asarray(self.color_directional, np.float32).reshape(3),
                              np.asarray(self.light_pos, np.float32).reshape(3), np.asarray(self.view_pos, np.float32).reshape(3)])
        return (C.c_float * 14)(*[float(x) for x in cfg.astype(np.float32)])

    def _render_dev(self, img, v, topo, color, keys, light):
        lib = L.load()
        st = L.stream_ptr(img.device)
        normal = torch.empty_like(v)
        L.check(lib.romp_sim3dr_normals(L.ptr(v), L.ptr(topo.tri), L.ptr(topo.adj_off), L.ptr(topo.adj_ent), topo.nver, L.ptr(normal), st))
        L.check(lib.romp_sim3dr_light(L.ptr(v), L.ptr(normal), topo.nver, self._light_cfg(color), L.ptr(light), st))
        _rasterize_dev(img, v, topo, light, False, keys)

    def render(self, vertices, triangles, bg, color=np.array([[1, 0.6, 0.4]]), texture=None):
        """renderer.py:76-118 for one mesh (numpy in; `bg` is modified in place and returned)."""
        if texture is not None:
            raise NotImplementedError('textured rendering is not on the device path')
        dev = _device(self.device)
        img = torch.from_numpy(np.ascontiguousarray(bg)).to(dev)
        v = torch.from_numpy(np.ascontiguousarray(vertices, np.float32)).to(dev)
        keys = torch.empty(img.shape[0] * img.shape[1], dtype=torch.int64, device=dev)
        self._render_dev(img, v, _topology(triangles, v.shape[0], dev), color, keys, torch.empty_like(v))
        bg[...] = img.cpu().numpy()
        return bg

    def __call__(self, verts_list, triangles, bg, mesh_colors=np.array([[1, 0.6, 0.4]])):
        dev = _device(self.device)
        img = torch.from_numpy(np.ascontiguousarray(bg)).to(dev)            # a copy: the reference returns bg.copy()
        keys = torch.empty(img.shape[0] * img.shape[1], dtype=torch.int64, device=dev)
        same_topology = len(np.shape(triangles)) == 2
        if torch.is_tensor(verts_list):
            verts_dev = verts_list.to(dev, torch.float32).contiguous()
        else:
            verts_dev = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(v, np.float32) for v in verts_list]))).to(dev)
        light = torch.empty_like(verts_dev[0]) if len(verts_dev) else None
        for ind in range(len(verts_dev)):
            tri = triangles if same_topology else triangles[ind]
            topo = _topology(tri, verts_dev.shape[1], dev)
            self._render_dev(img, verts_dev[ind], topo, np.asarray(mesh_colors)[[ind % len(mesh_colors)]], keys, light)
        return img.cpu().numpy()
