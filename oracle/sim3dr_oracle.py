"""CPU restatement of the Sim3DR renderer -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
path (romp_amd/renderer.py -> libromp_hip.so) never does.

Follows, in float32 and in the reference's operation order (so that the uint8 image is bit-identical):
  * vertex normals      simple_romp/vis_human/sim3drender/lib/rasterize_kernel.cpp:171-229  (_get_normal)
  * barycentric weights rasterize_kernel.cpp:56-85                                          (get_point_weight)
  * z-buffer rasterizer rasterize_kernel.cpp:233-300                                        (_rasterize)
  * lighting + driver   simple_romp/vis_human/sim3drender/renderer.py:64-133                (Sim3DR.render / __call__)

Pinned two ways: `load_ref()` binds oracle/_ref/libsim3dr_ref.so -- the reference's own C++ compiled in
place from /root/reference by oracle/Makefile -- and tests/golden/sim3dr_scene.npz holds a scene rendered
by the reference's renderer.py driving that library (oracle/make_golden_sim3dr.py).
"""
import ctypes as C
import os

import numpy as np

F = np.float32
_REF = None


def load_ref():
    """ctypes handle of the compiled reference rasterizer, or None if oracle/_ref has not been built."""
    global _REF
    if _REF is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'libsim3dr_ref.so')
        if not os.path.exists(path):
            return None
        lib = C.CDLL(path)
        fp, ip, up = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_ubyte)
        lib.ref_get_normal.argtypes = [fp, fp, ip, C.c_int, C.c_int]
        lib.ref_rasterize.argtypes = [up, fp, ip, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
        _REF = lib
    return _REF


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def ref_get_normal(vertices, triangles):
    """The reference C++ `_get_normal` (through oracle/_ref)."""
    lib = load_ref()
    v = np.ascontiguousarray(vertices, F)
    t = np.ascontiguousarray(triangles, np.int32)
    out = np.zeros_like(v)
    lib.ref_get_normal(_p(out, C.c_float), _p(v, C.c_float), _p(t, C.c_int), v.shape[0], t.shape[0])
    return out


def ref_rasterize(image, vertices, triangles, colors, reverse=False, alpha=1.0):
    """The reference C++ `_rasterize` (through oracle/_ref), in place on `image` like the reference."""
    lib = load_ref()
    h, w, c = image.shape
    v = np.ascontiguousarray(vertices, F)
    t = np.ascontiguousarray(triangles, np.int32)
    col = np.ascontiguousarray(colors, F)
    depth = np.zeros((h, w), F) - F(1e8)
    lib.ref_rasterize(_p(image, C.c_ubyte), _p(v, C.c_float), _p(t, C.c_int), _p(col, C.c_float), _p(depth, C.c_float),
                      t.shape[0], h, w, c, alpha, int(reverse))
    return image


# ---------------------------------------------------------------------------------------------- normals
def get_normal(vertices, triangles):
    """rasterize_kernel.cpp:171-229: un-normalised face normals accumulated onto their three vertices in
    triangle order, then normalised (|n| <= 0 -> 1e-6)."""
    v = np.asarray(vertices, F)
    t = np.asarray(triangles, np.int64)
    a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    e1, e2 = b - a, c - a
    tn = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1],
                   e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2],
                   e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], 1).astype(F)
    vn = np.zeros_like(v)
    np.add.at(vn, t.reshape(-1), np.repeat(tn, 3, axis=0))       # unbuffered, in (triangle, corner) order
    det = np.sqrt((vn[:, 0] * vn[:, 0] + vn[:, 1] * vn[:, 1]) + vn[:, 2] * vn[:, 2]).astype(F)
    det = np.where(det <= 0, F(1e-6), det)
    return (vn / det[:, None]).astype(F)


# ---------------------------------------------------------------------------------------------- raster
def _weights(px, py, p0, p1, p2):
    """rasterize_kernel.cpp:56-85 for arrays of pixel centres; returns (w0, w1, w2) float32."""
    v0x, v0y = p2[0] - p0[0], p2[1] - p0[1]
    v1x, v1y = p1[0] - p0[0], p1[1] - p0[1]
    v2x, v2y = px - p0[0], py - p0[1]
    d00 = v0x * v0x + v0y * v0y
    d01 = v0x * v1x + v0y * v1y
    d02 = v0x * v2x + v0y * v2y
    d11 = v1x * v1x + v1y * v1y
    d12 = v1x * v2x + v1y * v2y
    den = d00 * d11 - d01 * d01
    inv = F(0) if den == 0 else F(1) / den
    u = (d11 * d02 - d01 * d12) * inv
    vv = (d00 * d12 - d01 * d02) * inv
    return (F(1) - u) - vv, vv, u


def rasterize(image, vertices, triangles, colors, reverse=False):
    """rasterize_kernel.cpp:233-300 with alpha = 1 (the only value the reference's Python ever passes,
    rasterize.pyx:122): triangles in order, strict `>` z-test against a -1e8 buffer, colour truncated
    to uint8.  In place on `image` (H,W,C) uint8."""
    h, w, c = image.shape
    v = np.asarray(vertices, F)
    t = np.asarray(triangles, np.int64)
    col = np.asarray(colors, F)
    depth = np.zeros((h, w), F) - F(1e8)
    one, a255 = F(1), F(1) * F(255)
    with np.errstate(all='ignore'):
        for i0, i1, i2 in t:
            p0, p1, p2 = v[i0], v[i1], v[i2]
            x_min = max(int(np.ceil(min(p0[0], p1[0], p2[0]))), 0)
            x_max = min(int(np.floor(max(p0[0], p1[0], p2[0]))), w - 1)
            y_min = max(int(np.ceil(min(p0[1], p1[1], p2[1]))), 0)
            y_max = min(int(np.floor(max(p0[1], p1[1], p2[1]))), h - 1)
            if x_max < x_min or y_max < y_min:
                continue
            ys, xs = np.mgrid[y_min:y_max + 1, x_min:x_max + 1]
            w0, w1, w2 = _weights(xs.astype(F), ys.astype(F), p0, p1, p2)
            inside = (w2 >= 0) & (w1 >= 0) & (w0 > 0)
            pd = (w0 * p0[2] + w1 * p1[2]) + w2 * p2[2]
            hit = inside & (pd > depth[ys, xs])
            if not hit.any():
                continue
            yy, xx = ys[hit], xs[hit]
            for k in range(c):
                pc = (w0[hit] * col[i0, k] + w1[hit] * col[i1, k]) + w2[hit] * col[i2, k]
                row = (h - 1 - yy) if reverse else yy
                val = (one - one) * image[row, xx, k].astype(F) + a255 * pc
                image[row, xx, k] = val.astype(np.int64).astype(np.uint8)            # C cast: truncate toward zero
            depth[yy, xx] = pd[hit]
    return image


# ---------------------------------------------------------------------------------------------- lighting
DEFAULTS = dict(intensity_ambient=0.66, intensity_directional=0.36, intensity_specular=0.1, specular_exp=1,
                color_directional=(1, 1, 1), light_pos=(0, 0, -5), view_pos=(0, 0, 5))


def _unit(a):
    return a / np.sqrt(np.sum(a ** 2, axis=1))[:, None]


def vertex_light(vertices, normal, color, cfg=None):
    """renderer.py:77-110: ambient + clipped diffuse + specular per vertex, float32 (the ambient product is
    formed in float64 and rounded when it is added into the float32 buffer, as numpy does there)."""
    cfg = dict(DEFAULTS, **(cfg or {}))
    v = np.asarray(vertices, F)
    lp = np.array(cfg['light_pos'], F)[None, :]
    vp = np.array(cfg['view_pos'], F)[None, :]
    cd = np.array(cfg['color_directional'], F)[None, :]
    light = np.zeros_like(v)
    if cfg['intensity_ambient'] > 0:
        light += cfg['intensity_ambient'] * np.array(color)
    vn = v.copy()                                   # norm_vertices (renderer.py:19-24)
    vn -= vn.min(0)[None, :]
    vn /= vn.max()
    vn *= 2
    vn -= vn.max(0)[None, :] / 2
    if cfg['intensity_directional'] > 0:
        direction = _unit(lp - vn)
        cos = np.sum(normal * direction, axis=1)[:, None]
        light += cfg['intensity_directional'] * (cd * np.clip(cos, 0, 1))
        if cfg['intensity_specular'] > 0:
            to_view = _unit(vp - vn)
            refl = 2 * cos * normal - direction
            spe = np.sum((to_view * refl) ** cfg['specular_exp'], axis=1)[:, None]
            spe = np.where(cos != 0, np.clip(spe, 0, 1), np.zeros_like(spe))
            light += cfg['intensity_specular'] * cd * np.clip(spe, 0, 1)
    return np.clip(light, 0, 1)


def render_meshes(verts_list, triangles, bg, mesh_colors=np.array([[1, 0.6, 0.4]]), cfg=None, use_ref=False):
    """renderer.py:120-133 (`Sim3DR.__call__`): meshes painted one after the other onto a copy of `bg`, each
    with a fresh z-buffer.  use_ref: run the C++ reference for normals and rasterization."""
    out = bg.copy()
    tris = [triangles] * len(verts_list) if np.asarray(triangles).ndim == 2 else triangles
    for i, verts in enumerate(verts_list):
        verts = np.ascontiguousarray(verts, F)
        tri = np.ascontiguousarray(tris[i], np.int32)
        normal = ref_get_normal(verts, tri) if use_ref else get_normal(verts, tri)
        light = vertex_light(verts, normal, mesh_colors[[i % len(mesh_colors)]], cfg).astype(F)
        if use_ref:
            ref_rasterize(out, verts, tri, light)
        else:
            rasterize(out, verts, tri, light)
    return out


# ---------------------------------------------------------------------------------------------- test scenes
def ellipsoid_mesh(n_lat, n_lon, centre, radii, rot=0.0):
    """Closed triangle mesh of an ellipsoid in pixel coordinates: ((n_lat-1)*n_lon + 2 vertices)."""
    th = np.linspace(0, np.pi, n_lat + 1)[1:-1]
    ph = np.linspace(0, 2 * np.pi, n_lon, endpoint=False) + rot
    pts = [[0, 0, 1]] + [[np.sin(a) * np.cos(b), np.sin(a) * np.sin(b), np.cos(a)] for a in th for b in ph] + [[0, 0, -1]]
    v = np.array(pts, np.float64) * np.array(radii)[None] + np.array(centre)[None]
    idx = lambda i, j: 1 + i * n_lon + (j % n_lon)
    tris = []
    for j in range(n_lon):
        tris.append([0, idx(0, j), idx(0, j + 1)])
        tris.append([len(pts) - 1, idx(n_lat - 2, j + 1), idx(n_lat - 2, j)])
    for i in range(n_lat - 2):
        for j in range(n_lon):
            tris.append([idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)])
            tris.append([idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)])
    return v.astype(F), np.array(tris, np.int32)


def make_scene(seed=0, h=160, w=208, n=3, n_lat=14, n_lon=20):
    """n overlapping ellipsoids (same topology) over a random background; some vertices outside the image."""
    rs = np.random.RandomState(seed)
    bg = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    verts, tri = [], None
    for i in range(n):
        c = [rs.uniform(0.2, 0.8) * w, rs.uniform(0.2, 0.8) * h, rs.uniform(-40, 40)]
        r = [rs.uniform(0.15, 0.35) * w, rs.uniform(0.2, 0.45) * h, rs.uniform(20, 60)]
        v, tri = ellipsoid_mesh(n_lat, n_lon, c, r, rot=rs.uniform(0, 1))
        v += rs.normal(0, 0.75, v.shape).astype(F)                 # break symmetry / exact ties
        verts.append(v)
    colors = rs.uniform(0.2, 1.0, (n, 3))
    return np.stack(verts), tri, bg, colors
