"""Generate tests/golden/sim3dr_scene.npz with THE REFERENCE: simple_romp/vis_human/sim3drender/renderer.py
(imported by file path) driving the reference's C++ rasterizer compiled in place into oracle/_ref
(oracle/Makefile).  The Cython module the reference imports (`Sim3DR_Cython`, rasterize.pyx) is replaced by
a ctypes shim with the same two functions and argument order.  Run here (needs /root/reference):

    make -C oracle && python oracle/make_golden_sim3dr.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sim3dr_oracle as SO  # noqa: E402

REF_PY = '/root/reference/simple_romp/vis_human/sim3drender/renderer.py'


def load_reference_renderer():
    lib = SO.load_ref()
    assert lib is not None, 'build oracle/_ref first: make -C oracle'
    import ctypes as C
    shim = types.ModuleType('Sim3DR_Cython')

    def get_normal(normal, vertices, triangles, nver, ntri):                      # rasterize.pyx:56-63
        lib.ref_get_normal(SO._p(normal, C.c_float), SO._p(vertices, C.c_float), SO._p(triangles, C.c_int), nver, ntri)

    def rasterize(image, vertices, triangles, colors, depth_buffer, ntri, h, w, c, alpha=1, reverse=False):   # rasterize.pyx:114-127
        lib.ref_rasterize(SO._p(image, C.c_ubyte), SO._p(vertices, C.c_float), SO._p(triangles, C.c_int),
                          SO._p(colors, C.c_float), SO._p(depth_buffer, C.c_float), ntri, h, w, c, alpha, int(reverse))

    shim.get_normal, shim.rasterize = get_normal, rasterize
    sys.modules['Sim3DR_Cython'] = shim
    spec = importlib.util.spec_from_file_location('ref_sim3dr_renderer', REF_PY)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference_renderer()
    verts, tri, bg, colors = SO.make_scene(seed=0)
    image = ref.Sim3DR()(verts, tri, bg, mesh_colors=colors)
    normal0 = ref.get_normal(np.ascontiguousarray(verts[0]), tri)
    # the restatement must reproduce the reference bit for bit before the fixture is trusted
    mine = SO.render_meshes(verts, tri, bg, colors)
    assert np.array_equal(mine, image), 'numpy restatement differs from the reference in %d bytes' % (mine != image).sum()
    assert np.array_equal(SO.get_normal(verts[0], tri), normal0)
    out = os.path.join(ROOT, 'tests', 'golden', 'sim3dr_scene.npz')
    np.savez_compressed(out, verts=verts, triangles=tri, bg=bg, colors=colors, image=image, normal0=normal0)
    print('wrote', out, os.path.getsize(out), 'bytes; changed pixels:', int((image != bg).any(2).sum()))


if __name__ == '__main__':
    main()
