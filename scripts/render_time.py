"""Timing of the Sim3DR renderer: 12 SMPL-sized meshes on a 512x512 frame, HIP vs the reference C++ (oracle/_ref)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import sim3dr_oracle as SO
from romp_amd.renderer import Sim3DR
rs = np.random.RandomState(5)
base, tri = SO.ellipsoid_mesh(84, 82, [0, 0, 0], [1, 1, 1])
verts = np.stack([(base * np.array([rs.uniform(30, 70), rs.uniform(60, 140), rs.uniform(20, 60)])[None]
                   + np.array([rs.uniform(60, 450), rs.uniform(60, 450), rs.uniform(-100, 100)])[None]).astype(np.float32) for _ in range(12)])
bg = rs.randint(0, 256, (512, 512, 3)).astype(np.uint8)
colors = rs.uniform(0.3, 1.0, (5, 3))
r = Sim3DR()
vd = torch.from_numpy(verts).cuda()
for _ in range(3):
    img = r(vd, tri, bg, mesh_colors=colors)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    img = r(vd, tri, bg, mesh_colors=colors)
torch.cuda.synchronize(); hip_ms = (time.perf_counter() - t0) / 20 * 1e3
t0 = time.perf_counter()
for _ in range(3):
    ref = SO.render_meshes(verts, tri, bg, colors, use_ref=True)
ref_ms = (time.perf_counter() - t0) / 3 * 1e3
print('12 meshes x %d faces on 512x512: HIP %.2f ms (incl. H2D of the frame, D2H of the image), reference C++ + numpy lighting %.1f ms, identical: %s'
      % (len(tri), hip_ms, ref_ms, np.array_equal(img, ref)))
