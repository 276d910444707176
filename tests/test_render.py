"""Sim3DR renderer.  CPU part: the numpy restatement (oracle/sim3dr_oracle.py) against the
fixture rendered by the reference itself (renderer.py + its C++ rasterizer compiled into oracle/_ref) and,
when oracle/_ref is present, against that library directly on further scenes.  GPU part: the HIP renderer
through the C ABI (romp_amd/renderer.py) against both.  Bar: BIT-EXACT uint8 images and float32 normals."""
import os

import numpy as np
import pytest

from oracle import sim3dr_oracle as SO


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, 'sim3dr_scene.npz'))


def _stress_scene(seed, h=96, w=128, nver=400, ntri=900):
    """Random soup: big overlapping triangles, degenerate ones (repeated vertex, zero area), vertices outside
    the image, exact depth ties (two copies of the same triangle) -- everything the z-test has to order."""
    rs = np.random.RandomState(seed)
    v = np.stack([rs.uniform(-20, w + 20, nver), rs.uniform(-20, h + 20, nver), rs.uniform(-50, 50, nver)], 1).astype(np.float32)
    v[:40, :2] = np.round(v[:40, :2])                       # vertices exactly on pixel centres / edges through them
    t = rs.randint(0, nver, (ntri, 3)).astype(np.int32)
    t[5] = t[4]                                             # identical triangles: equal depth everywhere, lower index wins
    t[7, 1] = t[7, 0]                                       # degenerate (repeated vertex)
    col = rs.uniform(0, 1, (nver, 3)).astype(np.float32)
    bg = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    return v, t, col, bg


def test_oracle_matches_reference_fixture(golden_dir):
    g = _golden(golden_dir)
    img = SO.render_meshes(g['verts'], g['triangles'], g['bg'], g['colors'])
    assert np.array_equal(img, g['image'])
    assert np.array_equal(SO.get_normal(g['verts'][0], g['triangles']), g['normal0'])
    assert (img != g['bg']).any(2).sum() > 5000            # the scene really covers pixels


@pytest.mark.skipif(SO.load_ref() is None, reason='oracle/_ref not built (make -C oracle; needs /root/reference)')
@pytest.mark.parametrize('seed', [1, 2])
def test_oracle_matches_compiled_reference(seed):
    v, t, col, bg = _stress_scene(seed)
    a = SO.rasterize(bg.copy(), v, t, col)
    b = SO.ref_rasterize(bg.copy(), v, t, col)
    assert np.array_equal(a, b)
    assert np.array_equal(SO.rasterize(bg.copy(), v, t, col, reverse=True), SO.ref_rasterize(bg.copy(), v, t, col, reverse=True))
    assert np.array_equal(SO.get_normal(v, t), SO.ref_get_normal(v, t))
    verts, tri, bg2, colors = SO.make_scene(seed=seed + 10, h=120, w=100, n=4)
    assert np.array_equal(SO.render_meshes(verts, tri, bg2, colors), SO.render_meshes(verts, tri, bg2, colors, use_ref=True))


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope='module')
def dev():
    import torch
    assert torch.cuda.is_available()
    from romp_amd import lib
    lib.load()
    return torch.device('cuda:0')


@pytest.mark.gpu
def test_hip_renderer_vs_reference_fixture(dev, golden_dir):
    from romp_amd.renderer import Sim3DR, get_normal
    g = _golden(golden_dir)
    bg = g['bg'].copy()
    img = Sim3DR()(g['verts'], g['triangles'], bg, mesh_colors=g['colors'])
    assert img.dtype == np.uint8 and img.shape == bg.shape and np.array_equal(bg, g['bg'])       # bg untouched, like bg.copy()
    nd = int((img != g['image']).sum())
    print('HIP Sim3DR vs reference-rendered fixture: differing bytes', nd, 'of', img.size)
    assert nd == 0
    assert np.array_equal(get_normal(g['verts'][0], g['triangles']), g['normal0'])


@pytest.mark.gpu
@pytest.mark.parametrize('seed', [1, 2, 3])
def test_hip_rasterize_vs_oracle(dev, seed):
    from romp_amd.renderer import rasterize, get_normal
    v, t, col, bg = _stress_scene(seed)
    ref = SO.ref_rasterize if SO.load_ref() is not None else SO.rasterize
    for reverse in (False, True):
        out = rasterize(v, t, col, bg=bg.copy(), reverse=reverse)
        assert np.array_equal(out, ref(bg.copy(), v, t, col, reverse=reverse))
    assert np.array_equal(get_normal(v, t), SO.get_normal(v, t))
    blank = rasterize(v, t, col, height=40, width=50, channel=3)
    assert np.array_equal(blank, ref(np.zeros((40, 50, 3), np.uint8), v, t, col))


@pytest.mark.gpu
def test_hip_renderer_smpl_sized_batch(dev):
    """12 meshes of SMPL size (6890 vertices, 13776 faces) on a 512x512 frame, lights moved: vs the oracle driver."""
    from romp_amd.renderer import Sim3DR
    rs = np.random.RandomState(5)
    base, tri = SO.ellipsoid_mesh(84, 82, [0, 0, 0], [1, 1, 1])        # 6808 vertices, 13612 faces
    verts = []
    for i in range(12):
        c = np.array([rs.uniform(60, 450), rs.uniform(60, 450), rs.uniform(-100, 100)])
        r = np.array([rs.uniform(30, 70), rs.uniform(60, 140), rs.uniform(20, 60)])
        verts.append((base * r[None] + c[None]).astype(np.float32))
    verts = np.stack(verts)
    bg = rs.randint(0, 256, (512, 512, 3)).astype(np.uint8)
    colors = rs.uniform(0.3, 1.0, (5, 3))
    cfg = dict(light_pos=(1, -2, -4), intensity_specular=0.2)
    img = Sim3DR(**cfg)(verts, tri, bg, mesh_colors=colors)
    ref = SO.render_meshes(verts, tri, bg, colors, cfg=cfg, use_ref=SO.load_ref() is not None)
    nd = int((img != ref).sum())
    print('12 SMPL-sized meshes: differing bytes', nd)
    assert nd == 0


@pytest.mark.gpu
def test_romp_render_mesh_end_to_end(dev):
    """romp.ROMP(settings with --render_mesh)(image)['rendered_image'] (main.py:170-172, vis_human/main.py:23-113):
    [frame | Sim3DR rendering].  The rendering must equal the oracle renderer run on the oracle's projection
    (post_parser.py:81-88, utils.py:309-315) of the SAME meshes and cameras -- bit for bit."""
    import torch
    import romp_amd
    from oracle import romp_oracle as O
    from romp_amd.vis import mesh_color_left2right
    settings = romp_amd.romp_settings(['--render_mesh'])
    settings.GPU, settings.center_thresh = 0, 1.25
    sd = O.make_romp_state_dict(0, center_bias=2.0)
    smpl = O.make_synthetic_smpl(0)
    rs = np.random.RandomState(3)
    tri, base_tri = None, None
    _, base_tri = SO.ellipsoid_mesh(84, 82, [0, 0, 0], [1, 1, 1])
    faces = np.zeros((13776, 3), np.int64)                                   # a real closed surface over the first 6808 vertices
    faces[:len(base_tri)] = base_tri
    smpl = dict(smpl, f=torch.from_numpy(faces).float())
    model = romp_amd.ROMP(settings, state_dict=sd, smpl_model=smpl)
    image = rs.randint(0, 256, (360, 640, 3)).astype(np.uint8)
    out = model(image)
    assert out is not None and out['rendered_image'].shape == (360, 1280, 3) and out['rendered_image'].dtype == np.uint8
    assert np.array_equal(out['rendered_image'][:, :640], image)
    assert 'verts_camed_org' not in out and 'smpl_face' not in out           # dropped like utils.py:32-41
    # oracle pipeline from the returned meshes / cameras
    verts, cam, ct = out['verts'], out['cam'], out['cam_trans']
    pad = max(image.shape[:2])
    top, left = (pad - 360) // 2, (pad - 640) // 2
    F = np.float32
    vc = np.concatenate([verts[:, :, :2] * cam[:, None, 0:1] + cam[:, None, 1:], verts[:, :, 2:]], -1).astype(F)
    vo = np.stack([(vc[..., 0] + F(1)) * F(pad) / F(2) - F(left), (vc[..., 1] + F(1)) * F(pad) / F(2) - F(top),
                   (vc[..., 2] + F(1)) * F(pad) / F(2)], -1).astype(F)
    order = torch.sort(torch.from_numpy(ct[:, 2]), descending=True).indices.numpy()
    vo = vo[order]
    vo[:, :, 2] *= -1
    colors = mesh_color_left2right(torch.from_numpy(ct))[order]
    ref = SO.render_meshes(vo, faces.astype(np.int32), image, colors, use_ref=SO.load_ref() is not None)
    nd = int((out['rendered_image'][:, 640:] != ref).sum())
    print('rendered_image vs oracle pipeline: differing bytes', nd, 'persons', len(order), 'painted px', int((ref != image).any(2).sum()))
    assert nd == 0 and (ref != image).any()
