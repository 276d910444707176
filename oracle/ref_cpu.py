"""The REFERENCE's own CPU inference path, run from oracle/_ref/romp/*.pyc.

TEST / MEASUREMENT INFRASTRUCTURE (only tests/, __graft_entry__ and bench.py's cpu_baseline leg may import this).
`oracle/Makefile` byte-compiles simple_romp/romp/{model,smpl,post_parser,utils}.py from /root/reference into sourceless
modules under the git-ignored oracle/_ref/romp/ (built in the container by __graft_entry__.build(); the directory travels to
the GPU box like the built .so files).  This module imports them as the package `romp` -- with an empty `cv2` stub exactly
as oracle/make_golden.py does: OpenCV is only used by I/O / PnP helpers that are not on the timed path -- and exposes the
reference pipeline of simple_romp/romp/main.py:74-77,109-126 on CPU tensors:

    ROMPv1 (model.py:420-481)  ->  params_maps[:, 0] = 1.1 ** params_maps[:, 0] (main.py:113)
    -> parsing_outputs / CenterMap (post_parser.py:27-47,135-146)  ->  SMPL (smpl.py:62-108)
"""
import importlib.machinery
import importlib.util
import os
import sys
import tempfile
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, '_ref', 'romp')
MODS = ('model', 'smpl', 'utils', 'post_parser')


def available():
    return all(os.path.exists(os.path.join(REF_DIR, m + '.pyc')) for m in MODS)


def load():
    """-> dict of the reference's modules (imported once per process), or None when oracle/_ref/romp is not staged."""
    if not available():
        return None
    if 'romp.model' in sys.modules and getattr(sys.modules['romp.model'], '__romp_ref__', False):
        return {m: sys.modules['romp.' + m] for m in MODS}
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    pkg = types.ModuleType('romp')
    pkg.__path__ = [REF_DIR]
    sys.modules['romp'] = pkg
    mods = {}
    for name in MODS:
        path = os.path.join(REF_DIR, name + '.pyc')
        loader = importlib.machinery.SourcelessFileLoader('romp.' + name, path)
        spec = importlib.util.spec_from_loader('romp.' + name, loader, origin=path)
        m = importlib.util.module_from_spec(spec)
        m.__romp_ref__ = True
        sys.modules['romp.' + name] = m
        loader.exec_module(m)
        mods[name] = m
    return mods


BEV_DIR = os.path.join(HERE, '_ref', 'bev')
BEV_MODS = ('post_parser', 'model')


def bev_available():
    return available() and all(os.path.exists(os.path.join(BEV_DIR, m + '.pyc')) for m in BEV_MODS)


def load_bev():
    """-> dict of the reference's BEV modules (package `bev`, on top of the staged `romp` package), or None when not staged."""
    if not bev_available():
        return None
    load()
    if 'bev.model' in sys.modules and getattr(sys.modules['bev.model'], '__romp_ref__', False):
        return {m: sys.modules['bev.' + m] for m in BEV_MODS}
    pkg = types.ModuleType('bev')
    pkg.__path__ = [BEV_DIR]
    sys.modules['bev'] = pkg
    mods = {}
    for name in BEV_MODS:
        path = os.path.join(BEV_DIR, name + '.pyc')
        loader = importlib.machinery.SourcelessFileLoader('bev.' + name, path)
        spec = importlib.util.spec_from_loader('bev.' + name, loader, origin=path)
        m = importlib.util.module_from_spec(spec)
        m.__romp_ref__ = True
        sys.modules['bev.' + name] = m
        loader.exec_module(m)
        mods[name] = m
    return mods


class ReferenceBev:
    """BEVv1 of the reference (simple_romp/bev/model.py:104-250: backbone, coarse-to-fine 3-D localisation, 3-D centre-map parse,
    mesh parameter regression) on the CPU, holding the caller's (synthetic) weights."""

    def __init__(self, state_dict, center_thresh):
        ref = load_bev()
        assert ref is not None, 'oracle/_ref/bev is not staged (make -C oracle in the build container)'
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):                         # (the constructors print their settings)
            self.net = ref['model'].BEVv1(center_thresh=center_thresh).eval()
        sd = {k: v for k, v in state_dict.items() if k != 'coordmap_3d'}        # the module's own constant buffer (model.py:127-128)
        missing = self.net.load_state_dict(sd, strict=False)
        assert all(k.endswith('num_batches_tracked') or k == 'coordmap_3d' for k in missing.missing_keys) and not missing.unexpected_keys, missing

    @torch.no_grad()
    def __call__(self, images):
        """images (B,512,512,3) float 0..255 (CPU) -> BEVv1.forward's dict, or None."""
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            return self.net(images)


class ReferencePipeline:
    """ROMPv1 + CenterMap parser + SMPL of the reference, holding the caller's (synthetic) weights."""

    def __init__(self, state_dict, smpl_model, center_thresh):
        ref = load()
        assert ref is not None, 'oracle/_ref/romp is not staged (make -C oracle in the build container)'
        self.ref = ref
        self.net = ref['model'].ROMPv1().eval()
        missing = self.net.load_state_dict(state_dict, strict=False)
        assert all(k.endswith('num_batches_tracked') for k in missing.missing_keys) and not missing.unexpected_keys, missing
        self.parser = ref['post_parser'].CenterMap(conf_thresh=center_thresh)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, 'smpl.pth')
            torch.save(smpl_model, path)
            self.smpl = ref['smpl'].SMPL(path, model_type='smpl').eval()

    @torch.no_grad()
    def __call__(self, images):
        """images (B,512,512,3) float 0..255 (CPU) -> the reference's output dict with verts / joints, or None."""
        cm, pm = self.net(images)
        pm[:, 0] = torch.pow(1.1, pm[:, 0])                                     # main.py:113
        out = self.ref['post_parser'].parsing_outputs(cm, pm, self.parser)
        if out is None:
            return None
        v, j, _ = self.smpl(out['smpl_betas'], out['smpl_thetas'])
        out['verts'], out['joints'] = v, j
        return out


def reference_smpl(smpl_model):
    """The reference's own SMPL module (simple_romp/romp/smpl.py:38-108) on the CPU, holding the caller's (synthetic) model file
    contents -- bench.py --workload smpl times it as cpu_baseline kind "reference" (VERDICT r3 #8)."""
    ref = load()
    assert ref is not None, 'oracle/_ref/romp is not staged (make -C oracle in the build container)'
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, 'smpl.pth')
        torch.save(smpl_model, path)
        return ref['smpl'].SMPL(path, model_type='smpl').eval()
