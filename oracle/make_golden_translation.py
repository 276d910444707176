"""Records tests/golden/translation_lsq.npz from the reference's estimate_translation (simple_romp/romp/utils.py:391-434).

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference):  python oracle/make_golden_translation.py

utils.py imports cv2, which is not installed here: it is stubbed with an empty module, so estimate_translation_cv2 raises and
estimate_translation takes its `except` branch -- the reference's own linear least squares, estimate_translation_np
(utils.py:347-389) -- which is the algorithm csrc/parse.hip translation_lsq_kernel implements.  Inputs: synthetic joints
(N, 71, 3) and their weak-perspective projections, used exactly as convert_cam_to_3d_trans2 does (post_parser.py:96-101).
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = '/root/reference/simple_romp/romp/utils.py'


def main():
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    spec = importlib.util.spec_from_file_location('ref_romp_utils', REF)
    U = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(U)
    rs = np.random.RandomState(7)
    N = 12
    joints = (rs.randn(N, 71, 3) * 0.35).astype(np.float32)
    cam = np.stack([rs.uniform(0.3, 1.4, N), rs.uniform(-0.6, 0.6, N), rs.uniform(-0.6, 0.6, N)], 1).astype(np.float32)
    pj2d = (joints[:, :, :2] * cam[:, None, 0:1] + cam[:, None, 1:]).astype(np.float32)        # batch_orth_proj, mode '2d'
    j24 = joints[:, :24]
    p24 = (pj2d[:, :24] + 1) * 256                                                             # post_parser.py:98
    trans = U.estimate_translation(j24, p24, focal_length=443.4, img_size=np.array([512, 512])).numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'translation_lsq.npz')
    np.savez_compressed(path, joints=joints, cam=cam, pj2d=pj2d, trans=trans)
    print('wrote', os.path.normpath(path), 'trans[0..2] =', trans[:3])


if __name__ == '__main__':
    main()
