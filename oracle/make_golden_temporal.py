"""tests/golden/temporal_seq.npz from THE REFERENCE: simple_romp/romp/utils.py create_OneEuroFilter + smooth_results
(imported by file path with an empty cv2 stub) run over oracle.temporal_oracle.make_sequence.  Needs /root/reference."""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import temporal_oracle as TO  # noqa: E402


def main():
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    spec = importlib.util.spec_from_file_location('ref_romp_utils', '/root/reference/simple_romp/romp/utils.py')
    ru = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ru)
    out = {}
    for coeff in (3.0, 1.0):
        seq = TO.make_sequence(seed=int(coeff))
        filters, mine = ru.create_OneEuroFilter(coeff), TO.make_filters(coeff)
        T, Bt, Cm = [], [], []
        for th, be, ca in seq:
            t, b, c = ru.smooth_results(filters, th.clone(), be.clone(), ca.clone())
            t2, b2, c2 = TO.smooth(mine, th.clone(), be.clone(), ca.clone())
            for a, r in ((t2, t), (b2, b), (c2, c)):
                assert (a - r).abs().max() < 2e-6, (a - r).abs().max()
            T.append(t.numpy()); Bt.append(b.numpy()); Cm.append(c.numpy())
        out['thetas_%g' % coeff], out['betas_%g' % coeff], out['cam_%g' % coeff] = np.stack(T), np.stack(Bt), np.stack(Cm)
    path = os.path.join(ROOT, 'tests', 'golden', 'temporal_seq.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
