"""Records tests/golden/tracker_seq.npz from the REFERENCE's own tracker classes.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference):  python oracle/make_golden_tracker.py

The reference's simple_romp/tracker/{byte_tracker_3dcenter,kalman_filter_3dcenter,matching,basetrack}.py are imported as
they are.  matching.py imports three third-party modules that are not installed here; they are stubbed:
  * cv2, cython_bbox: unused on this path (IoU matching is not called by the 3D-centre tracker);
  * lap (unpinned in simple_romp/setup.py): `lapjv(cost, extend_cost=True, cost_limit=t)` is restated from its published
    behaviour -- the rectangular cost padded to (n+m) x (n+m) with t/2 in the two off-diagonal blocks and 0 in the lower right
    block, solved exactly, rows/columns assigned to a padding cell reported as -1.  The stub solves the padded problem with
    scipy's exact solver.  The fixture therefore pins everything except lap's tie-breaking between equal-cost optima, which
    random real-valued tracking points do not produce.

The sequences: persons on smooth random trajectories in BEV's tracking space ((x, y) image coordinates 0..256, depth*30,
scale*64; bev/main.py:270-272) with births, deaths, occlusion gaps (a person missing for a few frames, or for longer than the
60-frame buffer), confidence dips below det_thresh (0.12) and below low_conf_det_thresh (0.05), and two persons crossing
closer than the duplicate distance (60).
"""
import os
import sys
import types

import numpy as np
from scipy.optimize import linear_sum_assignment

REF = '/root/reference/simple_romp'


def _install_stubs():
    lap = types.ModuleType('lap')

    def lapjv(cost, extend_cost=False, cost_limit=np.inf):
        n, m = cost.shape
        pad = np.zeros((n + m, n + m))
        pad[:n, :m] = cost
        pad[:n, m:] = cost_limit / 2.
        pad[n:, :m] = cost_limit / 2.
        r, c = linear_sum_assignment(pad)
        x, y = -np.ones(n, dtype=int), -np.ones(m, dtype=int)
        for i, j in zip(r, c):
            if i < n and j < m:
                x[i], y[j] = j, i
        return float(pad[r, c].sum()), x, y

    lap.lapjv = lapjv
    sys.modules['lap'] = lap
    sys.modules['cv2'] = types.ModuleType('cv2')
    cb = types.ModuleType('cython_bbox')
    cb.bbox_overlaps = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError)
    sys.modules['cython_bbox'] = cb


def make_sequence(seed, frames, persons):
    """-> list of (points (n, 4) float32, scores (n,) float32) per frame; detections are shuffled every frame."""
    rng = np.random.RandomState(seed)
    start = np.stack([rng.uniform(20, 236, persons), rng.uniform(20, 236, persons), rng.uniform(30, 400, persons),
                      rng.uniform(15, 60, persons)], 1)
    vel = rng.normal(0, 1.0, (persons, 4)) * np.array([3.0, 2.0, 3.0, 0.2])
    birth = rng.randint(0, frames // 3, persons)
    birth[:max(1, persons // 2)] = 0
    death = np.minimum(frames, birth + rng.randint(frames // 3, frames + 80, persons))
    gap_at = rng.randint(5, frames, persons)
    gap_len = rng.choice([0, 0, 2, 5, 9, 70], persons)
    dip_at = rng.randint(3, frames, persons)
    seq = []
    pos = start.copy()
    for f in range(frames):
        vel += rng.normal(0, 0.15, vel.shape) * np.array([1.0, 1.0, 1.0, 0.05])
        pos = pos + vel
        pts, sc = [], []
        for p in range(persons):
            if not (birth[p] <= f < death[p]) or gap_at[p] <= f < gap_at[p] + gap_len[p]:
                continue
            s = rng.uniform(0.2, 0.9)
            if dip_at[p] <= f < dip_at[p] + 3:
                s = rng.choice([0.08, 0.03, 0.11])
            pts.append(pos[p] + rng.normal(0, 0.8, 4))
            sc.append(s)
        if f % 37 == 36:                                  # a spurious one-frame detection
            pts.append(np.array([rng.uniform(0, 256), rng.uniform(0, 256), rng.uniform(30, 400), rng.uniform(15, 60)]))
            sc.append(rng.uniform(0.13, 0.3))
        order = rng.permutation(len(pts))
        seq.append((np.asarray(pts, np.float32).reshape(-1, 4)[order], np.asarray(sc, np.float32)[order]))
    return seq


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    from tracker.byte_tracker_3dcenter import Tracker
    from tracker.basetrack import BaseTrack
    out = {}
    cases = [(0, 160, 6), (1, 120, 12), (2, 200, 3), (3, 90, 25)]
    for ci, (seed, frames, persons) in enumerate(cases):
        BaseTrack._count = 0
        trk = Tracker(det_thresh=0.12, low_conf_det_thresh=0.05, track_buffer=60, match_thresh=300, frame_rate=30)
        seq = make_sequence(seed, frames, persons)
        n_det = np.array([len(p) for p, _ in seq], np.int32)
        ids_all, inds_all, n_out = [], [], []
        for pts, sc in seq:
            if len(pts) == 0:                            # BEV returns before the tracker when nobody is detected
                n_out.append(0)
                continue
            ids, inds = trk.update(pts, sc)
            ids_all += list(ids)
            inds_all += [int(i) for i in inds]
            n_out.append(len(ids))
        out['c%d_points' % ci] = np.concatenate([p for p, _ in seq], 0)
        out['c%d_scores' % ci] = np.concatenate([s for _, s in seq], 0)
        out['c%d_n_det' % ci] = n_det
        out['c%d_n_out' % ci] = np.asarray(n_out, np.int32)
        out['c%d_ids' % ci] = np.asarray(ids_all, np.int32)
        out['c%d_inds' % ci] = np.asarray(inds_all, np.int32)
        print('case', ci, 'frames', frames, 'detections', int(n_det.sum()), 'reported', len(ids_all), 'distinct ids',
              len(set(ids_all)), 'max id', max(ids_all))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'tracker_seq.npz')
    np.savez_compressed(path, **out)
    print('wrote', os.path.normpath(path), os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
