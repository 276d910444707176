"""romp_amd.tracker against the fixture recorded from the reference's tracker classes (oracle/make_golden_tracker.py)."""
import os

import numpy as np
import pytest

from romp_amd import tracker as T

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'tracker_seq.npz')


def replay(points, scores, n_det):
    T.Track.last_id = 0
    trk = T.Tracker(det_thresh=0.12, low_conf_det_thresh=0.05, track_buffer=60, match_thresh=300, frame_rate=30)
    ids_all, inds_all, n_out, at = [], [], [], 0
    for n in n_det:
        pts, sc = points[at:at + n], scores[at:at + n]
        at += n
        if n == 0:
            n_out.append(0)
            continue
        ids, inds = trk.update(pts, sc)
        ids_all += ids
        inds_all += inds
        n_out.append(len(ids))
    return np.asarray(ids_all), np.asarray(inds_all), np.asarray(n_out)


@pytest.mark.parametrize('case', [0, 1, 2, 3])
def test_tracker_reproduces_reference_sequence(case):
    g = np.load(GOLD)
    k = 'c%d_' % case
    ids, inds, n_out = replay(g[k + 'points'], g[k + 'scores'], g[k + 'n_det'])
    assert np.array_equal(n_out, g[k + 'n_out'])
    assert np.array_equal(ids, g[k + 'ids'])
    assert np.array_equal(inds, g[k + 'inds'])


def test_assign_threshold_and_empty():
    pairs, ur, uc = T.assign(np.zeros((0, 3)), 10.)
    assert len(pairs) == 0 and ur == [] and uc == [0, 1, 2]
    cost = np.array([[1., 50.], [60., 2.], [70., 80.]])
    pairs, ur, uc = T.assign(cost, 300.)
    assert pairs.tolist() == [[0, 0], [1, 1]] and ur == [2] and uc == []
    # a pair dearer than the limit stays unassigned (padding costs limit/2 twice = limit)
    pairs, ur, uc = T.assign(np.array([[11.]]), 10.)
    assert len(pairs) == 0 and ur == [0] and uc == [0]


def test_new_tracks_confirmed_only_on_first_frame():
    T.Track.last_id = 0
    trk = T.Tracker()
    p = np.array([[10., 10., 100., 30.], [200., 50., 150., 30.]], np.float32)
    s = np.array([0.5, 0.6], np.float32)
    ids, inds = trk.update(p, s)
    assert ids == [1, 2] and inds == [0, 1]
    # a person appearing on frame 2 is reported from frame 3 on (byte_tracker_3dcenter.py:233-235)
    p2 = np.concatenate([p, [[100., 200., 80., 25.]]]).astype(np.float32)
    s2 = np.array([0.5, 0.6, 0.7], np.float32)
    ids, _ = trk.update(p2, s2)
    assert ids == [1, 2]
    ids, inds = trk.update(p2[::-1].copy(), s2[::-1].copy())
    assert ids == [1, 2, 3] and inds == [2, 1, 0]
