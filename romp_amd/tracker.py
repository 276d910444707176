"""ByteTrack-3D association of BEV's video mode (``-t``).

Behavioural mirror of ``simple_romp/tracker/byte_tracker_3dcenter.py`` (``Tracker.update`` :21-147, ``get_tracked_ids_byte``
:149-160, ``STrack`` :207-294) together with its Kalman filter (``kalman_filter_3dcenter.py``: 8-state constant-velocity model
over (x, y, z, h)) and the two helpers of ``matching.py`` it calls (``linear_assignment`` :38-49, ``euclidean_distance``
:62-78) -- re-organised around one association routine and a batch Kalman filter, same decisions frame by frame
(tests/test_tracker.py replays a fixture recorded from the reference's own classes).

This is per-frame bookkeeping over <= 64 detections (a 64 x 64 cost matrix, 8 x 8 covariances): the reference runs it in
numpy on the host between two network calls, and so does this module; the device work of the temporal mode is the OneEuro
filtering (``temporal.py`` / csrc/temporal.hip).  Quirks are kept: the second association re-uses the HIGH-score detections
(:77-79), a brand-new track is `activated` only on frame 1 (:233-235), duplicate removal compares the first two coordinates.

The reference solves the assignment with ``lap.lapjv(cost, extend_cost=True, cost_limit=t)`` (third-party `lap`, not installed
here): the cost matrix padded to a square with t / 2 in the "stay unassigned" blocks, solved exactly.  `assign` builds the same
padded matrix for ``scipy.optimize.linear_sum_assignment`` (same optimum; equal-cost optima may be broken differently).
"""
import numpy as np
from scipy.optimize import linear_sum_assignment

NEW, TRACKED, LOST, REMOVED = 0, 1, 2, 3
POS_W, VEL_W = 1. / 20, 1. / 160                      # kalman_filter_3dcenter.py:33-34
F = np.eye(8)
F[:4, 4:] = np.eye(4)                                 # x += v  (dt = 1)
H = np.eye(4, 8)


def kf_start(z):
    """initiate (:36-66): mean = [z, 0], std 2/20 h on positions, 10/160 h on velocities."""
    z = np.asarray(z, dtype=np.float64)
    sd = np.r_[np.full(4, 2 * POS_W * z[3]), np.full(4, 10 * VEL_W * z[3])]
    return np.r_[z, np.zeros(4)], np.diag(sd ** 2)


def kf_predict_batch(means, covs):
    """multi_predict (:131-164) for (n, 8) means and (n, 8, 8) covariances."""
    q = np.concatenate([np.repeat((POS_W * means[:, 3:4]) ** 2, 4, 1), np.repeat((VEL_W * means[:, 3:4]) ** 2, 4, 1)], 1)
    means = means @ F.T
    covs = np.einsum('ij,njk,lk->nil', F, covs, F)
    covs[:, np.arange(8), np.arange(8)] += q
    return means, covs


def kf_correct(mean, cov, z):
    """update (:166-194): S = H P H^T + R, K = P H^T S^-1, Joseph-free form of the reference."""
    r = np.diag(np.full(4, (POS_W * mean[3]) ** 2))
    s = H @ cov @ H.T + r
    k = np.linalg.solve(s, (cov @ H.T).T).T
    return mean + (np.asarray(z, dtype=np.float64) - H @ mean) @ k.T, cov - k @ s @ k.T


def assign(cost, limit):
    """matching.linear_assignment: -> (pairs (k, 2), unmatched rows, unmatched columns)."""
    n, m = cost.shape
    if n == 0 or m == 0:
        return np.empty((0, 2), dtype=int), list(range(n)), list(range(m))
    padded = np.zeros((n + m, n + m))
    padded[:n, :m] = cost
    padded[:n, m:] = padded[n:, :m] = limit / 2.
    rows, cols = linear_sum_assignment(padded)
    pairs = [(r, c) for r, c in zip(rows, cols) if r < n and c < m]
    hit_r, hit_c = {r for r, _ in pairs}, {c for _, c in pairs}
    return np.asarray(pairs, dtype=int).reshape(-1, 2), [r for r in range(n) if r not in hit_r], [c for c in range(m) if c not in hit_c]


class Track(object):
    """One body-centre track (STrack): a detection until `begin` gives it an id and a filter."""
    __slots__ = ('z0', 'score', 'mean', 'cov', 'state', 'confirmed', 'tid', 'frame', 'born', 'hits')
    last_id = 0

    def __init__(self, z, score):
        self.z0, self.score = np.asarray(z, dtype=np.float32), score
        self.mean = self.cov = None
        self.state, self.confirmed, self.tid, self.frame, self.born, self.hits = NEW, False, 0, 0, 0, 0

    @property
    def pos(self):                                    # STrack.trans
        return self.z0.copy() if self.mean is None else self.mean[:4].copy()

    def begin(self, frame):                           # STrack.activate
        Track.last_id += 1
        self.tid = Track.last_id
        self.mean, self.cov = kf_start(self.z0)
        self.state, self.hits, self.frame, self.born = TRACKED, 0, frame, frame
        self.confirmed = self.confirmed or frame == 1

    def absorb(self, det, frame):                     # STrack.update (tracked) / re_activate (lost)
        self.mean, self.cov = kf_correct(self.mean, self.cov, det.pos)
        self.hits = self.hits + 1 if self.state == TRACKED else 0
        self.state, self.confirmed, self.frame, self.score = TRACKED, True, frame, det.score


def distances(a, b, dim=4):
    if not a or not b:
        return np.zeros((len(a), len(b)), dtype=np.float32)
    pa, pb = np.array([t.pos[:dim] for t in a]), np.array([t.pos[:dim] for t in b])
    return np.linalg.norm(pa[:, None] - pb[None], axis=2)


def union(first, second):                             # joint_stracks: order of `first`, then the new ids of `second`
    seen = {t.tid for t in first}
    out = list(first)
    for t in second:
        if t.tid not in seen:
            seen.add(t.tid)
            out.append(t)
    return out


def minus(first, second):                             # sub_stracks
    drop = {t.tid for t in second}
    by_id = {}
    for t in first:
        by_id[t.tid] = t
    return [t for tid, t in by_id.items() if tid not in drop]


def nearest_detection(points, tracks_out):
    """get_tracked_ids_byte: for every reported track the index of the closest detection of this frame."""
    ids, rows = [], []
    for rec in tracks_out:
        rows.append(int(np.argmin(np.linalg.norm(points - rec[None, :4], axis=1))))
        ids.append(int(rec[4]))
    return ids, rows


class Tracker(object):
    def __init__(self, det_thresh=0.12, low_conf_det_thresh=0.05, track_buffer=60, match_thresh=300, frame_rate=30):
        self.tracked_stracks, self.lost_stracks, self.removed_stracks = [], [], []
        self.frame_id = 0
        self.match_thresh, self.det_thresh, self.low_conf_det_thresh = match_thresh, det_thresh, low_conf_det_thresh
        self.max_time_lost = self.buffer_size = int(frame_rate / 30.0 * track_buffer)
        self.duplicat_dist_thresh = 60

    def _match(self, tracks, dets, limit, kept, revived, dim=4):
        """One association round: matched tracks absorb their detection; returns (unmatched track rows, unmatched det columns)."""
        pairs, free_t, free_d = assign(distances(tracks, dets, dim), limit)
        for r, c in pairs:
            (kept if tracks[r].state == TRACKED else revived).append(tracks[r])
            tracks[r].absorb(dets[c], self.frame_id)
        return free_t, free_d

    def update(self, trans3D, scores):
        """-> (track ids, row of `trans3D` each belongs to) for the confirmed tracks of this frame."""
        self.frame_id += 1
        strong = scores > self.det_thresh
        any_weak = bool(np.logical_and(scores > self.low_conf_det_thresh, scores < self.det_thresh).any())
        mk = lambda: [Track(z, s) for z, s in zip(trans3D[strong], scores[strong])]
        dets = mk()
        confirmed = [t for t in self.tracked_stracks if t.confirmed]
        tentative = [t for t in self.tracked_stracks if not t.confirmed]
        pool = union(confirmed, self.lost_stracks)
        if pool:                                       # STrack.multi_predict: lost tracks coast with vh = 0
            m = np.asarray([t.mean.copy() for t in pool])
            for i, t in enumerate(pool):
                if t.state != TRACKED:
                    m[i, 7] = 0
            m, c = kf_predict_batch(m, np.asarray([t.cov for t in pool]))
            for t, mi, ci in zip(pool, m, c):
                t.mean, t.cov = mi, ci
        kept, revived, newly_lost, dropped = [], [], [], []
        free_t, free_d = self._match(pool, dets, self.match_thresh, kept, revived)
        still = [pool[i] for i in free_t if pool[i].state == TRACKED]
        free_t2, _ = self._match(still, mk() if any_weak else [], self.match_thresh * 2, kept, revived)
        for i in free_t2:
            if still[i].state != LOST:
                still[i].state = LOST
                newly_lost.append(still[i])
        rest = [dets[i] for i in free_d]
        free_u, free_r = self._match(tentative, rest, self.match_thresh * 3, kept, kept)
        for i in free_u:
            tentative[i].state = REMOVED
            dropped.append(tentative[i])
        for i in free_r:
            if rest[i].score >= self.det_thresh:
                rest[i].begin(self.frame_id)
                kept.append(rest[i])
        for t in self.lost_stracks:
            if self.frame_id - t.frame > self.max_time_lost:
                t.state = REMOVED
                dropped.append(t)
        live = union(union([t for t in self.tracked_stracks if t.state == TRACKED], kept), revived)
        lost = minus(self.lost_stracks, live) + newly_lost
        lost = minus(lost, self.removed_stracks)
        self.removed_stracks.extend(dropped)
        # remove_duplicate_stracks: a live and a lost track closer than 60 in (x, y): the younger one goes
        d = distances(live, lost, dim=2)
        kill_live, kill_lost = set(), set()
        for a, b in zip(*np.where(d < self.duplicat_dist_thresh)):
            if live[a].frame - live[a].born > lost[b].frame - lost[b].born:
                kill_lost.add(b)
            else:
                kill_live.add(a)
        self.tracked_stracks = [t for i, t in enumerate(live) if i not in kill_live]
        self.lost_stracks = [t for i, t in enumerate(lost) if i not in kill_lost]
        out = np.array([np.r_[t.pos, t.tid] for t in self.tracked_stracks if t.confirmed])
        if len(out) == 0:
            return [], []
        return nearest_detection(trans3D, out)
