"""Temporal smoothing of the per-person estimates (video / webcam mode, ``-t``): mirror of
``simple_romp/romp/utils.py`` ``create_OneEuroFilter`` / ``smooth_results`` / ``check_filter_state`` (:246-269)
and of ``ROMP.temporal_optimization`` (``main.py:117-157``).  The filters run on the device
(``romp_oneeuro_smooth``, csrc/temporal.hip); this module only keeps the track -> state-slot table.
"""
import torch

from . import lib as L


class OneEuroBank(object):
    """Filter states of up to `capacity` tracks (per signal source) in one device buffer."""

    def __init__(self, device, smooth_coeff=3., n_betas=10, capacity=256):
        self.device, self.smooth_coeff, self.n_betas = torch.device(device), float(smooth_coeff), int(n_betas)
        self.lib = L.load()
        self.stride = self.lib.romp_oneeuro_state_floats(self.n_betas)
        self.state = torch.zeros(capacity, self.stride, device=self.device)
        self.slots = {}                                   # track id -> row of self.state

    def _slots_for(self, track_ids):
        """Rows of self.state for the tracks of ONE call.  The table is cleared (utils.py:253-254 drops all filters of a source
        that has grown too large) BEFORE any row of this call is handed out, so two tracks of a call never share a row."""
        if len(set(track_ids)) > self.state.shape[0]:
            raise L.RompHipError('OneEuroBank.smooth: %d tracks, capacity %d' % (len(track_ids), self.state.shape[0]))
        fresh = [t for t in track_ids if t not in self.slots]
        if len(self.slots) + len(fresh) > self.state.shape[0]:
            self.slots.clear()
            fresh = list(track_ids)
        if fresh:
            used = set(self.slots.values())
            free = (i for i in range(self.state.shape[0]) if i not in used)
            rows = [next(free) for _ in fresh]
            self.state[torch.as_tensor(rows, device=self.device)] = 0
            self.slots.update(zip(fresh, rows))
        return [self.slots[t] for t in track_ids]

    def smooth(self, track_ids, thetas, betas, cam):
        """In place on thetas (N,72), betas (N,nb), cam (N,3) (device float32, contiguous); returns them.
        Two detections of a frame may carry the SAME track id (ROMP's get_tracked_ids gives each detection the id of its nearest
        tracked object, utils.py:493-533); the reference then runs that track's filters twice, in detection order
        (main.py:141-157).  The kernel updates every row of a call from the same previous state, so the rounds below do the same:
        first occurrences in one batched launch, second occurrences in the next (on the state the first just wrote), ..."""
        assert thetas.is_contiguous() and betas.is_contiguous() and cam.is_contiguous() and thetas.dtype == torch.float32
        track_ids = list(track_ids)
        unique = list(dict.fromkeys(track_ids))
        rows_of = dict(zip(unique, self._slots_for(unique)))
        seen, rounds = {}, []
        for i, t in enumerate(track_ids):
            k = seen.get(t, 0)
            seen[t] = k + 1
            while len(rounds) <= k:
                rounds.append([])
            rounds[k].append(i)
        with torch.cuda.device(self.device):
            for idx in rounds:
                slots = torch.tensor([rows_of[track_ids[i]] for i in idx], dtype=torch.int32, device=self.device)
                if len(idx) == len(track_ids):                    # the common case: no duplicates, filter in place
                    th, be, ca = thetas, betas, cam
                else:
                    sel = torch.tensor(idx, dtype=torch.int64, device=self.device)
                    th, be, ca = thetas[sel].contiguous(), betas[sel].contiguous(), cam[sel].contiguous()
                L.check(self.lib.romp_oneeuro_smooth(L.ptr(self.state), L.ptr(slots), len(idx), self.n_betas, self.smooth_coeff,
                                                     L.ptr(th), L.ptr(be), L.ptr(ca), L.stream_ptr(self.device)))
                if th is not thetas:
                    thetas[sel], betas[sel], cam[sel] = th, be, ca
        return thetas, betas, cam
