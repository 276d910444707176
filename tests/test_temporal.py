"""Temporal smoothing: OneEuro filters of smooth_results.  CPU: the oracle restatement
against the fixture produced by the reference's own utils.py.  GPU: the device filter bank through the C ABI
against fixture and oracle.  Tolerance 2e-5 max-abs on thetas / betas / cam (float32 chains with sin/cos/atan2)."""
import os

import numpy as np
import pytest
import torch

from oracle import temporal_oracle as TO


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, 'temporal_seq.npz'))


@pytest.mark.parametrize('coeff', [3.0, 1.0])
def test_oracle_matches_reference_fixture(golden_dir, coeff):
    g = _golden(golden_dir)
    filters = TO.make_filters(coeff)
    for f, (th, be, ca) in enumerate(TO.make_sequence(seed=int(coeff))):
        t, b, c = TO.smooth(filters, th, be, ca)
        assert np.abs(t.numpy() - g['thetas_%g' % coeff][f]).max() < 5e-6
        assert np.abs(b.numpy() - g['betas_%g' % coeff][f]).max() < 5e-6
        assert np.abs(c.numpy() - g['cam_%g' % coeff][f]).max() < 5e-6
    # the filter really smooths: later frames differ from the raw input
    assert np.abs(g['cam_3'][5] - TO.make_sequence(3)[5][2].numpy()).max() > 1e-4


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from romp_amd import lib
    lib.load()
    return torch.device('cuda:0')


@pytest.mark.gpu
@pytest.mark.parametrize('coeff', [3.0, 1.0])
def test_hip_filter_vs_reference_fixture(dev, golden_dir, coeff):
    from romp_amd.temporal import OneEuroBank
    g = _golden(golden_dir)
    bank = OneEuroBank(dev, coeff, 10)
    worst = 0.0
    for f, (th, be, ca) in enumerate(TO.make_sequence(seed=int(coeff))):
        t, b, c = bank.smooth([7], th[None].to(dev).contiguous(), be[None].to(dev).contiguous(), ca[None].to(dev).contiguous())
        worst = max(worst, np.abs(t.cpu().numpy()[0] - g['thetas_%g' % coeff][f]).max(), np.abs(b.cpu().numpy()[0] - g['betas_%g' % coeff][f]).max(),
                    np.abs(c.cpu().numpy()[0] - g['cam_%g' % coeff][f]).max())
    print('OneEuro bank vs reference fixture, smooth_coeff %g: max-abs %.3e' % (coeff, worst))
    assert worst < 2e-5


@pytest.mark.gpu
def test_hip_filter_multi_track(dev):
    """Three persons (SMPL-A: 11 betas) whose rows change order between frames, one of them appearing late and one
    re-created after the bank overflowed: every track must follow its own oracle filter."""
    from romp_amd.temporal import OneEuroBank
    seqs = {tid: TO.make_sequence(seed=10 + tid, frames=8, n_betas=11) for tid in (3, 5, 9)}
    oracle = {tid: TO.make_filters(2.0) for tid in seqs}
    bank = OneEuroBank(dev, 2.0, 11, capacity=4)
    rs = np.random.RandomState(0)
    for f in range(8):
        ids = [t for t in (3, 5, 9) if not (t == 9 and f < 3)]
        rs.shuffle(ids)
        th = torch.stack([seqs[t][f][0] for t in ids]).to(dev).contiguous()
        be = torch.stack([seqs[t][f][1] for t in ids]).to(dev).contiguous()
        ca = torch.stack([seqs[t][f][2] for t in ids]).to(dev).contiguous()
        bank.smooth(ids, th, be, ca)
        for r, t in enumerate(ids):
            to, bo, co = TO.smooth(oracle[t], *seqs[t][f])
            assert (th[r].cpu() - to).abs().max() < 2e-5 and (be[r].cpu() - bo).abs().max() < 2e-5 and (ca[r].cpu() - co).abs().max() < 2e-5
    assert len(bank.slots) == 3
    for tid in (20, 21):                                    # overflow: capacity 4 -> the table is dropped and restarted
        bank.smooth([tid], torch.zeros(1, 72, device=dev), torch.zeros(1, 11, device=dev), torch.ones(1, 3, device=dev))
    assert len(bank.slots) <= 4 and 21 in bank.slots


@pytest.mark.gpu
def test_hip_filter_duplicate_track_ids(dev):
    """Two detections of one frame with the SAME track id (ROMP's nearest-track association can do that): the reference runs the
    track's filters twice, in detection order (main.py:141-157) -- so must the bank, instead of refusing the frame."""
    from romp_amd.temporal import OneEuroBank
    seq_a, seq_b = TO.make_sequence(seed=31, frames=4), TO.make_sequence(seed=32, frames=4)
    bank = OneEuroBank(dev, 3.0, 10)
    f7, f8 = TO.make_filters(3.0), TO.make_filters(3.0)
    for f in range(4):
        rows = [(7, seq_a[f]), (8, seq_b[f]), (7, seq_b[f])] if f % 2 else [(7, seq_a[f]), (8, seq_b[f])]
        th = torch.stack([r[1][0] for r in rows]).to(dev).contiguous()
        be = torch.stack([r[1][1] for r in rows]).to(dev).contiguous()
        ca = torch.stack([r[1][2] for r in rows]).to(dev).contiguous()
        bank.smooth([r[0] for r in rows], th, be, ca)
        for i, (tid, x) in enumerate(rows):                       # the oracle, applied sequentially in detection order
            to, bo, co = TO.smooth(f7 if tid == 7 else f8, *x)
            assert (th[i].cpu() - to).abs().max() < 2e-5 and (be[i].cpu() - bo).abs().max() < 2e-5 and (ca[i].cpu() - co).abs().max() < 2e-5
    assert len(bank.slots) == 2


@pytest.mark.gpu
def test_romp_temporal_show_largest(dev):
    """ROMP(settings: -t --show_largest): the largest person's thetas/betas/cam are filtered across frames
    (main.py:119-126); the meshes are computed from the smoothed parameters."""
    import romp_amd
    from oracle import romp_oracle as O
    sd = O.make_romp_state_dict(0, center_bias=2.0)
    smpl = O.make_synthetic_smpl(0)
    base = romp_amd.romp_settings([])
    base.GPU, base.center_thresh, base.host_preprocess = 0, 1.25, True
    raw_model = romp_amd.ROMP(base, state_dict=sd, smpl_model=smpl)
    s = romp_amd.romp_settings(['-t', '--show_largest'])
    s.GPU, s.center_thresh, s.host_preprocess = 0, 1.25, True
    assert s.temporal_optimize and s.show_largest
    model = romp_amd.ROMP(s, state_dict=sd, smpl_model=smpl)
    rs = np.random.RandomState(1)
    frame0 = rs.randint(0, 256, (360, 640, 3)).astype(np.uint8)
    filters = TO.make_filters(s.smooth_coeff)
    for f in range(4):
        frame = np.clip(frame0.astype(np.int32) + rs.randint(-6, 7, frame0.shape), 0, 255).astype(np.uint8)
        raw = raw_model(frame)
        out = model(frame)
        k = int(np.argmax(raw['cam'][:, 0]))
        t, b, c = TO.smooth(filters, torch.from_numpy(raw['smpl_thetas'][k]), torch.from_numpy(raw['smpl_betas'][k]), torch.from_numpy(raw['cam'][k]))
        assert out['smpl_thetas'].shape == (1, 72) and out['verts'].shape == (1, 6890, 3)
        e = max(np.abs(out['smpl_thetas'][0] - t.numpy()).max(), np.abs(out['smpl_betas'][0] - b.numpy()).max(), np.abs(out['cam'][0] - c.numpy()).max())
        print('frame %d: smoothed largest person vs oracle max-abs %.3e' % (f, e))
        assert e < 5e-5
        vo, _, _ = O.smpl_forward(smpl, out['smpl_betas'], out['smpl_thetas'])
        assert np.abs(out['verts'] - vo).max() < 1e-4
