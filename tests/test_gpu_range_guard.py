"""GPU tests of the default-on range guard: the reference's network is float32 and has no activation range to leave
(simple_romp/romp/main.py:106-115); the f16x2 kernels clamp beyond 65504 / 2^act_shift.  No call of the API may return maps that
went through a clamp: every flow reads the net's saturation counter back with its detection count (romp_parse_watch) and, if it
moved, answers with the exact-f32 program of the same weights.

Recipes for leaving the range: a pre-processed float batch far outside 0..255 on a CALIBRATED net (forward_batch / forward_chunks),
and -- the blown-up BatchNorm of test_net_saturation_is_observable with calibration off -- a uint8 frame through ROMP(settings)(image).
Bar: the API result equals that of a conv_math='f32' model on the same input within 1e-5 (it is the same float32 program), with
identical detections; calls that stay in range are NOT re-run and still match the oracle-gated f16x2 numbers."""
import warnings

import numpy as np
import pytest
import torch

from oracle import romp_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    from romp_amd import lib
    lib.load()
    return torch.device('cuda:0')


def _model(sd, smpl, math, max_batch, thresh=1.3, **kw):
    import romp_amd
    s = romp_amd.romp_settings([])
    s.GPU, s.center_thresh, s.max_batch, s.conv_math = 0, thresh, max_batch, math
    for k, v in kw.items():
        setattr(s, k, v)
    return romp_amd.ROMP(s, state_dict=sd, smpl_model=smpl)


def _same(out, bids, ref, rbids, tol=1e-5):
    assert (out is None) == (ref is None)
    if out is None:
        return
    assert torch.equal(bids, rbids), 'detections differ'
    assert torch.equal(out['center_preds'], ref['center_preds'])
    for k in ('cam', 'smpl_thetas', 'smpl_betas', 'verts', 'joints'):
        e = float((out[k] - ref[k]).abs().max())
        assert e <= tol * max(1.0, float(ref[k].abs().max())), (k, e)


def test_parse_watch_rides_with_the_counts(dev):
    """romp_parse_watch: the watched device word comes back in the same call as the counts, in both forms (count_host given /
    asynchronous: left in workspace[B*(2K+2)])."""
    import ctypes as C
    from romp_amd import lib as L
    from romp_amd.post_parser import _parse
    g = torch.Generator().manual_seed(5)
    cm = torch.rand(3, 64, 64, generator=g).to(dev)
    pm = torch.randn(3, 64, 64, 145, generator=g).to(dev)
    word = torch.tensor([123456789], dtype=torch.int32, device=dev)
    r, seen = _parse(cm, pm, 0.995, 64, watch=word.data_ptr())
    r0, seen0 = _parse(cm, pm, 0.995, 64)
    assert seen == 123456789 and seen0 is None
    assert r is not None and all(torch.equal(r[k], r0[k]) for k in r)


def test_forward_batch_out_of_range_input_is_answered_in_float32(dev):
    sd, smpl = O.make_romp_state_dict(0, center_bias=2.0), O.make_synthetic_smpl(0)
    m = _model(sd, smpl, 'f16x2', 4)
    m32 = _model(sd, smpl, 'f32', 4)
    img = O.make_images(4, seed=3).to(dev)
    assert m.range_guard.enabled and not m32.range_guard.enabled
    out, bids = m.forward_batch(img)                      # in range: the f16x2 program's answer, no re-run
    assert m.range_guard.reruns == 0 and out is not None
    ref, rbids = m32.forward_batch(img)
    assert torch.equal(bids, rbids)
    big = img * 2000.0                                    # far beyond the calibrated range: the stem's H2 output clamps at 4094
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        out, bids = m.forward_batch(big)
    assert m.range_guard.reruns == 1
    assert any('calibrated range' in str(x.message) for x in w), 'the first re-run warns'
    assert 'clamping ops' in ' '.join(str(x.message) for x in w)
    ref, rbids = m32.forward_batch(big)
    _same(out, bids, ref, rbids)
    # what the guard prevents: the f16x2 program's own maps on that input are NOT the float32 ones
    c16, p16 = m.model.forward_nhwc(big)
    c32, p32 = m32.model.forward_nhwc(big)
    assert float((p16 - p32).abs().max()) > 1e-2 * float(p32.abs().max())
    m.range_guard.resync()
    out, bids = m.forward_batch(img)                      # back in range: no further re-run
    assert m.range_guard.reruns == 1 and out is not None


def test_forward_chunks_charges_the_right_chunks(dev):
    """Pipelined: the network of chunk i+1 is in flight when chunk i's counter is read.  The bad chunk (and, conservatively, its
    successor) are re-run; every chunk's result equals the float32 model's; the chunks before it are not re-run."""
    sd, smpl = O.make_romp_state_dict(0, center_bias=2.0), O.make_synthetic_smpl(0)
    m = _model(sd, smpl, 'f16x2', 2)
    m32 = _model(sd, smpl, 'f32', 2)
    img = O.make_images(10, seed=4).to(dev)
    img[4:6] *= 2000.0                                    # chunk 2 of 5 leaves the range
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        got = [(o, b, c0, m.range_guard.reruns) for o, b, c0 in m.forward_chunks(img, 2)]
    after = [g[3] for g in got]
    rerun = [after[0]] + [after[i] - after[i - 1] for i in range(1, 5)]
    # chunk 2 clamps while chunk 1 is being parsed or later: the re-run set is {1,2}, {2,3} or {1,2,3}; never 0 or 4
    assert rerun[0] == 0 and rerun[2] == 1 and rerun[4] == 0 and sum(rerun) in (2, 3), rerun
    for i, ((o, b, c0, _), (ro, rb, rc0)) in enumerate(zip(got, m32.forward_chunks(img, 2))):
        assert c0 == rc0 and (o is None) == (ro is None)
        if o is None:
            continue
        assert torch.equal(b, rb) and torch.equal(o['center_preds'], ro['center_preds']), 'chunk %d: detections differ' % i
        tol = 1e-5 if rerun[i] else 1e-3          # a re-run chunk IS the float32 program; a clean one is the f16x2 program (1e-4-class maps)
        for k in ('cam', 'smpl_betas', 'verts'):
            assert float((o[k] - ro[k]).abs().max()) <= tol * max(1.0, float(ro[k].abs().max())), (i, k)


def test_api_single_frame_blown_up_net(dev):
    """ROMP(settings)(frame) on an uncalibrated net with a blown-up BatchNorm (everything downstream ~3e4 x larger): both the
    latency-arranged flow and the standard flow return the float32 program's result."""
    sd, smpl = O.make_romp_state_dict(0, center_bias=2.0), O.make_synthetic_smpl(0)
    big = {k: v.clone() for k, v in sd.items()}
    key_w = [k for k in big if k.endswith('bn2.weight') and k.count('.') <= 2][0]
    big[key_w] *= 3e4
    big[key_w.replace('weight', 'bias')] *= 3e4
    frame = np.random.RandomState(1).randint(0, 256, (480, 640, 3)).astype(np.uint8)
    m = _model(big, smpl, 'f16x2', 1, thresh=0.25, no_calibrate=True)
    m32 = _model(big, smpl, 'f32', 1, thresh=0.25)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fast = m(frame)
        assert m.range_guard.reruns == 1
        m.fast_single = False
        std = m(frame)
        assert m.range_guard.reruns == 2
    ref = m32(frame)
    assert (ref is None) == (fast is None) == (std is None)
    if ref is not None:
        for got in (fast, std):
            assert np.array_equal(got['center_preds'], ref['center_preds'])
            for k in ('cam', 'smpl_thetas', 'verts', 'joints', 'pj2d_org'):
                # (maps 3e4 x their usual size: 1.1 ** scale overflows float32 in the reference arithmetic itself -- the same
                # non-finite entries on both sides, everything finite within 1e-5)
                fin = np.isfinite(ref[k])
                assert np.array_equal(fin, np.isfinite(got[k])), k
                if fin.any():
                    assert np.abs(got[k][fin] - ref[k][fin]).max() <= 1e-5 * max(1.0, np.abs(ref[k][fin]).max()), k


def test_plan_file_net_fails_loudly_instead_of_returning_clamped_maps(dev, tmp_path):
    """A net loaded from a plan file has no float32 program to fall back to: leaving the range raises, it never passes silently."""
    import romp_amd
    from romp_amd import export, lib as L
    sd, smpl = O.make_romp_state_dict(0, center_bias=2.0), O.make_synthetic_smpl(0)
    m = _model(sd, smpl, 'f16x2', 2)
    path = str(tmp_path / 'romp.plan')
    export.save_plan(m.model, path)
    s = romp_amd.romp_settings([])
    s.GPU, s.center_thresh, s.max_batch, s.plan_path = 0, 1.3, 2, path
    p = romp_amd.ROMP(s, smpl_model=smpl)
    img = O.make_images(2, seed=3).to(dev)
    out, bids = p.forward_batch(img)
    assert out is not None and p.range_guard.reruns == 0
    with pytest.raises(L.RompHipError, match='calibrated range'):
        p.forward_batch(img * 2000.0)


def test_bev_guard(dev):
    from oracle import bev_oracle as BO
    from romp_amd import bev
    from romp_amd import synthetic as S
    sd = S.make_bev_state_dict(0)
    smpla, smil = S.make_smpl_model(0, 11), S.make_smpl_model(5, 10)

    def make(math):
        s = bev.bev_settings([])
        s.GPU, s.max_batch, s.conv_math = 0, 2, math
        return bev.BEV(s, state_dict=sd, smpla_model=smpla, smil_model=smil)
    m, m32 = make('f16x2'), make('f32')
    img = S.make_images(2, seed=4, device=dev)
    m.model.centermap_parser.conf_thresh = m32.model.centermap_parser.conf_thresh = 0.05
    a = m.model(img)
    assert m.model.range_guard.reruns == 0
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        a = m.model(img * 2000.0)
    assert m.model.range_guard.reruns == 1
    b = m32.model(img * 2000.0)
    assert (a is None) == (b is None)
    if a is not None:
        assert torch.equal(a['pred_czyxs'], b['pred_czyxs'])
        assert float((a['params_pred'] - b['params_pred']).abs().max()) <= 1e-5 * max(1.0, float(b['params_pred'].abs().max()))
