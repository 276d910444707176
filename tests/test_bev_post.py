"""BEV post-processing (projection, duplicate suppression, outlier removal) and device pre-processing.
CPU: oracle vs the reference-generated fixture bev_post.npz.  GPU: HIP kernels (csrc/post.hip) vs both."""
import os

import numpy as np
import pytest
import torch

from oracle import bev_post_oracle as PO


def _gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'bev_post.npz'))


def test_bev_post_oracle_vs_reference(golden_dir):
    g = _gold(golden_dir)
    r = PO.postprocess(g['joints'], g['cam'], g['pad'], 20.0, 1.6)
    np.testing.assert_allclose(r['pj2d_org'], g['pj2d_org'], rtol=1e-5, atol=2e-3)
    np.testing.assert_allclose(r['cam_trans'], g['cam_trans'], rtol=1e-5, atol=1e-6)
    assert np.nonzero(r['keep'])[0].tolist() == g['kept_final'].tolist()
    r2 = PO.postprocess(g['joints'], g['cam'], g['pad'], 20.0, 1e9)      # outlier removal disabled
    assert np.nonzero(r2['keep'])[0].tolist() == g['kept_after_nms'].tolist()


@pytest.mark.gpu
def test_bev_post_hip(golden_dir):
    from romp_amd import lib as L
    lib = L.load()
    dev = torch.device('cuda:0')
    g = _gold(golden_dir)
    # two images: the fixture persons + 3 more persons in a second image (N<3 after nms -> no outlier test)
    rs = np.random.RandomState(1)
    j2 = (0.25 * rs.randn(3, 71, 3)).astype(np.float32)
    c2 = np.array([[0.7, 0.1, 0.2], [0.69, 0.1, 0.2], [0.4, -0.5, 0.3]], np.float32)
    j2[1] = j2[0] + 0.001
    pad2 = np.array([0, 512, 64, 448, 512, 384], np.float32)
    joints = torch.from_numpy(np.concatenate([g['joints'], j2])).to(dev)
    cam = torch.from_numpy(np.concatenate([g['cam'], c2])).to(dev)
    N = cam.shape[0]
    offsets = torch.tensor([0, 9, 12], dtype=torch.int32, device=dev)
    pads = torch.from_numpy(np.stack([g['pad'], pad2])).float().to(dev)
    pj, pjo, tr = torch.empty(N, 71, 2, device=dev), torch.empty(N, 71, 2, device=dev), torch.empty(N, 3, device=dev)
    keep = torch.empty(N, dtype=torch.int32, device=dev)
    L.check(lib.romp_bev_postprocess(L.ptr(joints), L.ptr(cam), L.ptr(offsets), 2, L.ptr(pads), 20.0, 1.6, L.ptr(pj),
                                     L.ptr(pjo), L.ptr(tr), L.ptr(keep), L.stream_ptr(dev)))
    torch.cuda.synchronize()
    np.testing.assert_allclose(pjo[:9].cpu().numpy(), g['pj2d_org'], rtol=1e-5, atol=5e-3)
    np.testing.assert_allclose(tr[:9].cpu().numpy(), g['cam_trans'], rtol=1e-5, atol=1e-6)
    assert np.nonzero(keep[:9].cpu().numpy())[0].tolist() == g['kept_final'].tolist()
    r2 = PO.postprocess(j2, c2, pad2, 20.0, 1.6)
    assert keep[9:].cpu().numpy().astype(bool).tolist() == r2['keep'].tolist()
    np.testing.assert_allclose(pjo[9:].cpu().numpy(), r2['pj2d_org'], rtol=1e-5, atol=5e-3)
    np.testing.assert_allclose(pj[9:].cpu().numpy(), r2['pj2d'], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(360, 640), (720, 1280), (1080, 1920), (600, 400), (512, 512), (37, 53)])
def test_preprocess_hip_vs_oracle(shape):
    """Device img_preprocess (csrc/post.hip) vs the CPU oracle of the reference's pre-processing (oracle/cv_resize_oracle.py:
    cv2.cvtColor + padding_image + cv2.resize INTER_CUBIC restated in OpenCV's fixed-point arithmetic): BIT-EXACT, single frame
    and batched."""
    import ctypes as C
    from oracle import cv_resize_oracle as CV
    from romp_amd import lib as L
    from romp_amd.utils import img_preprocess_device
    rs = np.random.RandomState(shape[0])
    img = rs.randint(0, 256, shape + (3,)).astype(np.uint8)
    ref, pad_ref = CV.img_preprocess(img)
    dev = torch.device('cuda:0')
    out, pad = img_preprocess_device(img, dev)
    assert pad.tolist() == pad_ref.tolist()
    d = np.abs(out.cpu().numpy() - ref)
    print(shape, 'differing values', int((d > 0).sum()))
    assert d.max() == 0.0
    # batch of 3 different frames in one launch
    frames = rs.randint(0, 256, (3,) + shape + (3,)).astype(np.uint8)
    fd = torch.from_numpy(frames).to(dev)
    ob = torch.empty(3, 512, 512, 3, device=dev)
    pi = (C.c_float * 6)()
    L.check(L.load().romp_preprocess_batch(L.ptr(fd), 3, shape[0], shape[1], L.ptr(ob), 512, pi, L.stream_ptr(dev)))
    torch.cuda.synchronize()
    for b in range(3):
        rb, _ = CV.img_preprocess(frames[b])
        assert np.array_equal(ob[b].cpu().numpy(), rb[0])
    assert list(pi) == pad_ref.tolist()
