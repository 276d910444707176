"""I/O helpers of the drop-in API -- mirror of the parts of simple_romp/romp/utils.py that the
ROMP class touches (img_preprocess :26-30, padding_image :16-24, convert_tensor2numpy :32-41,
ResultSaver :43-85, determine_device :734-739).  Host-side plumbing only; no hot-path maths.
"""
import os
import os.path as osp

import numpy as np
import torch


def padding_image(image):
    """utils.py:16-24 -- zero-pad to a centred square; returns the pad info vector."""
    h, w = image.shape[:2]
    side = max(h, w)
    pad_image = np.zeros((side, side, 3), dtype=np.uint8)
    top, left = int((side - h) // 2), int((side - w) // 2)
    bottom, right = int(top + h), int(left + w)
    pad_image[top:bottom, left:right] = image
    return pad_image, torch.Tensor([top, bottom, left, right, h, w])


def _cv_cubic_table(src, dst):
    """cv::resize INTER_CUBIC tables (OpenCV resize.cpp): per destination coordinate the first tap index and the four weights
    in 11-bit fixed point -- position in double, weights in float32 (A = -0.75), saturate_cast<short>(w * 2048)."""
    f32 = np.float32
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * (1.0 / (float(dst) / float(src))) - 0.5).astype(f32)
    s = np.floor(f).astype(np.int64)
    x = f - s.astype(f32)
    A, xp, xm = f32(-0.75), x + f32(1), f32(1) - x
    c0 = ((A * xp - f32(5) * A) * xp + f32(8) * A) * xp - f32(4) * A
    c1 = ((A + f32(2)) * x - (A + f32(3))) * x * x + f32(1)
    c2 = ((A + f32(2)) * xm - (A + f32(3))) * xm * xm + f32(1)
    c3 = f32(1) - c0 - c1 - c2
    w = np.rint(np.stack([c0, c1, c2, c3], 1).astype(f32) * f32(2048)).astype(np.int64)
    return s - 1, np.clip(w, -32768, 32767)


def resize_bicubic_u8(img, size):
    """cv2.resize(img, (size, size), interpolation=cv2.INTER_CUBIC) for uint8 images without OpenCV: the library's scalar
    fixed-point algorithm (replicated border, int32 horizontal pass, (v + 2^21) >> 22, saturate) -- the arithmetic the device
    kernel (csrc/post.hip) uses too.  OpenCV's SIMD builds round the vertical pass in float32 and may differ by one grey level
    on rare pixels (pre-processing parity with a real cv2 stays unpinned: the module is not installed here)."""
    h, w = img.shape[:2]
    xs, xa = _cv_cubic_table(w, size)
    ys, yb = _cv_cubic_table(h, size)
    src = img.astype(np.int64)
    hp = (src[:, np.clip(xs[:, None] + np.arange(4), 0, w - 1)] * xa[None, :, :, None]).sum(2)
    vp = (hp[np.clip(ys[:, None] + np.arange(4), 0, h - 1)] * yb[:, :, None, None]).sum(1)
    return np.clip((vp + (1 << 21)) >> 22, 0, 255).astype(np.uint8)


def img_preprocess(image, input_size=512):
    """utils.py:26-30: BGR->RGB, centred zero-pad to square, bicubic resize -> (1,S,S,3) float."""
    image = np.ascontiguousarray(image[:, :, ::-1])
    pad_image, image_pad_info = padding_image(image)
    try:
        import cv2
        resized = cv2.resize(pad_image, (input_size, input_size), interpolation=cv2.INTER_CUBIC)
    except ImportError:
        resized = resize_bicubic_u8(pad_image, input_size)
    return torch.from_numpy(resized)[None].float(), image_pad_info


def img_preprocess_device(image, device, input_size=512):
    """img_preprocess on the HIP device (csrc/post.hip): uploads the uint8 BGR frame (1/4 of the bytes of
    the float tensor) and pads / resizes / converts there.  -> ((1,S,S,3) float32 device tensor, pad info)."""
    import ctypes as C
    from . import lib as L
    lib = L.load()
    img = torch.from_numpy(np.ascontiguousarray(image)).to(device)
    assert img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3
    out = torch.empty(1, input_size, input_size, 3, device=device, dtype=torch.float32)
    pad = (C.c_float * 6)()
    with torch.cuda.device(device):
        L.check(lib.romp_preprocess(L.ptr(img), int(img.shape[0]), int(img.shape[1]), L.ptr(out), input_size, pad,
                                    L.stream_ptr(device)))
    return out, torch.Tensor(list(pad))


def convert_tensor2numpy(outputs, del_keys=('verts_camed', 'smpl_face', 'pj2d', 'verts_camed_org')):
    """utils.py:32-41.  The device tensors of one dtype travel in ONE device-to-host copy (a dozen small `.cpu()` calls cost
    more than the single-image network's post-processing); the arrays returned are views of that host block."""
    for key in del_keys:
        if key in outputs:
            del outputs[key]
    groups = {}
    for key, v in outputs.items():
        if isinstance(v, torch.Tensor):
            if v.is_cuda:
                groups.setdefault((v.dtype, v.device), []).append(key)
            else:
                outputs[key] = v.numpy()
    for (dtype, device), keys in groups.items():
        if len(keys) == 1:
            outputs[keys[0]] = outputs[keys[0]].cpu().numpy()
            continue
        flat = torch.cat([outputs[k].reshape(-1) for k in keys]).cpu().numpy()
        at = 0
        for k in keys:
            n = outputs[k].numel()
            outputs[k] = flat[at:at + n].reshape(tuple(outputs[k].shape))
            at += n
    return outputs


def determine_device(gpu_id):
    """utils.py:734-739."""
    return torch.device('cuda:{}'.format(gpu_id)) if gpu_id != -1 else torch.device('cpu')


class ResultSaver:
    """utils.py:43-85 (npz saving; image writing needs cv2 and is skipped without it)."""

    def __init__(self, mode='image', save_path=None, save_npz=True):
        self.is_dir = len(osp.splitext(save_path)[1]) == 0
        self.mode, self.save_path, self.save_npz = mode, save_path, save_npz
        self.save_dir = save_path if self.is_dir else osp.dirname(save_path)
        if self.mode in ['image', 'video']:
            os.makedirs(self.save_dir, exist_ok=True)
        if self.mode == 'video':
            self.frame_save_paths = []

    def __call__(self, outputs, input_path, prefix=None, img_ext='.png'):
        if self.mode == 'video' or self.is_dir:
            save_name = osp.basename(input_path)
            save_path = osp.join(self.save_dir, osp.splitext(save_name)[0]) + img_ext
        else:
            save_path = self.save_path
        if prefix is not None:
            save_path = osp.splitext(save_path)[0] + f'_{prefix}' + osp.splitext(save_path)[1]
        if outputs is not None and 'rendered_image' in outputs:
            try:
                import cv2
                cv2.imwrite(save_path, outputs.pop('rendered_image'))
            except ImportError:
                outputs.pop('rendered_image')
        if self.save_npz and outputs is not None:
            np.savez(osp.splitext(save_path)[0] + '.npz', results=outputs)
        if self.mode == 'video':
            self.frame_save_paths.append(save_path)


class WebcamVideoStream(object):
    """utils.py:118-146 (threaded cv2.VideoCapture reader); needs OpenCV."""

    def __init__(self, src=0):
        import cv2
        from threading import Thread
        self._Thread = Thread
        self.stream = cv2.VideoCapture(src)
        (self.grabbed, self.frame) = self.stream.read()
        self.stopped = False

    def start(self):
        self._Thread(target=self.update, args=(), daemon=True).start()
        return self

    def update(self):
        while not self.stopped:
            (self.grabbed, self.frame) = self.stream.read()

    def read(self):
        return self.frame

    def stop(self):
        self.stopped = True


def euclidean_distance(detection, tracked_object):
    """utils.py:272-273 (norfair distance function)."""
    return np.linalg.norm(detection.points - tracked_object.estimate)


def get_tracked_ids(detections, tracked_objects):
    """utils.py:275-280: id of the nearest tracked object for every detection."""
    ids = np.array([obj.id for obj in tracked_objects])
    tracked = np.array([obj.last_detection.points[0] for obj in tracked_objects])
    return [ids[np.argmin(np.linalg.norm(tracked - np.asarray(d.points)[None], axis=1))] for d in detections]
