"""CPU oracle of the reference's image pre-processing (simple_romp/romp/utils.py:16-30):

    image = cv2.cvtColor(image, cv2.COLOR_BGR2RGB)                      utils.py:27
    pad_image, image_pad_info = padding_image(image)                   utils.py:16-24 (centred zero pad to a square)
    cv2.resize(pad_image, (512, 512), interpolation=cv2.INTER_CUBIC)   utils.py:29

TEST INFRASTRUCTURE ONLY (imported by tests/ -- never by romp_amd/).

The arithmetic is OpenCV's, a third-party dependency (`opencv-python`, unpinned in simple_romp/setup.py:9) whose source is
not under /root/reference and whose module is not installed here.  This file restates the PUBLISHED algorithm of
cv::resize for CV_8UC3 + INTER_CUBIC (OpenCV 4.x modules/imgproc/src/resize.cpp: `resizeGeneric_<HResizeCubic<uchar,int,short>,
VResizeCubic<uchar,int,short,FixedPtCast<int,uchar,INTER_RESIZE_COEF_BITS*2>, ...>>` and `interpolateCubic`):

  * sampling: fx = (float)((dx + 0.5) * scale - 0.5) with scale = 1.0 / ((double)dst / src) in double; sx = floor(fx);
    fx -= sx (float);
  * cubic coefficients in FLOAT with A = -0.75:  c0 = ((A*(x+1) - 5A)*(x+1) + 8A)*(x+1) - 4A,  c1 = ((A+2)*x - (A+3))*x*x + 1,
    c2 = ((A+2)*(1-x) - (A+3))*(1-x)*(1-x) + 1,  c3 = 1 - c0 - c1 - c2;
  * fixed point: INTER_RESIZE_COEF_BITS = 11; every coefficient becomes saturate_cast<short>(c * 2048) (round half to even);
    the four need not sum to 2048 (no correction is applied for INTER_CUBIC);
  * horizontal pass: int32 D = sum_j S[clamp(sx - 1 + j)] * alpha_j (taps outside the image are clamped = replicated border);
  * vertical pass over rows clamp(sy - 1 + k): uchar = saturate((sum_k D_k * beta_k + (1 << 21)) >> 22).

PARITY UNPINNED for the last bit: OpenCV builds with SIMD (every shipped wheel) run the vertical pass of all but a row's
tail pixels in float32 (`VResizeCubicVec_32s8u`: round(S0*b0 + S1*b1 + S2*b2 + S3*b3) with b_k = beta_k / 2^22), which can
differ from the fixed-point formula above by one grey level on rare pixels; there is no cv2 here to generate a golden image.
What is pinned: the HIP kernel reproduces THIS restatement bit for bit (tests/test_bev_post.py).
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def padding_image(image):
    """utils.py:16-24."""
    h, w = image.shape[:2]
    side = max(h, w)
    pad = np.zeros((side, side, 3), dtype=np.uint8)
    top, left = (side - h) // 2, (side - w) // 2
    pad[top:top + h, left:left + w] = image
    return pad, np.array([top, top + h, left, left + w, h, w], np.float32)


def cubic_tables(src, dst):
    """Per destination coordinate: first tap index s - 1 (unclamped) and the four 11-bit fixed-point coefficients."""
    scale = 1.0 / (float(dst) / float(src))                             # double, as resize.cpp computes scale_x
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)                    # (float)((dx+0.5)*scale_x - 0.5)
    s = np.floor(f).astype(np.int64)
    x = (f - s.astype(np.float32)).astype(np.float32)
    A = np.float32(-0.75)
    one = np.float32(1.0)
    xp = (x + one).astype(np.float32)
    c0 = ((A * xp - np.float32(5) * A) * xp + np.float32(8) * A) * xp - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    xm = (one - x).astype(np.float32)
    c2 = ((A + np.float32(2)) * xm - (A + np.float32(3))) * xm * xm + one
    c3 = one - c0 - c1 - c2
    c = np.stack([c0, c1, c2, c3], 1).astype(np.float32)
    coef = np.clip(np.rint(c * np.float32(COEF_SCALE)), -32768, 32767).astype(np.int64)      # saturate_cast<short>(cvRound)
    return s - 1, coef


def resize_cubic_u8(img, size):
    """cv::resize(img, (size, size), INTER_CUBIC) for uint8 HxWxC, fixed-point path."""
    h, w = img.shape[:2]
    xs, xa = cubic_tables(w, size)
    ys, yb = cubic_tables(h, size)
    src = img.astype(np.int64)
    xi = np.clip(xs[:, None] + np.arange(4)[None], 0, w - 1)            # (size, 4) clamped taps
    hp = (src[:, xi] * xa[None, :, :, None]).sum(2)                     # (h, size, C) int32-range
    yi = np.clip(ys[:, None] + np.arange(4)[None], 0, h - 1)
    vp = (hp[yi] * yb[:, :, None, None]).sum(1)                         # (size, size, C)
    out = (vp + (1 << (2 * COEF_BITS - 1))) >> (2 * COEF_BITS)
    return np.clip(out, 0, 255).astype(np.uint8)


def img_preprocess(image_bgr, input_size=512):
    """utils.py:26-30 -> (float32 (1,S,S,3) RGB 0..255, pad info [top, bottom, left, right, h, w])."""
    rgb = np.ascontiguousarray(image_bgr[:, :, ::-1])
    pad, info = padding_image(rgb)
    return resize_cubic_u8(pad, input_size)[None].astype(np.float32), info
