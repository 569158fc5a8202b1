"""CPU restatement of the reference's image pre-processing -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference resamples with cv2.resize(..., INTER_LINEAR) on uint8 images (yolov5_official/utils.py:71,
data/transformers.py:457).  OpenCV is a third-party dependency that is neither in the reference tree nor
installed here, and the reference pins no version of it: `resize_linear_u8` restates the published generic
8-bit path of OpenCV's imgproc/resize.cpp and is PARITY UNPINNED for the resampled pixel values.  What the
reference itself defines -- the letterbox geometry, the border, the channel order and the normalisation
arithmetic -- is pinned by tests/golden/g10_preproc.npz (tools/gen_golden.py runs the reference's own
letterbox / base_transform with this resize standing in for cv2's).
"""
import numpy as np


def _axis(dst, src, zero_at_edge):
    """Per destination index: the two source indices and their 11-bit coefficients (resize.cpp: fx =
    (dx + 0.5) * scale_x - 0.5 with scale_x = 1 / (dsize / ssize) in double, cvFloor, cvRound(coef * 2048))."""
    scale = 1.0 / (float(dst) / float(src))
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if zero_at_edge:
        lo, hi = s < 0, s >= src - 1
        f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
        s = np.where(lo, 0, np.where(hi, src - 1, s))
        i0, i1 = s, np.minimum(s + 1, src - 1)
    else:
        i0, i1 = np.clip(s, 0, src - 1), np.clip(s + 1, 0, src - 1)
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    c1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return i0, i1, c0, c1


def resize_linear_u8(img, dsize):
    """cv2.resize(img, dsize=(width, height), interpolation=cv2.INTER_LINEAR) for a uint8 HWC image."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = img.shape[:2]
    if (dw, dh) == (sw, sh):
        return img.copy()
    if sw == 2 * dw and sh == 2 * dh:         # INTER_LINEAR at an exact 2x reduction is replaced by the box mean
        a = img.astype(np.int64)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, a0, a1 = _axis(dw, sw, True)
    y0, y1, b0, b1 = _axis(dh, sh, False)
    src = img.astype(np.int64)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]          # horizontal pass, [sh, dw, 3]
    s0, s1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def letterbox_geometry(shape, new_shape=(640, 640), auto=True, scaleFill=False, scaleup=True, stride=32):
    """yolov5_official/utils.py:48-73 without the pixels: (new_unpad (w, h), ratio, (dw, dh), top, bottom, left, right)."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    ratio = r, r
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    elif scaleFill:
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[1], new_shape[0])
        ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_unpad, ratio, (dw, dh), top, bottom, left, right


def letterbox(img, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleFill=False, scaleup=True, stride=32):
    """yolov5_official/utils.py:40-75."""
    new_unpad, ratio, pad, top, bottom, left, right = letterbox_geometry(img.shape[:2], new_shape, auto, scaleFill,
                                                                          scaleup, stride)
    if img.shape[:2][::-1] != new_unpad:
        img = resize_linear_u8(img, new_unpad)
    out = np.empty((img.shape[0] + top + bottom, img.shape[1] + left + right, 3), np.uint8)
    out[:] = np.asarray(color, np.uint8)
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out, ratio, pad


def v5_input(img_bgr, new_shape=(640, 640), **kw):
    """letterbox -> BGR to RGB, HWC to CHW -> float / 255 (yolov5_official/val.py:68-69 on the dataloader's output)."""
    lb = letterbox(img_bgr, new_shape, **kw)[0]
    chw = np.ascontiguousarray(lb[:, :, ::-1].transpose(2, 0, 1))
    return chw.astype(np.float32) / np.float32(255)


def base_transform(image, size, mean, std):
    """data/transformers.py:456-461."""
    x = resize_linear_u8(image, (size, size)).astype(np.float32)
    x /= 255.
    x -= np.asarray(mean, np.float32)
    x /= np.asarray(std, np.float32)
    return x


def family_a_input(image_bgr, size, mean=(0.406, 0.456, 0.485), std=(0.225, 0.224, 0.229)):
    """base_transform -> img[:, :, (2, 1, 0)] -> permute(2, 0, 1) (data/dataset.py:224-229)."""
    x = base_transform(image_bgr, size, mean, std)
    return np.ascontiguousarray(x[:, :, (2, 1, 0)].transpose(2, 0, 1))
