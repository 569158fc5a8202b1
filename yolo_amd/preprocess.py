"""Batched image pre-processing on the device (ydet_preprocess_u8): uint8 HWC images of any sizes -> one input
tensor, in one launch.  The geometry helpers restate what the reference computes on the host around its cv2 calls
(yolov5_official/utils.py:48-73 letterbox, data/transformers.py:456-461 base_transform); the pixels are produced by
the HIP kernel only."""
import ctypes as C

import numpy as np
import torch

from . import _ffi

_OUT_DT = {torch.float32: _ffi.F32, torch.bfloat16: _ffi.BF16, torch.uint8: _ffi.U8}


def letterbox_geometry(shape, new_shape=(640, 640), auto=True, scaleFill=False, scaleup=True, stride=32):
    """For an image of `shape` (h, w): (new_unpad (w, h), ratio (w, h), (dw, dh), top, bottom, left, right) exactly as
    the reference's letterbox derives them (yolov5_official/utils.py:48-73)."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    ratio = (r, r)
    new_unpad = (int(round(shape[1] * r)), int(round(shape[0] * r)))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    elif scaleFill:
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[1], new_shape[0])
        ratio = (new_shape[1] / shape[1], new_shape[0] / shape[0])
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_unpad, ratio, (dw, dh), top, bottom, left, right


def _as_device_u8(im, device):
    """uint8 HWC (3 channels) numpy array or tensor -> CUDA tensor with contiguous pixels."""
    if isinstance(im, np.ndarray):
        if im.dtype != np.uint8:
            raise TypeError(f"images must be uint8 (got {im.dtype})")
        im = torch.from_numpy(np.ascontiguousarray(im))
    if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
        raise TypeError(f"images must be uint8 [H, W, 3] (got {im.dtype} {tuple(im.shape)})")
    im = im.to(device, non_blocking=True)
    if im.stride(2) != 1 or im.stride(1) != 3:
        im = im.contiguous()
    return im


def preprocess_u8(images, placements, out_hw, device=None, pad_value=(114, 114, 114), swap_rb=False, scale_mode=1,
                  mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), dtype=torch.float32, channels_last=False):
    """images: list of uint8 [H, W, 3] arrays / tensors; placements: per image (rw, rh, left, top) -- the size it
    is resampled to and where it sits in the out_hw = (H, W) canvas.  scale_mode 0: raw values, 1: / 255,
    2: (/ 255 - mean) / std with mean / std per SOURCE channel.  Returns [N, 3, H, W] (or [N, H, W, 3])."""
    if device is None:
        device = next((im.device for im in images if isinstance(im, torch.Tensor) and im.is_cuda), None)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("preprocess_u8 runs on the GPU only (there is no CPU path here)")
    if dtype not in _OUT_DT:
        raise TypeError(f"unsupported output dtype {dtype}")
    lib = _ffi.lib()
    n = len(images)
    H, W = int(out_hw[0]), int(out_hw[1])
    out = torch.empty((n, H, W, 3) if channels_last else (n, 3, H, W), device=device, dtype=dtype)
    if n == 0:
        return out
    held = [_as_device_u8(im, device) for im in images]
    table = (_ffi.U8Image * n)()
    for i, (im, (rw, rh, left, top)) in enumerate(zip(held, placements)):
        h, w = int(im.shape[0]), int(im.shape[1])
        rw, rh, left, top = int(rw), int(rh), int(left), int(top)
        if rw <= 0 or rh <= 0 or h <= 0 or w <= 0:
            raise ValueError(f"image {i}: empty source or target ({h}x{w} -> {rh}x{rw})")
        if left < 0 or top < 0 or left + rw > W or top + rh > H:
            raise ValueError(f"image {i}: {rh}x{rw} at ({top},{left}) does not fit the {H}x{W} canvas")
        e = table[i]
        e.data, e.h, e.w, e.pitch = im.data_ptr(), h, w, int(im.stride(0))
        e.rh, e.rw, e.top, e.left = rh, rw, top, left
    raw = np.frombuffer(table, dtype=np.uint8).copy()
    dtable = torch.from_numpy(raw).to(device, non_blocking=True)
    d = _ffi.PreprocDesc()
    d.N, d.out_h, d.out_w = n, H, W
    d.out_dtype, d.layout, d.swap_rb, d.scale_mode = _OUT_DT[dtype], int(channels_last), int(bool(swap_rb)), int(scale_mode)
    for c in range(3):
        d.pad_value[c], d.mean[c], d.std[c] = int(pad_value[c]), float(mean[c]), float(std[c])
    with torch.cuda.device(device):
        _ffi.check(lib.ydet_preprocess_u8(dtable.data_ptr(), C.byref(d), out.data_ptr(), _ffi.current_stream()),
                   "ydet_preprocess_u8")
    # `held` / `dtable` were allocated (or already lived) on the launch stream: the caching allocator reuses their memory
    # in stream order, so the asynchronous kernel is done with them before anything else can get it
    return out


def letterbox_batch(images, new_shape=(640, 640), color=(114, 114, 114), auto=False, scaleFill=False, scaleup=True,
                    stride=32, bgr_to_rgb=True, scale=True, dtype=torch.float32, device=None):
    """The YOLOv5 input chain for a list of BGR uint8 HWC images as one launch: letterbox (reference
    yolov5_official/utils.py:40-75) -> [:, :, ::-1].transpose(2, 0, 1) -> / 255 (val.py:68-69).  All images must
    letterbox to the same canvas (always true with auto=False).  Returns (x [N, 3, H, W], ratios, pads) with
    ratios / pads as the reference's letterbox returns them per image (inputs of scale_coords' ratio_pad)."""
    geo = [letterbox_geometry(tuple(im.shape[:2]), new_shape, auto, scaleFill, scaleup, stride) for im in images]
    canvases = {(g[0][1] + g[3] + g[4], g[0][0] + g[5] + g[6]) for g in geo}
    if len(canvases) > 1:
        raise ValueError(f"images letterbox to different shapes {sorted(canvases)}: use auto=False or batch by shape")
    out_hw = canvases.pop() if canvases else (new_shape, new_shape) if isinstance(new_shape, int) else tuple(new_shape)
    x = preprocess_u8(images, [(g[0][0], g[0][1], g[5], g[3]) for g in geo], out_hw, device=device, pad_value=color,
                      swap_rb=bgr_to_rgb, scale_mode=1 if scale else 0, dtype=dtype if scale else torch.uint8)
    return x, [g[1] for g in geo], [g[2] for g in geo]
