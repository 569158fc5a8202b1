"""Tensor-level wrappers of the decode / NMS / IoU entry points of libydet.so.

Inputs and outputs are CUDA tensors; everything is enqueued on torch's current stream and nothing
here synchronises with the host.
"""
import ctypes as C

import torch

from . import _ffi


def _dt(t):
    if t.dtype == torch.float32:
        return _ffi.F32
    if t.dtype == torch.bfloat16:
        return _ffi.BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _need_cuda(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("yolo_amd GPU operator called with a CPU tensor (there is no CPU path here)")


def decode_v5(heads, strides, anchors_px, na, no, grid_transposed=False, row_max=True):
    """heads: NHWC maps [N, ny, nx, >=na*no] (fp32 or bf16, last dim contiguous), one per level.
    strides: per-level float; anchors_px: per-level list of 2*na pixel sizes.
    Returns z [N, sum(na*ny*nx), no] fp32 (yolov5_official/model.py:56-81).
    With row_max, the kernel also emits per box max_c(cls_c * obj); it rides on the result as `z._ydet_row_max`
    (tensor, version of z) and lets nms_v5 skip its scan of z while z stays unmodified."""
    _need_cuda(*heads)
    lib = _ffi.lib()
    nl = len(heads)
    N = heads[0].shape[0]
    levels = (_ffi.V5Level * nl)()
    rows = 0
    for l, h in enumerate(heads):
        assert h.dim() == 4 and h.stride(3) == 1 and h.shape[0] == N
        assert h.stride(2) == h.stride(1) // h.shape[2] and h.stride(0) == h.stride(1) * h.shape[1], "head not NHWC-dense"
        L = levels[l]
        L.head, L.ny, L.nx, L.ld = h.data_ptr(), h.shape[1], h.shape[2], h.stride(2)
        L.stride = float(strides[l])
        for i, v in enumerate(anchors_px[l]):
            L.anchor_wh[i] = float(v)
        rows += na * h.shape[1] * h.shape[2]
    out = torch.empty(N, rows, no, device=heads[0].device, dtype=torch.float32)
    rm = torch.empty(N, rows, device=out.device, dtype=torch.float32) if row_max and no > 5 else None
    with torch.cuda.device(out.device):
        _ffi.check(lib.ydet_decode_v5(levels, nl, N, na, no, int(grid_transposed), _dt(heads[0]), out.data_ptr(),
                                      rm.data_ptr() if rm is not None else None, _ffi.current_stream()),
                   "ydet_decode_v5")
    if rm is not None:
        out._ydet_row_max = (rm, out._version)
    return out


def row_max_of(pred):
    """The row maxima decode_v5 attached to `pred`, if `pred` is still that very tensor, unmodified."""
    hint = getattr(pred, "_ydet_row_max", None)
    if hint is None:
        return None
    rm, version = hint
    if version != pred._version or rm.shape != pred.shape[:2] or rm.device != pred.device:
        return None
    return rm


def decode_v2v3(heads, anchors, A, nc, mode, input_size, mult=32.0, grid_transposed=True):
    """heads: NHWC maps [N, H, W, >=A*(5+nc)] fine -> coarse; anchors: per-level list of 2*A values.
    mode 0 = utils/utils.py xywh_to_xxyy ("v2"), 1 = xywh_to_xxyyv3.
    Returns (boxes [N, n, 4] normalised+clamped xyxy, scores [N, n, nc]) fp32."""
    _need_cuda(*heads)
    lib = _ffi.lib()
    nl = len(heads)
    N = heads[0].shape[0]
    levels = (_ffi.FALevel * nl)()
    n = 0
    for l, h in enumerate(heads):
        assert h.dim() == 4 and h.stride(3) == 1 and h.shape[0] == N
        assert h.stride(2) == h.stride(1) // h.shape[2] and h.stride(0) == h.stride(1) * h.shape[1], "head not NHWC-dense"
        L = levels[l]
        L.head, L.H, L.W, L.ld = h.data_ptr(), h.shape[1], h.shape[2], h.stride(2)
        for i, v in enumerate(anchors[l]):
            L.anchor_wh[i] = float(v)
        n += A * h.shape[1] * h.shape[2]
    dev = heads[0].device
    boxes = torch.empty(N, n, 4, device=dev, dtype=torch.float32)
    scores = torch.empty(N, n, nc, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _ffi.check(lib.ydet_decode_v2v3(levels, nl, N, A, nc, int(mode), float(mult), float(input_size),
                                        int(grid_transposed), _dt(heads[0]), boxes.data_ptr(), scores.data_ptr(),
                                        _ffi.current_stream()), "ydet_decode_v2v3")
    return boxes, scores


def nms_v5(pred, conf_thres=0.25, iou_thres=0.45, multi_label=False, agnostic=False, max_det=300, max_nms=None,
           max_wh=4096.0, classes=None, row_max="auto"):
    """pred [N, nboxes, 5+nc] fp32 CUDA.  Returns (det [N, max_det, 6], count [N] int32) on the device;
    det rows are (x1, y1, x2, y2, score, cls) in descending score order.
    row_max: "auto" uses the row maxima decode_v5 attached to `pred` (if still valid), None forces the scan."""
    _need_cuda(pred)
    lib = _ffi.lib()
    rm = row_max_of(pred) if isinstance(row_max, str) else row_max
    if pred.dtype != torch.float32 or not pred.is_contiguous():
        rm = None
    if pred.dtype != torch.float32:
        pred = pred.float()
    pred = pred.contiguous()
    N, nboxes, no = pred.shape
    max_nms = max_det if max_nms is None else max_nms
    dev = pred.device
    det = torch.empty(N, max_det, 6, device=dev, dtype=torch.float32)
    count = torch.empty(N, device=dev, dtype=torch.int32)
    if N == 0 or nboxes == 0:
        return det.zero_(), count.zero_()
    ws_bytes = lib.ydet_nms_v5_workspace_bytes(N, nboxes, no)
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    mask_ptr = None
    if classes is not None:
        m = torch.zeros(no - 5, dtype=torch.uint8)
        m[torch.as_tensor(list(classes), dtype=torch.long)] = 1
        mask = m.to(dev)
        mask_ptr = mask.data_ptr()
    with torch.cuda.device(dev):
        _ffi.check(lib.ydet_nms_v5(pred.data_ptr(), N, nboxes, no, float(conf_thres), float(iou_thres),
                                   int(bool(multi_label)), int(bool(agnostic)), int(max_nms), int(max_det),
                                   float(max_wh), mask_ptr, rm.data_ptr() if rm is not None else None,
                                   det.data_ptr(), count.data_ptr(), ws.data_ptr(), ws_bytes,
                                   _ffi.current_stream()), "ydet_nms_v5")
    return det, count


NMS2_METHODS = {"hard_nms_batch": 0, "hard_nms": 1, "hard_nms_myself": 2, "and": 3, "merge": 4, "soft_nms": 5, "diou_nms": 6}


def nms2(pred, conf_thres=0.25, nms_thres=0.6, multi_cls=True, method="diou_nms"):
    """pred [N, nboxes, 5+nc] fp32 CUDA -> (rows [N, cap, 6], count [N] int32, overflow [1] int32) on the device:
    yolov5_official/utils.py:318-460, rows (x1, y1, x2, y2, score, cls) in descending score order."""
    _need_cuda(pred)
    if method not in NMS2_METHODS:
        raise ValueError(f"unknown NMS method {method!r}")
    if method == "merge" and not nms_thres < 1.0:
        raise ValueError("merge needs nms_thres < 1 (the reference does not terminate otherwise)")
    lib = _ffi.lib()
    pred = pred.float().contiguous()
    N, nboxes, no = pred.shape
    dev = pred.device
    m = NMS2_METHODS[method]
    cap = min(lib.ydet_nms2_out_cap(no, m), max(nboxes * (no - 5), 1))
    rows = torch.empty(N, cap, 6, device=dev, dtype=torch.float32)
    count = torch.zeros(N, device=dev, dtype=torch.int32)
    overflow = torch.zeros(1, device=dev, dtype=torch.int32)
    if N == 0 or nboxes == 0:
        return rows, count, overflow
    ws_bytes = lib.ydet_nms2_workspace_bytes(N, nboxes, no, m)
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        _ffi.check(lib.ydet_nms2(pred.data_ptr(), N, nboxes, no, float(conf_thres), float(nms_thres), int(bool(multi_cls)), m,
                                 rows.data_ptr(), cap, count.data_ptr(), overflow.data_ptr(), ws.data_ptr(), ws_bytes,
                                 _ffi.current_stream()), "ydet_nms2")
    return rows, count, overflow


def nms_cluster_diou(pred, conf_thres=0.1, iou_thres=0.6, max_box=1500, classes=None, agnostic=False):
    """pred [N, nboxes, 5+nc] fp32 CUDA -> (rows [N, max_box, 6], count [N] int32): yolov5_official/utils.py:463-548."""
    _need_cuda(pred)
    lib = _ffi.lib()
    pred = pred.float().contiguous()
    N, nboxes, no = pred.shape
    dev = pred.device
    rows = torch.empty(N, max_box, 6, device=dev, dtype=torch.float32)
    count = torch.zeros(N, device=dev, dtype=torch.int32)
    if N == 0:
        return rows, count
    mask_ptr = None
    if classes:
        m = torch.zeros(no - 5, dtype=torch.uint8)
        m[torch.as_tensor(list(classes), dtype=torch.long)] = 1
        mask = m.to(dev)
        mask_ptr = mask.data_ptr()
    ws_bytes = lib.ydet_nms_cluster_diou_workspace_bytes(N, nboxes, no, max_box)
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        _ffi.check(lib.ydet_nms_cluster_diou(pred.data_ptr(), N, nboxes, no, float(conf_thres), float(iou_thres), int(max_box),
                                             int(bool(agnostic)), mask_ptr, rows.data_ptr(), count.data_ptr(), ws.data_ptr(),
                                             ws_bytes, _ffi.current_stream()), "ydet_nms_cluster_diou")
    return rows, count


def nms_perclass(boxes, scores, conf_thresh, nms_thresh, workspace_bytes=None):
    """boxes [N, n, 4], scores [N, n, nc] fp32 CUDA.
    Returns (keep [N, n] uint8, cls [N, n] int32, best [N, n] fp32): utils/utils.py:101-133.
    workspace_bytes: None = the full size (bit matrices for long class segments); "min" or a byte count in between
    leave some or all segments to the round-by-round walk (same result)."""
    _need_cuda(boxes, scores)
    lib = _ffi.lib()
    boxes = boxes.float().contiguous()
    scores = scores.float().contiguous()
    N, n, nc = scores.shape
    dev = boxes.device
    keep = torch.empty(N, n, device=dev, dtype=torch.uint8)
    cls = torch.empty(N, n, device=dev, dtype=torch.int32)
    best = torch.empty(N, n, device=dev, dtype=torch.float32)
    if N == 0 or n == 0:
        return keep, cls, best
    ws_bytes = lib.ydet_nms_perclass_workspace_bytes(N, n, nc)
    if workspace_bytes is not None:
        ws_min = lib.ydet_nms_perclass_workspace_bytes_min(N, n, nc)
        ws_bytes = ws_min if workspace_bytes == "min" else max(ws_min, min(ws_bytes, int(workspace_bytes)))
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        _ffi.check(lib.ydet_nms_perclass(boxes.data_ptr(), scores.data_ptr(), N, n, nc, float(conf_thresh),
                                         float(nms_thresh), keep.data_ptr(), cls.data_ptr(), best.data_ptr(),
                                         ws.data_ptr(), ws_bytes, _ffi.current_stream()), "ydet_nms_perclass")
    return keep, cls, best


def box_iou(box1, box2):
    """IoU matrix [n1, n2] of xyxy boxes (yolov5_official/utils.py:601-611)."""
    _need_cuda(box1, box2)
    lib = _ffi.lib()
    a = box1.float().contiguous()
    b = box2.float().contiguous()
    out = torch.empty(a.shape[0], b.shape[0], device=a.device, dtype=torch.float32)
    with torch.cuda.device(a.device):
        _ffi.check(lib.ydet_box_iou(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], out.data_ptr(),
                                    _ffi.current_stream()), "ydet_box_iou")
    return out


def process_batch(det, count, labels, label_start, iouv):
    """TP matrices of a batch in one launch (yolov5_official/utils.py:1103-1123 per image).
    det [B, max_det, 6] fp32, count [B] int32, labels [L, 5] (cls, xyxy) grouped by image, label_start [B + 1] int32,
    iouv [k] fp32, all CUDA.  Returns correct [B, max_det, k] bool (rows >= count are False)."""
    _need_cuda(det, count, label_start, iouv)
    lib = _ffi.lib()
    det = det.float().contiguous()
    count = count.to(torch.int32).contiguous()
    labels = labels.float().contiguous()
    label_start = label_start.to(torch.int32).contiguous()
    iouv = iouv.float().contiguous()
    B, max_det, _ = det.shape
    out = torch.empty(B, max_det, iouv.numel(), device=det.device, dtype=torch.uint8)
    if B == 0:
        return out.bool()
    assert label_start.numel() == B + 1 and count.numel() == B
    with torch.cuda.device(det.device):
        _ffi.check(lib.ydet_process_batch(det.data_ptr(), count.data_ptr(), B, max_det, labels.data_ptr(),
                                          label_start.data_ptr(), iouv.data_ptr(), iouv.numel(), out.data_ptr(),
                                          _ffi.current_stream()), "ydet_process_batch")
    return out.bool()


def voc_match(boxes, members, group_start, gt_boxes, gt_difficult, gt_start, ovthresh=0.5):
    """The matching loop of the VOC metric (data/vocapi_evaluator.py:284-318) for all (class, image) groups in one
    launch.  boxes [n, 4] float64; members [n] int32 (detection indices group by group, in visiting order);
    group_start / gt_start [G + 1] int32; gt_boxes [m, 4] float64; gt_difficult [m] uint8 -- all CUDA.
    Returns flags [n] uint8 per detection: 1 true positive, 2 false positive, 0 neither."""
    _need_cuda(boxes, members, group_start, gt_boxes, gt_difficult, gt_start)
    lib = _ffi.lib()
    boxes = boxes.to(torch.float64).contiguous()
    gt_boxes = gt_boxes.to(torch.float64).contiguous()
    members = members.to(torch.int32).contiguous()
    group_start = group_start.to(torch.int32).contiguous()
    gt_start = gt_start.to(torch.int32).contiguous()
    gt_difficult = gt_difficult.to(torch.uint8).contiguous()
    n, G = boxes.shape[0], group_start.numel() - 1
    assert members.numel() == n and gt_start.numel() == G + 1 and gt_boxes.shape[0] == gt_difficult.numel()
    flags = torch.zeros(n, device=boxes.device, dtype=torch.uint8)
    if n == 0 or G <= 0:
        return flags
    nbytes = lib.ydet_voc_match_workspace_bytes(n)
    ws = torch.empty((nbytes + 7) // 8, device=boxes.device, dtype=torch.float64)
    with torch.cuda.device(boxes.device):
        _ffi.check(lib.ydet_voc_match(boxes.data_ptr(), n, members.data_ptr(), group_start.data_ptr(), G,
                                      gt_boxes.data_ptr(), gt_difficult.data_ptr(), gt_start.data_ptr(), float(ovthresh),
                                      flags.data_ptr(), ws.data_ptr(), ws.numel() * 8, _ffi.current_stream()),
                   "ydet_voc_match")
    return flags
