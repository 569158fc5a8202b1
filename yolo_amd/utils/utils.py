"""Post-processing API of the v2/v3/v4 detectors (reference utils/utils.py:15-213), GPU-backed.

`nms` and `postprocess` keep the reference's numpy-in / numpy-out contract but run ydet_nms_perclass on
the device; `decode_boxes` and the grid builders are small tensor helpers (the inference path of the
detectors does not call them: it uses the fused ydet_decode_v2v3 kernel on the raw head maps)."""
import numpy as np
import torch

from .. import ops


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("yolo_amd.utils.utils: NMS runs on the GPU only and no ROCm device is visible")
    return torch.device("cuda", torch.cuda.current_device())


def nms(dets, scores, threshold):
    """Greedy NMS of one class (reference :15-42): boxes are visited by descending score, the intersection
    sides are floored at 1e-28 (not 0) and a box survives while ovr <= threshold.
    Returns the kept indices in visiting order.  Equal scores are visited higher index first."""
    dets = np.asarray(dets, dtype=np.float32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    if len(scores) == 0:
        return []
    dev = _device()
    b = torch.from_numpy(dets).to(dev)[None]
    s = torch.from_numpy(scores).to(dev)[None, :, None]
    keep, _, _ = ops.nms_perclass(b, s, float("-inf"), threshold)
    idx = np.nonzero(keep[0].cpu().numpy())[0]
    order = np.lexsort((-idx, -scores[idx]))  # score descending, then index descending
    return [int(i) for i in idx[order]]


def postprocess(bb, scores, conf_thresh, num_classes, nms_threshold):
    """argmax class -> score >= conf_thresh -> per-class NMS; survivors in original box order
    (reference :101-133).  bb [n,4], scores [n,nc] numpy -> (bb[K,4] f32, scores[K] f32, cls[K] i64)."""
    bb = np.asarray(bb, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    if len(bb) == 0:
        return bb.reshape(0, 4), np.zeros(0, np.float32), np.zeros(0, np.int64)
    dev = _device()
    keep, cls, best = ops.nms_perclass(torch.from_numpy(bb).to(dev)[None], torch.from_numpy(scores).to(dev)[None],
                                       conf_thresh, nms_threshold)
    k = keep[0].bool().cpu().numpy()
    return bb[k], best[0].cpu().numpy()[k], cls[0].cpu().numpy()[k].astype(np.int64)


def select_kept(boxes, keep, cls, best, image=0):
    """Device tensors of one batch -> the reference's numpy triple for `image`."""
    k = keep[image].bool()
    return (boxes[image][k].cpu().numpy(), best[image][k].cpu().numpy(),
            cls[image][k].cpu().numpy().astype(np.int64))


# ---- tensor helpers ------------------------------------------------------------------------------
def xywh_to_xxyy(xywh, grid_cell, all_anchor_wh, stride=32):
    """'v2' decode: (sigmoid(txy)+grid, exp(twh)*anchor) * stride (reference :46-58)."""
    B, HW, A, _ = xywh.shape
    xy = torch.sigmoid(xywh[..., :2]) + grid_cell.float()
    wh = torch.exp(xywh[..., 2:]) * all_anchor_wh
    return torch.cat([xy, wh], -1).view(B, HW * A, 4) * stride


def xywh_to_xxyyv3(xywh, grid_cell, all_anchor_wh, stride=32):
    """'v3' decode: (sigmoid(txy)+grid)*stride, exp(twh)*anchor (reference :61-74)."""
    B, HW, A, _ = xywh.shape
    xy = (torch.sigmoid(xywh[..., :2]) + grid_cell.float()) * stride
    wh = torch.exp(xywh[..., 2:]) * all_anchor_wh
    return torch.cat([xy, wh], -1).view(B, HW * A, 4)


def decode_boxes(xywh_pred, all_anchor_wh, grid_cell, requires_grad=False, v2=True):
    """[B, HW, A, 4] (tx,ty,tw,th) -> [B, HW*A, 4] (xmin,ymin,xmax,ymax) in pixels (reference :77-98)."""
    c = xywh_to_xxyy(xywh_pred, grid_cell, all_anchor_wh) if v2 else xywh_to_xxyyv3(xywh_pred, grid_cell, all_anchor_wh)
    half = c[..., 2:] / 2
    return torch.cat([c[..., :2] - half, c[..., :2] + half], -1)


def _grid(nx, ny):
    """Cell offsets exactly as the reference builds them (:167-169, :185-187): np.meshgrid(arange(y), arange(x))
    stacked as (grid_x, grid_y), i.e. flat cell k = row*W + col stores (row, col)."""
    gy, gx = np.meshgrid(np.arange(ny), np.arange(nx))
    return torch.from_numpy(np.stack([gx, gy], axis=-1).astype(np.float64).reshape(1, nx * ny, 1, 2))


def create_grid(inputsize, stride, anchor_size, device=None):
    n = inputsize // stride
    grid = _grid(n, n)
    anchors = anchor_size.repeat(n * n, 1, 1).unsqueeze(0)
    return (grid.to(device), anchors.to(device)) if device is not None else (grid, anchors)


def create_gridv3(input_size, strides, num_anchors, anchor_size, device=None):
    grids, stride_t, anchors = [], [], []
    for i, s in enumerate(strides):
        nx = ny = input_size // s
        grids.append(_grid(nx, ny))
        stride_t.append(torch.ones(1, nx * ny, num_anchors, 2) * s)
        anchors.append(anchor_size[i].repeat(nx * ny, 1, 1))
    out = torch.cat(grids, 1), torch.cat(stride_t, 1), torch.cat(anchors, 0).unsqueeze(0)
    return tuple(t.to(device) for t in out) if device is not None else out


def iou_score(bboxes_a, bboxes_b):
    """Row-wise IoU of two [n,4] xyxy tensors (reference :566-578)."""
    tl = torch.max(bboxes_a[:, :2], bboxes_b[:, :2])
    br = torch.min(bboxes_a[:, 2:], bboxes_b[:, 2:])
    area_a = torch.prod(bboxes_a[:, 2:] - bboxes_a[:, :2], 1)
    area_b = torch.prod(bboxes_b[:, 2:] - bboxes_b[:, :2], 1)
    en = (tl < br).type(tl.type()).prod(dim=1)
    area_i = torch.prod(br - tl, 1) * en
    return area_i / (area_a + area_b - area_i)
