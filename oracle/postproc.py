"""numpy restatement of the reference's decode and NMS code (test infrastructure; see oracle/__init__.py).

All arithmetic is float32, written in the same operation order as the reference expressions.
"""
import numpy as np

f32 = np.float32


def sigmoid(x):
    x = np.asarray(x, dtype=f32)
    return (f32(1) / (f32(1) + np.exp(-x, dtype=f32))).astype(f32)


# ---------------------------------------------------------------------------------------------------
# YOLOv5 Detect decode -- yolov5_official/model.py:56-81 (grid :83-85); models/yolo.py:42-71 (transposed)
# ---------------------------------------------------------------------------------------------------
def detect_decode(heads_nchw, strides, anchor_grid, na, no, grid_transposed=False):
    """heads_nchw: per level [bs, na*no, ny, nx] raw conv outputs; anchor_grid [nl, na, 2] pixels.
    Returns (z [bs, sum na*ny*nx, no], [x_i (bs, na, ny, nx, no)])."""
    z, xs = [], []
    for i, h in enumerate(heads_nchw):
        h = np.asarray(h, dtype=f32)
        bs, _, ny, nx = h.shape
        x = h.reshape(bs, na, no, ny, nx).transpose(0, 1, 3, 4, 2).copy()          # model.py:63
        if grid_transposed:                                                           # models/yolo.py:68-71
            gy, gx = np.meshgrid(np.arange(ny), np.arange(nx))
            grid = np.stack((gx, gy), 2).reshape(1, 1, ny, nx, 2).astype(f32)
        else:                                                                         # model.py:83-85
            yv, xv = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
            grid = np.stack((xv, yv), 2).reshape(1, 1, ny, nx, 2).astype(f32)
        y = sigmoid(x)                                                                # :70
        xy = (y[..., 0:2] * f32(2.) - f32(0.5) + grid) * f32(strides[i])               # :72
        wh = (y[..., 2:4] * f32(2)) ** 2 * np.asarray(anchor_grid[i], f32).reshape(1, na, 1, 1, 2)  # :73
        y = np.concatenate((xy, wh, y[..., 4:]), -1).astype(f32)
        z.append(y.reshape(bs, -1, no))                                               # :78
        xs.append(x)
    return np.concatenate(z, 1), xs


# ---------------------------------------------------------------------------------------------------
# box helpers -- yolov5_official/utils.py:591-611
# ---------------------------------------------------------------------------------------------------
def xywh2xyxy(x):
    x = np.asarray(x, dtype=f32)
    y = np.copy(x)
    y[:, 0] = x[:, 0] - x[:, 2] / f32(2)
    y[:, 1] = x[:, 1] - x[:, 3] / f32(2)
    y[:, 2] = x[:, 0] + x[:, 2] / f32(2)
    y[:, 3] = x[:, 1] + x[:, 3] / f32(2)
    return y


def box_iou(box1, box2):
    box1, box2 = np.asarray(box1, f32), np.asarray(box2, f32)
    a1 = (box1[:, 2] - box1[:, 0]) * (box1[:, 3] - box1[:, 1])
    a2 = (box2[:, 2] - box2[:, 0]) * (box2[:, 3] - box2[:, 1])
    wh = np.clip(np.minimum(box1[:, None, 2:], box2[:, 2:]) - np.maximum(box1[:, None, :2], box2[:, :2]), 0, None)
    inter = wh[..., 0] * wh[..., 1]
    return (inter / (a1[:, None] + a2 - inter)).astype(f32)


# ---------------------------------------------------------------------------------------------------
# torchvision.ops.nms -- NOT in the reference tree (parity unpinned): documented greedy algorithm,
# descending score, ties -> lower index first, suppress when IoU > threshold.
# ---------------------------------------------------------------------------------------------------
def torchvision_nms(boxes, scores, iou_threshold):
    boxes, scores = np.asarray(boxes, f32), np.asarray(scores, f32)
    n = len(scores)
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-scores, kind="stable")
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = f32(iou_threshold)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(f32(0), xx2 - xx1)
        h = np.maximum(f32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(invalid="ignore", divide="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > thr]] = True
    return np.asarray(keep, dtype=np.int64)


# ---------------------------------------------------------------------------------------------------
# non_max_suppression -- yolov5_official/utils.py:113-195
# ---------------------------------------------------------------------------------------------------
def nms_candidates(x, conf_thres, multi_label, classes=None, max_nms=300, agnostic=False, max_wh=4096):
    """Lines :135-181 for ONE image `x` [n, 5+nc]: returns (x6 [m,6] = xyxy,conf,cls ; boxes+offset ; scores)
    exactly as they reach the torchvision.ops.nms call."""
    x = np.array(x, dtype=f32)
    nc = x.shape[1] - 5
    multi_label = multi_label and nc > 1                                              # :128
    x = x[x[:, 4] > f32(conf_thres)]                                                  # :122,135
    if not x.shape[0]:
        return np.zeros((0, 6), f32), np.zeros((0, 4), f32), np.zeros(0, f32)
    x[:, 5:] *= x[:, 4:5]                                                             # :151
    box = xywh2xyxy(x[:, :4])                                                         # :154
    if multi_label:                                                                   # :157-159
        i, j = np.nonzero(x[:, 5:] > f32(conf_thres))
        x = np.concatenate((box[i], x[i, j + 5, None], j[:, None].astype(f32)), 1)
    else:                                                                             # :160-162
        j = x[:, 5:].argmax(1)
        conf = x[np.arange(len(x)), j + 5]
        x = np.concatenate((box, conf[:, None], j[:, None].astype(f32)), 1)[conf > f32(conf_thres)]
    if classes is not None:                                                           # :165-166
        x = x[np.isin(x[:, 5], np.asarray(classes, f32))]
    n = x.shape[0]
    if not n:
        return np.zeros((0, 6), f32), np.zeros((0, 4), f32), np.zeros(0, f32)
    if n > max_nms:                                                                   # :176-177
        x = x[np.argsort(-x[:, 4], kind="stable")[:max_nms]]
    c = x[:, 5:6] * f32(0 if agnostic else max_wh)                                    # :180
    return x.astype(f32), (x[:, :4] + c).astype(f32), x[:, 4].astype(f32)             # :181


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=(), max_det=300, nms_fn=torchvision_nms):
    prediction = np.asarray(prediction, dtype=f32)
    nc = prediction.shape[2] - 5
    output = [np.zeros((0, 6), f32)] * prediction.shape[0]
    for xi, x in enumerate(prediction):
        if labels and len(labels[xi]):                                                # :138-144 (after the obj filter)
            l = np.asarray(labels[xi], f32)
            x = x[x[:, 4] > f32(conf_thres)]
            v = np.zeros((len(l), nc + 5), f32)
            v[:, :4] = l[:, 1:5]
            v[:, 4] = 1.0
            v[np.arange(len(l)), l[:, 0].astype(np.int64) + 5] = 1.0
            x = np.concatenate((x, v), 0)
        x6, boxes, scores = nms_candidates(x, conf_thres, multi_label, classes, max_det, agnostic)
        if not len(x6):
            continue
        i = nms_fn(boxes, scores, iou_thres)[:max_det]                                # :182-184
        output[xi] = x6[i]
    return output


# ---------------------------------------------------------------------------------------------------
# the alternative NMS flavours -- yolov5_official/utils.py:198-244 (jaccard_diou), :265-315 (bbox_iou2),
# :318-460 (non_max_suppression2), :463-548 (diou_non_max_suppression)
# ---------------------------------------------------------------------------------------------------
def bbox_iou2(box1, box2, DIoU=False):
    """:265-315 with x1y1x2y2=True: IoU (or DIoU) of ONE xyxy box against [n,4] boxes, float32."""
    box1 = np.asarray(box1, f32)
    box2 = np.atleast_2d(np.asarray(box2, f32))[:, :4].T                                   # rows may carry (score, cls) too
    b1_x1, b1_y1, b1_x2, b1_y2 = box1[0], box1[1], box1[2], box1[3]
    b2_x1, b2_y1, b2_x2, b2_y2 = box2[0], box2[1], box2[2], box2[3]
    inter = np.maximum(np.minimum(b1_x2, b2_x2) - np.maximum(b1_x1, b2_x1), f32(0)) * \
        np.maximum(np.minimum(b1_y2, b2_y2) - np.maximum(b1_y1, b2_y1), f32(0))            # :288-289
    w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1
    w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1
    union = (w1 * h1 + f32(1e-16)) + w2 * h2 - inter                                       # :294
    iou = (inter / union).astype(f32)
    if DIoU:                                                                               # :297-309
        cw = np.maximum(b1_x2, b2_x2) - np.minimum(b1_x1, b2_x1)
        ch = np.maximum(b1_y2, b2_y2) - np.minimum(b1_y1, b2_y1)
        c2 = cw * cw + ch * ch + f32(1e-16)
        dx, dy = (b2_x1 + b2_x2) - (b1_x1 + b1_x2), (b2_y1 + b2_y2) - (b1_y1 + b1_y2)
        rho2 = dx * dx / f32(4) + dy * dy / f32(4)
        return (iou - rho2 / c2).astype(f32)
    return iou


NMS2_METHODS = ("hard_nms_batch", "hard_nms", "hard_nms_myself", "and", "merge", "soft_nms", "diou_nms")


def nms2_candidates(pred, conf_thres, multi_cls):
    """:336-371 for ONE image: rows (xyxy, score, cls) in descending score order (ties: lower row first)."""
    pred = np.array(pred, dtype=f32)
    min_wh, max_wh = f32(2), f32(4096)
    pred = pred[pred[:, 4] > f32(conf_thres)]                                              # :336
    pred = pred[(pred[:, 2:4] > min_wh).all(1) & (pred[:, 2:4] < max_wh).all(1)]           # :339
    if len(pred) == 0:
        return np.zeros((0, 6), f32)
    pred[:, 5:] *= pred[:, 4:5]                                                            # :346
    box = xywh2xyxy(pred[:, :4])
    if multi_cls or conf_thres < 0.01:                                                     # :352-359
        i, j = np.nonzero(pred[:, 5:] > f32(conf_thres))
        pred = np.concatenate((box[i], pred[i, j + 5, None], j[:, None].astype(f32)), 1)
    else:                                                                                  # :360-362
        j = pred[:, 5:].argmax(1)
        conf = pred[np.arange(len(pred)), j + 5]
        pred = np.concatenate((box, conf[:, None], j[:, None].astype(f32)), 1)[conf > f32(conf_thres)]
    if len(pred) == 0:
        return np.zeros((0, 6), f32)
    return pred[np.argsort(-pred[:, 4], kind="stable")].astype(f32)                        # :371


def nms2_one_class(dc, conf_thres, nms_thres, method):
    """:386-454 for the rows of one class (descending score): the rows appended to det_max, in order."""
    dc = np.array(dc, dtype=f32)
    n = len(dc)
    keep = []
    if n == 1:                                                                             # :387-389
        return dc
    if n > 500:                                                                            # :390-391
        dc = dc[:500]
    thr = f32(nms_thres)
    if method in ("hard_nms", "hard_nms_batch"):                                           # :377, :395 (torchvision)
        return dc[torchvision_nms(dc[:, :4], dc[:, 4], nms_thres)]
    if method in ("hard_nms_myself", "diou_nms"):                                          # :398-405, :447-454
        while dc.shape[0]:
            keep.append(dc[:1])
            if len(dc) == 1:
                break
            v = bbox_iou2(dc[0], dc[1:], DIoU=method == "diou_nms")
            dc = dc[1:][v < thr]
    elif method == "and":                                                                  # :408-413
        while len(dc) > 1:
            iou = bbox_iou2(dc[0], dc[1:])
            if iou.max() > f32(0.5):
                keep.append(dc[:1])
            dc = dc[1:][iou < thr]
    elif method == "merge":                                                                # :416-425
        while len(dc):
            if len(dc) == 1:
                keep.append(dc)
                break
            i = bbox_iou2(dc[0], dc) > thr
            weights = dc[i, 4:5]
            dc[0, :4] = (weights * dc[i, :4]).sum(0, dtype=f32) / weights.sum(dtype=f32)
            keep.append(dc[:1].copy())
            dc = dc[~i]
    elif method == "soft_nms":                                                             # :428-444
        sigma = f32(0.5)
        while len(dc):
            keep.append(dc[:1])
            if len(dc) == 1:
                break
            iou = bbox_iou2(dc[0], dc[1:])
            dc = dc[1:].copy()
            dc[:, 4] *= np.exp(-(iou * iou) / sigma, dtype=f32)
            dc = dc[dc[:, 4] > f32(conf_thres)]
    else:
        raise ValueError(method)
    return np.concatenate(keep, 0) if keep else np.zeros((0, 6), f32)


def non_max_suppression2(prediction, conf_thres=0.25, nms_thres=0.6, multi_cls=True, method="diou_nms"):
    """:318-460.  Per image None (nothing passed the filters) or [k,6] rows (xyxy, score, cls), descending score."""
    prediction = np.asarray(prediction, dtype=f32)
    output = [None] * len(prediction)
    for image_i, pred in enumerate(prediction):
        pred = nms2_candidates(pred, conf_thres, multi_cls)
        if len(pred) == 0:
            continue
        if method == "hard_nms_batch":                                                     # :375-379: class-wise by coordinate offset
            off = pred[:, 5:6] * (pred[:, :4].max() + f32(1))
            output[image_i] = pred[torchvision_nms(pred[:, :4] + off, pred[:, 4], nms_thres)]
            continue
        det_max = []
        cls = pred[:, 5]
        for c in np.unique(cls):                                                           # :384
            out_c = nms2_one_class(pred[cls == c], conf_thres, nms_thres, method)
            if len(out_c):
                det_max.append(out_c)
        if len(det_max):                                                                   # :456-458
            det_max = np.concatenate(det_max, 0)
            output[image_i] = det_max[np.argsort(-det_max[:, 4], kind="stable")]
    return output


def jaccard_diou(box_a, box_b):
    """:210-244 (iscrowd=False) for [A,4] x [B,4] xyxy boxes: IoU / (union + 1e-7) - D ** 0.7."""
    a = np.asarray(box_a, f32)[:, None, :]
    b = np.asarray(box_b, f32)[None, :, :]
    wh = np.maximum(np.minimum(a[..., 2:], b[..., 2:]) - np.maximum(a[..., :2], b[..., :2]), f32(0))   # :202-207
    inter = wh[..., 0] * wh[..., 1]
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    union = area_a + area_b - inter
    x1, y1 = (a[..., 2] + a[..., 0]) / f32(2), (a[..., 3] + a[..., 1]) / f32(2)
    x2, y2 = (b[..., 2] + b[..., 0]) / f32(2), (b[..., 3] + b[..., 1]) / f32(2)
    cr, cl = np.maximum(a[..., 2], b[..., 2]), np.minimum(a[..., 0], b[..., 0])
    ct, cb = np.minimum(a[..., 1], b[..., 1]), np.maximum(a[..., 3], b[..., 3])
    dx, dy, ew, eh = x2 - x1, y2 - y1, cr - cl, cb - ct
    D = (dx * dx + dy * dy) / (ew * ew + eh * eh + f32(1e-7))                              # :242
    return (inter / (union + f32(1e-7)) - np.power(D, f32(0.7), dtype=f32)).astype(f32)    # :243


def diou_nms_candidates(x, conf_thres, classes=None, agnostic=False, max_box=1500, max_wh=4096):
    """:484-523 for ONE image: (rows [<=max_box,6] in descending score order, their boxes offset by class)."""
    x = np.array(x, dtype=f32)
    nc = x.shape[1] - 5
    x = x[x[:, 4] > f32(conf_thres)]
    if not x.shape[0]:
        return np.zeros((0, 6), f32), np.zeros((0, 4), f32)
    x[:, 5:] *= x[:, 4:5]
    box = xywh2xyxy(x[:, :4])
    if nc > 1:                                                                             # :475 overrides the argument
        i, j = np.nonzero(x[:, 5:] > f32(conf_thres))
        x = np.concatenate((box[i], x[i, j + 5, None], j[:, None].astype(f32)), 1)
    else:
        j = x[:, 5:].argmax(1)
        conf = x[np.arange(len(x)), j + 5]
        x = np.concatenate((box, conf[:, None], j[:, None].astype(f32)), 1)[conf > f32(conf_thres)]
    if classes:
        x = x[np.isin(x[:, 5], np.asarray(classes, f32))]
    if not x.shape[0]:
        return np.zeros((0, 6), f32), np.zeros((0, 4), f32)
    x = x[np.argsort(-x[:, 4], kind="stable")]                                             # :518
    c = x[:, 5] * f32(0) if agnostic else x[:, 5]
    boxes = (x[:, :4] + c[:, None] * f32(max_wh))[:max_box]                                # :521
    return x[:max_box].astype(f32), boxes.astype(f32)


def cluster_diou_keep(boxes_off, iou_thres, max_iter=200):
    """:527-536 for ONE image's [max_box,4] offset boxes: (keep mask, the final matrix B)."""
    iou = np.triu(jaccard_diou(boxes_off, boxes_off), 1)
    B = iou
    maxA = None
    for _ in range(max_iter):
        A = B
        maxA = A.max(0)                                                                    # torch: A.max(dim=1) of [b,i,j] -> over i
        E = (maxA < f32(iou_thres)).astype(f32)[:, None]
        B = iou * E
        if np.array_equal(A, B):
            break
    return maxA <= f32(iou_thres), B


def diou_non_max_suppression(prediction, conf_thres=0.1, iou_thres=0.6, max_box=1500, classes=None, agnostic=False,
                             max_det=300, labels=None, multi_label=True):
    """:463-548 (Cluster-Weighted DIoU-NMS).  Every image returns the kept rows of its max_box slots: real
    detections with weighted coordinates, and one all-padding row (NaN box, zero score) per unused slot."""
    prediction = np.asarray(prediction, dtype=f32)
    out = []
    for x in prediction:
        rows, boxes = diou_nms_candidates(x, conf_thres, classes, agnostic, max_box)
        pred1 = np.zeros((max_box, 6), f32)
        pred2 = np.zeros((max_box, 4), f32)
        pred1[:len(rows)] = rows
        pred2[:len(boxes)] = boxes
        keep, B = cluster_diou_keep(pred2, iou_thres)
        weights = (B * (B > f32(0.8)) + np.eye(max_box, dtype=f32)) * pred1[None, :, 4]    # :541-542
        with np.errstate(invalid="ignore", divide="ignore"):
            pred1[:, :4] = (weights @ pred1[:, :4]) / weights.sum(1, keepdims=True)        # :543
        out.append(pred1[keep])
    return out


# ---------------------------------------------------------------------------------------------------
# v2/v3/v4 family: head split + decode -- models/v3tiny.py:88-111,137-147; utils/utils.py:46-98,136-213
# ---------------------------------------------------------------------------------------------------
def fa_grid(nx, ny):
    """utils/utils.py:167-169 / :185-187 -- note the transposed (row, col) contents."""
    grid_y, grid_x = np.meshgrid(np.arange(ny), np.arange(nx))
    return np.stack([grid_x, grid_y], axis=-1).astype(np.float64).reshape(1, nx * ny, 1, 2)


def fa_decode(heads_nchw, level_anchors, A, nc, v2, input_size, stride=32):
    """heads_nchw: per level [bs, A*(5+nc), H, W] fine -> coarse; level_anchors: per level [A,2].
    Returns (bboxes [bs, n, 4] clamped, scores [bs, n, nc]) -- the reference computes image 0 only."""
    confs, clss, boxes, grids, anchors = [], [], [], [], []
    for l, pred in enumerate(heads_nchw):
        pred = np.asarray(pred, f32)
        bs, c, h, w = pred.shape
        p = pred.transpose(0, 2, 3, 1).reshape(bs, h * w, c)                          # v3tiny.py:90
        confs.append(p[:, :, :A].reshape(bs, h * w * A, 1))                           # :93
        clss.append(p[:, :, A:(1 + nc) * A].reshape(bs, h * w * A, nc))               # :94-96
        boxes.append(p[:, :, (1 + nc) * A:].reshape(bs, h * w * A, 4))                # :97-99
        grids.append(fa_grid(w, h))
        anchors.append(np.tile(np.asarray(level_anchors[l], f32).reshape(1, A, 2), (h * w, 1, 1)))
    conf = np.concatenate(confs, 1)
    cls = np.concatenate(clss, 1)
    t = np.concatenate(boxes, 1)
    grid = np.concatenate(grids, 1).astype(f32)                                       # .float() at use (:54,:70)
    anc = np.concatenate(anchors, 0)[None]
    bs, n = t.shape[0], t.shape[1]
    hw = n // A
    t = t.reshape(bs, hw, A, 4)
    if v2:                                                                            # utils.py:46-58
        xy = sigmoid(t[..., :2]) + grid
        wh = np.exp(t[..., 2:], dtype=f32) * anc
        xywh = np.concatenate([xy, wh], -1).reshape(bs, hw * A, 4) * f32(stride)
    else:                                                                             # utils.py:61-74
        xy = (sigmoid(t[..., :2]) + grid) * f32(stride)
        wh = np.exp(t[..., 2:], dtype=f32) * anc
        xywh = np.concatenate([xy, wh], -1).reshape(bs, hw * A, 4)
    xyxy = np.zeros_like(xywh)                                                        # utils.py:92-96
    xyxy[:, :, 0] = xywh[:, :, 0] - xywh[:, :, 2] / f32(2)
    xyxy[:, :, 1] = xywh[:, :, 1] - xywh[:, :, 3] / f32(2)
    xyxy[:, :, 2] = xywh[:, :, 0] + xywh[:, :, 2] / f32(2)
    xyxy[:, :, 3] = xywh[:, :, 1] + xywh[:, :, 3] / f32(2)
    bboxes = np.clip(xyxy / f32(input_size), f32(0.), f32(1.))                        # v3tiny.py:145-146
    m = cls.max(-1, keepdims=True)
    e = np.exp(cls - m, dtype=f32)
    scores = e / e.sum(-1, keepdims=True) * sigmoid(conf)                             # :144,147
    return bboxes.astype(f32), scores.astype(f32)


# ---------------------------------------------------------------------------------------------------
# v2/v3/v4 family: NMS -- utils/utils.py:15-42 and :101-133
# ---------------------------------------------------------------------------------------------------
def nms_numpy(dets, scores, threshold):
    """utils/utils.py:15-42.  Tie order made explicit: stable ascending argsort reversed."""
    dets, scores = np.asarray(dets, f32), np.asarray(scores, f32)
    x1, y1, x2, y2 = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(scores, kind="stable")[::-1]
    thr = f32(threshold)
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        xx1 = np.maximum(x1[i], x1[order[1:]])
        yy1 = np.maximum(y1[i], y1[order[1:]])
        xx2 = np.minimum(x2[i], x2[order[1:]])
        yy2 = np.minimum(y2[i], y2[order[1:]])
        w = np.maximum(f32(1e-28), xx2 - xx1)
        h = np.maximum(f32(1e-28), yy2 - yy1)
        inter = w * h
        with np.errstate(invalid="ignore", divide="ignore"):
            ovr = inter / (areas[i] + areas[order[1:]] - inter)
        inds = np.where(ovr <= thr)[0]
        order = order[inds + 1]
    return keep


def postprocess(bb, scores, conf_thresh, num_classes, nms_threshold):
    """utils/utils.py:101-133."""
    bb, scores = np.asarray(bb, f32), np.asarray(scores, f32)
    cls_index = np.argmax(scores, axis=1)
    scores = scores[(np.arange(scores.shape[0]), cls_index)]
    keep = np.where(scores >= f32(conf_thresh))
    bb, scores, cls_index = bb[keep], scores[keep], cls_index[keep]
    keep = np.zeros(len(bb), dtype=np.int64)
    for i in range(num_classes):
        index = np.where(cls_index == i)[0]
        if len(index) == 0:
            continue
        ckeep = nms_numpy(bb[index], scores[index], nms_threshold)
        keep[index[ckeep]] = 1
    keep = np.where(keep > 0)
    return bb[keep], scores[keep], cls_index[keep]
