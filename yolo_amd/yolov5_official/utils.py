"""Post-processing and metric helpers with the signatures of the reference's yolov5_official/utils.py.

`non_max_suppression` and `box_iou` run on the GPU through libydet.so; the box-format helpers and the
AP bookkeeping are small host-side numpy/torch routines, as in the reference.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import ops
# numpy greedy NMS of one class (reference :551-578): line for line the function of utils/utils.py:15-42 -- 1e-28 floor on the
# intersection sides, survivors while ovr <= threshold -- so the two import names share the GPU-backed implementation
from ..utils.utils import nms  # noqa: F401


# ---- box formats (reference :581-598) --------------------------------------------------------------
def _like(x):
    return x.clone() if isinstance(x, torch.Tensor) else np.copy(x)


def xyxy2xywh(x):
    y = _like(x)
    y[:, 0] = (x[:, 0] + x[:, 2]) / 2
    y[:, 1] = (x[:, 1] + x[:, 3]) / 2
    y[:, 2] = x[:, 2] - x[:, 0]
    y[:, 3] = x[:, 3] - x[:, 1]
    return y


def xywh2xyxy(x):
    y = _like(x)
    half_w, half_h = x[:, 2] / 2, x[:, 3] / 2
    y[:, 0] = x[:, 0] - half_w
    y[:, 1] = x[:, 1] - half_h
    y[:, 2] = x[:, 0] + half_w
    y[:, 3] = x[:, 1] + half_h
    return y


def box_iou(box1, box2):
    """IoU matrix [N, M] of xyxy boxes (reference :601-611).  CUDA tensors go through ydet_box_iou."""
    if isinstance(box1, torch.Tensor) and box1.is_cuda:
        return ops.box_iou(box1, box2)
    b1, b2 = torch.as_tensor(box1), torch.as_tensor(box2)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    inter = (torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])).clamp(0).prod(2)
    return inter / (a1[:, None] + a2 - inter)


def clip_coords(boxes, shape):
    """Clip xyxy boxes to an image of (height, width) in place (reference :1126-1135)."""
    if isinstance(boxes, torch.Tensor):
        boxes[:, 0].clamp_(0, shape[1])
        boxes[:, 1].clamp_(0, shape[0])
        boxes[:, 2].clamp_(0, shape[1])
        boxes[:, 3].clamp_(0, shape[0])
    else:
        boxes[:, [0, 2]] = boxes[:, [0, 2]].clip(0, shape[1])
        boxes[:, [1, 3]] = boxes[:, [1, 3]].clip(0, shape[0])


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None):
    """Map xyxy boxes from the letterboxed shape back to the native image (reference :1184-1197)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    clip_coords(coords, img0_shape)
    return coords


# ---- NMS (reference :113-195) ----------------------------------------------------------------------
def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=(), max_det=300):
    """Per-image NMS of Detect output [bs, n, 5+nc]; returns a list of [k, 6] (xyxy, conf, cls) tensors
    in descending confidence.  Same filtering, multi-label expansion, top-`max_det` pre-filter and
    class-offset trick as the reference; the work is done by ydet_nms_v5 on the GPU."""
    det, count = non_max_suppression_padded(prediction, conf_thres, iou_thres, classes, agnostic, multi_label, labels,
                                            max_det)
    counts = count.tolist()  # the one device->host sync of the post-process
    return [det[i, :c] for i, c in enumerate(counts)]


def non_max_suppression_padded(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                               multi_label=False, labels=(), max_det=300):
    """Same as non_max_suppression but stays on the device: (det [bs, max_det, 6], count [bs] int32)."""
    if not prediction.is_cuda:
        raise RuntimeError("non_max_suppression runs on the GPU only: pass a CUDA prediction tensor")
    if labels and any(len(l) for l in labels):
        prediction = _append_apriori_labels(prediction, labels)
    return ops.nms_v5(prediction, conf_thres, iou_thres, multi_label=multi_label, agnostic=agnostic, max_det=max_det,
                      max_nms=max_det, max_wh=4096.0, classes=classes)


def non_max_suppression2(prediction, conf_thres=0.25, nms_thres=0.6, multi_cls=True, method='diou_nms'):
    """Reference :318-460: per-class NMS by `method` ('hard_nms_batch', 'hard_nms', 'hard_nms_myself', 'and', 'merge',
    'soft_nms', 'diou_nms').  Returns, per image, None (nothing passed the filters) or a [k, 6] tensor (xyxy, score,
    cls) in descending score; the work is done by ydet_nms2 on the GPU."""
    if not prediction.is_cuda:
        raise RuntimeError("non_max_suppression2 runs on the GPU only: pass a CUDA prediction tensor")
    rows, count, overflow = ops.nms2(prediction, conf_thres, nms_thres, multi_cls, method)
    counts = count.tolist() + overflow.tolist()      # the one device->host sync
    if counts[-1]:
        raise RuntimeError("non_max_suppression2: a class holds more tied / unlimited candidates than the GPU lists (2048)")
    return [rows[i, :c] if c else None for i, c in enumerate(counts[:-1])]


def diou_non_max_suppression(prediction, conf_thres=0.1, iou_thres=0.6, max_box=1500, classes=None, agnostic=False,
                             use_gpu=False, max_det=300, labels=None, multi_label=True):
    """Reference :463-548 (Cluster-Weighted DIoU-NMS; `use_gpu`, `max_det`, `labels` and `multi_label` are accepted and
    unused there too: multi-label is forced by nc > 1).  Returns per image the kept rows of its max_box slots -- like
    the reference, unused slots come back as rows with a NaN box and zero score.  Runs ydet_nms_cluster_diou."""
    if not prediction.is_cuda:
        raise RuntimeError("diou_non_max_suppression runs on the GPU only: pass a CUDA prediction tensor")
    rows, count = ops.nms_cluster_diou(prediction, conf_thres, iou_thres, max_box, classes, agnostic)
    return [rows[i, :c] for i, c in enumerate(count.tolist())]


def _append_apriori_labels(prediction, labels):
    """Autolabelling rows (reference :138-144): each label becomes a box with obj = cls = 1 placed after
    the image's own boxes; images are padded with all-zero rows, which never pass the obj test."""
    bs, n, no = prediction.shape
    extra = max(len(l) for l in labels)
    out = prediction.new_zeros(bs, n + extra, no)
    out[:, :n] = prediction
    for i, l in enumerate(labels):
        if len(l):
            l = l.to(prediction.device)
            rows = torch.arange(len(l), device=prediction.device)
            out[i, n + rows, :4] = l[:, 1:5].to(out.dtype)
            out[i, n + rows, 4] = 1.0
            out[i, n + rows, l[:, 0].long() + 5] = 1.0
    return out


# ---- evaluation (reference :1103-1123, :1221-1291) ---------------------------------------------------
def process_batch(predictions, labels, iouv):
    """TP matrix [n_pred, n_iou] of one image: per class, predictions whose best IoU with a label of that
    class exceeds iouv[0] claim that label once, in prediction order."""
    n = predictions.shape[0]
    nl = labels.shape[0]
    if predictions.is_cuda and n and nl:      # one launch on the device (ydet_process_batch), nothing leaves the GPU
        dev = predictions.device
        return ops.process_batch(predictions[None, :, :6], torch.tensor([n], device=dev, dtype=torch.int32), labels.to(dev),
                                 torch.tensor([0, nl], device=dev, dtype=torch.int32), iouv.to(dev))[0]
    correct = np.zeros((n, len(iouv)), dtype=bool)
    if n and nl:
        iou_all = box_iou(predictions[:, :4], labels[:, 1:5]).cpu().numpy()
        pcls = predictions[:, 5].cpu().numpy()
        tcls = labels[:, 0].cpu().numpy()
        thr = iouv.cpu().numpy()
        claimed_total = 0
        for c in np.unique(tcls):
            ti = np.nonzero(tcls == c)[0]
            pi = np.nonzero(pcls == c)[0]
            if not len(pi):
                continue
            sub = iou_all[np.ix_(pi, ti)]
            best, arg = sub.max(1), sub.argmax(1)
            claimed = set()
            for j in np.nonzero(best > thr[0])[0]:
                d = int(ti[arg[j]])
                if d not in claimed:
                    claimed.add(d)
                    claimed_total += 1
                    correct[pi[j]] = best[j] > thr
                    if claimed_total == nl:
                        break
    return torch.from_numpy(correct).to(iouv.device)


_trapezoid = getattr(np, "trapezoid", None) or np.trapz


def compute_ap(recall, precision):
    """101-point interpolated AP of a PR curve (reference :1221-1246)."""
    mrec = np.concatenate(([0.0], recall, [recall[-1] + 0.01]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    x = np.linspace(0, 1, 101)
    ap = _trapezoid(np.interp(x, mrec, mpre), x)
    return ap, mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, plot=False, save_dir=".", names=()):
    """Per-class AP at every IoU threshold (reference :1249-1291); returns (p, r, ap, f1, classes)."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes = np.unique(target_cls)
    px = np.linspace(0, 1, 1000)
    ap = np.zeros((len(classes), tp.shape[1]))
    p, r = np.zeros((len(classes), 1000)), np.zeros((len(classes), 1000))
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        n_l, n_p = (target_cls == c).sum(), sel.sum()
        if n_p == 0 or n_l == 0:
            continue
        tpc = tp[sel].cumsum(0)
        fpc = (1 - tp[sel]).cumsum(0)
        recall = tpc / (n_l + 1e-16)
        precision = tpc / (tpc + fpc)
        r[ci] = np.interp(-px, -conf[sel], recall[:, 0], left=0)
        p[ci] = np.interp(-px, -conf[sel], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])[0]
    f1 = 2 * p * r / (p + r + 1e-16)
    i = f1.mean(0).argmax()
    return p[:, i], r[:, i], ap, f1[:, i], classes.astype("int32")


# ---- model utilities (reference :1674-1694, :1697-1762) ----------------------------------------------
def fuse_conv_and_bn(conv, bn):
    """Return a Conv2d(bias=True) equal to bn(conv(.)) in eval mode."""
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                      groups=conv.groups, bias=True).requires_grad_(False).to(conv.weight.device)
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    fused.weight.copy_((conv.weight.view(conv.out_channels, -1) * scale.view(-1, 1)).view(fused.weight.shape))
    b_conv = torch.zeros(conv.out_channels, device=conv.weight.device) if conv.bias is None else conv.bias
    fused.bias.copy_((b_conv - bn.running_mean) * scale + bn.bias)
    return fused


def make_divisible(x, divisor):
    return math.ceil(x / divisor) * divisor                                            # reference :981-983


def letterbox(img, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleFill=False, scaleup=True, stride=32):
    """Resize + pad one uint8 HWC image to `new_shape` keeping its aspect ratio (reference :40-75); the pixels are
    produced on the GPU.  Returns (uint8 HWC CUDA tensor, ratio, (dw, dh))."""
    from ..preprocess import letterbox_geometry, preprocess_u8
    new_unpad, ratio, pad, top, bottom, left, right = letterbox_geometry(tuple(img.shape[:2]), new_shape, auto,
                                                                         scaleFill, scaleup, stride)
    out_hw = (new_unpad[1] + top + bottom, new_unpad[0] + left + right)
    x = preprocess_u8([img], [(new_unpad[0], new_unpad[1], left, top)], out_hw, pad_value=color, scale_mode=0,
                      dtype=torch.uint8, channels_last=True)
    return x[0], ratio, pad


class Detections:
    """Per-image results of AutoShape (data fields of reference :1765-1780; the display helpers are not part of
    the inference path)."""

    def __init__(self, imgs, pred, files, names=None, shape=None):
        d = pred[0].device
        gn = [torch.tensor([*[im.shape[i] for i in [1, 0, 1, 0]], 1., 1.], device=d) for im in imgs]
        self.imgs, self.pred, self.names, self.files = imgs, pred, names, files
        self.xyxy = pred
        self.xywh = [xyxy2xywh(x) for x in pred]
        self.xyxyn = [x / g for x, g in zip(self.xyxy, gn)]
        self.xywhn = [x / g for x, g in zip(self.xywh, gn)]
        self.n = len(self.pred)
        self.s = shape

    def __len__(self):
        return self.n


class AutoShape(nn.Module):
    """Input-robust wrapper (reference :1697-1762).  A tensor batch goes straight through the model, as in the
    reference.  numpy HWC / CHW images (RGB, uint8) are letterboxed to a common stride-multiple shape, scaled by
    1/255 and stacked by ONE pre-processing launch on the GPU, then model -> NMS -> boxes mapped back to each
    image's own pixels.  File names, URLs and PIL images need decoders that live outside this package."""
    conf, iou, classes, max_det = 0.25, 0.45, None, 1000

    def __init__(self, model):
        super().__init__()
        self.model = model.eval()
        self.stride = getattr(model, "stride", torch.tensor([32.]))
        self.names = getattr(model, "names", None)

    def autoshape(self):
        return self

    @torch.no_grad()
    def forward(self, imgs, size=640, augment=False, profile=False):
        from ..preprocess import letterbox_batch
        p = next(self.model.parameters())
        if isinstance(imgs, torch.Tensor):
            return self.model(imgs.to(p.device).type_as(p), augment)
        n, imgs = (len(imgs), list(imgs)) if isinstance(imgs, (list, tuple)) else (1, [imgs])
        shape0, shape1, files = [], [], []
        for i, im in enumerate(imgs):
            if not isinstance(im, np.ndarray):
                raise NotImplementedError("AutoShape: pass numpy images or a tensor batch (file / URL / PIL inputs "
                                          "need an image decoder)")
            if im.shape[0] < 5:
                im = im.transpose((1, 2, 0))
            im = im[..., :3] if im.ndim == 3 else np.tile(im[..., None], 3)
            s = im.shape[:2]
            shape0.append(s)
            g = size / max(s)
            shape1.append([y * g for y in s])
            imgs[i] = np.ascontiguousarray(im)
            files.append(f"image{i}.jpg")
        shape1 = [make_divisible(x, int(torch.as_tensor(self.stride).max())) for x in np.stack(shape1, 0).max(0)]
        # uint8 canvas; the model's first kernel applies the / 255 while loading (the reference's `x / 255.`, :1751)
        x, _, _ = letterbox_batch(imgs, new_shape=shape1, auto=False, bgr_to_rgb=False, scale=False, device=p.device)
        y = self.model(x, augment)[0]
        y = non_max_suppression(y, self.conf, iou_thres=self.iou, classes=self.classes, max_det=self.max_det)
        for i in range(n):
            scale_coords(shape1, y[i][:, :4], shape0[i])
        return Detections(imgs, y, files, self.names, x.shape)
