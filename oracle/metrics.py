"""numpy restatement of the reference's evaluation arithmetic (test infrastructure; see oracle/__init__.py)."""
import numpy as np

from .postproc import box_iou


def process_batch(predictions, labels, iouv):
    """yolov5_official/utils.py:1103-1123.  predictions [n,6] xyxy,conf,cls; labels [m,5] cls,xyxy; iouv [k]."""
    predictions, labels, iouv = np.asarray(predictions, np.float32), np.asarray(labels, np.float32), np.asarray(iouv, np.float32)
    correct = np.zeros((predictions.shape[0], len(iouv)), dtype=bool)
    detected = []
    tcls, pcls = labels[:, 0], predictions[:, 5]
    nl = labels.shape[0]
    for cls in np.unique(tcls):
        ti = np.nonzero(cls == tcls)[0]
        pi = np.nonzero(cls == pcls)[0]
        if pi.shape[0]:
            m = box_iou(predictions[pi, 0:4], labels[ti, 1:5])
            ious, i = m.max(1), m.argmax(1)
            detected_set = set()
            for j in np.nonzero(ious > iouv[0])[0]:
                d = ti[i[j]]
                if int(d) not in detected_set:
                    detected_set.add(int(d))
                    detected.append(d)
                    correct[pi[j]] = ious[j] > iouv
                    if len(detected) == nl:
                        break
    return correct


def compute_ap(recall, precision):
    """yolov5_official/utils.py:1221-1246 ('interp' method)."""
    mrec = np.concatenate(([0.], recall, [recall[-1] + 0.01]))
    mpre = np.concatenate(([1.], precision, [0.]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    y = np.interp(x, mrec, mpre)
    ap = np.sum((y[1:] + y[:-1]) * np.diff(x) / 2.0)   # np.trapz(y, x)
    return ap, mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls):
    """yolov5_official/utils.py:1249-1291."""
    i = np.argsort(-conf)
    tp, conf, pred_cls = tp[i], conf[i], pred_cls[i]
    unique_classes = np.unique(target_cls)
    nc = unique_classes.shape[0]
    px = np.linspace(0, 1, 1000)
    ap, p, r = np.zeros((nc, tp.shape[1])), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    for ci, c in enumerate(unique_classes):
        i = pred_cls == c
        n_l = (target_cls == c).sum()
        n_p = i.sum()
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[i]).cumsum(0)
        tpc = tp[i].cumsum(0)
        recall = tpc / (n_l + 1e-16)
        r[ci] = np.interp(-px, -conf[i], recall[:, 0], left=0)
        precision = tpc / (tpc + fpc)
        p[ci] = np.interp(-px, -conf[i], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])[0]
    f1 = 2 * p * r / (p + r + 1e-16)
    i = f1.mean(0).argmax()
    return p[:, i], r[:, i], ap, f1[:, i], unique_classes.astype('int32')


def voc_ap(rec, prec, use_07_metric=True):
    """data/vocapi_evaluator.py:194-225."""
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap = ap + p / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def voc_eval_lines(lines, class_recs, npos, ovthresh=0.5, use_07_metric=True):
    """data/vocapi_evaluator.py:267-333 from the point where the per-class result file has been read:
    `lines` are its text lines 'image score x1 y1 x2 y2'; class_recs {image: {'bbox','difficult','det'}}."""
    if not any(lines):
        return -1., -1., -1.
    splitlines = [x.strip().split(' ') for x in lines]
    image_ids = [x[0] for x in splitlines]
    confidence = np.array([float(x[1]) for x in splitlines])
    BB = np.array([[float(z) for z in x[2:]] for x in splitlines])
    sorted_ind = np.argsort(-confidence)
    BB = BB[sorted_ind, :]
    image_ids = [image_ids[x] for x in sorted_ind]
    nd = len(image_ids)
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d in range(nd):
        R = class_recs[image_ids[d]]
        bb = BB[d, :].astype(float)
        ovmax = -np.inf
        BBGT = R['bbox'].astype(float)
        if BBGT.size > 0:
            ixmin = np.maximum(BBGT[:, 0], bb[0])
            iymin = np.maximum(BBGT[:, 1], bb[1])
            ixmax = np.minimum(BBGT[:, 2], bb[2])
            iymax = np.minimum(BBGT[:, 3], bb[3])
            iw = np.maximum(ixmax - ixmin, 0.)
            ih = np.maximum(iymax - iymin, 0.)
            inters = iw * ih
            uni = ((bb[2] - bb[0]) * (bb[3] - bb[1]) + (BBGT[:, 2] - BBGT[:, 0]) * (BBGT[:, 3] - BBGT[:, 1]) - inters)
            overlaps = inters / uni
            ovmax = np.max(overlaps)
            jmax = np.argmax(overlaps)
        if ovmax > ovthresh:
            if not R['difficult'][jmax]:
                if not R['det'][jmax]:
                    tp[d] = 1.
                    R['det'][jmax] = 1
                else:
                    fp[d] = 1.
        else:
            fp[d] = 1.
    fp = np.cumsum(fp)
    tp = np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)
