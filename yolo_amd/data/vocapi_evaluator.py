"""VOC2007 average-precision evaluation with the interface of the reference's data/vocapi_evaluator.py.

The metric definition is the reference's: detections are rounded exactly like its per-class result
files (`{:.3f}` score, `{:.1f}` coordinates after +1), ranked by descending score, matched greedily
to ground truth at IoU > 0.5 (difficult boxes neither TP nor FP), AP by the 11-point VOC07 rule, and a
class without any detection contributes -1 to the mean.  The text files themselves are optional:
the rounding is applied in memory.  Detection runs on the GPU (`net(x)` of an AnchorDetector);
when torch.distributed is initialised the image list is sharded across ranks and the padded
per-image detections are exchanged with one all-gather before the AP computation.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np
import torch

from .. import distributed as ydist


def voc_ap(rec, prec, use_07_metric=True):
    """AP from recall / precision arrays (reference :194-225)."""
    rec, prec = np.asarray(rec), np.asarray(prec)   # the input dtype is kept: a float32 curve accumulates in float32
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            above = rec >= t
            ap = ap + (prec[above].max() if above.any() else 0.0) / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    step = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1])


def round_like_results_file(dets):
    """[n,5] (x1,y1,x2,y2,score) -> what the reference reads back from its result files (:149-152).  The `+ 1` is
    done in the array's own dtype, as the reference's `dets[k, 0] + 1` is (float32 for the evaluator's boxes)."""
    dets = np.asarray(dets)
    dets = dets.reshape(-1, 5) if dets.dtype == np.float32 else dets.astype(np.float64).reshape(-1, 5)
    out = np.empty(dets.shape, dtype=np.float64)
    for k in range(dets.shape[0]):
        out[k, 4] = float("{:.3f}".format(dets[k, 4]))
        for j in range(4):
            out[k, j] = float("{:.1f}".format(dets[k, j] + 1))
    return out


def voc_eval_class(dets_per_image, gt_per_image, ovthresh=0.5, use_07_metric=True):
    """One class.  dets_per_image: {image: [n,5] rounded dets}; gt_per_image: {image: (bbox [m,4], difficult [m])}.
    Returns (rec, prec, ap); (-1, -1, -1) when the class has no detection at all (reference :228-333)."""
    image_ids, conf, bb = [], [], []
    for image in dets_per_image:  # file order = image order, detections in their stored order
        d = dets_per_image[image]
        for k in range(len(d)):
            image_ids.append(image)
            conf.append(d[k, 4])
            bb.append(d[k, :4])
    npos = int(sum((~g[1]).sum() for g in gt_per_image.values()))
    if not image_ids:
        return -1.0, -1.0, -1.0
    conf, bb = np.array(conf), np.array(bb, dtype=float)
    order = np.argsort(-conf)
    claimed = {image: np.zeros(len(g[0]), dtype=bool) for image, g in gt_per_image.items()}
    tp, fp = np.zeros(len(order)), np.zeros(len(order))
    for d, k in enumerate(order):
        image = image_ids[k]
        gtb, difficult = gt_per_image.get(image, (np.zeros((0, 4)), np.zeros(0, dtype=bool)))
        ovmax, jmax = -np.inf, -1
        if len(gtb):
            g = gtb.astype(float)
            iw = np.maximum(np.minimum(g[:, 2], bb[k, 2]) - np.maximum(g[:, 0], bb[k, 0]), 0.0)
            ih = np.maximum(np.minimum(g[:, 3], bb[k, 3]) - np.maximum(g[:, 1], bb[k, 1]), 0.0)
            inter = iw * ih
            union = (bb[k, 2] - bb[k, 0]) * (bb[k, 3] - bb[k, 1]) + (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1]) - inter
            ov = inter / union
            ovmax, jmax = ov.max(), int(ov.argmax())
        if ovmax > ovthresh:
            if not difficult[jmax]:
                if not claimed[image][jmax]:
                    tp[d] = 1.0
                    claimed[image][jmax] = True
                else:
                    fp[d] = 1.0
        else:
            fp[d] = 1.0
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def round_like_results_file_fast(dets):
    """round_like_results_file without the per-value string formatting: for float32 inputs x*10 and x*1000 are exact
    in float64, so rint (half to even, as the decimal formatter rounds the exact value) followed by the division gives
    the double nearest to the printed decimal -- the value float() parses back."""
    d = np.asarray(dets)
    if d.dtype != np.float32:
        return round_like_results_file(d)
    d = d.astype(np.float64).reshape(-1, 5)
    out = np.empty_like(d)
    out[:, :4] = np.rint((d[:, :4].astype(np.float32) + np.float32(1)).astype(np.float64) * 10.0) / 10.0
    out[:, 4] = np.rint(d[:, 4] * 1000.0) / 1000.0
    return out


def voc_eval_all_device(box_list, names, recs, labelmap, device, ovthresh=0.5, use_07_metric=True):
    """voc_eval for every class with the matching loop on the GPU (one ydet_voc_match launch for the whole
    evaluation).  box_list[class][image] = [k, 5] detections; recs[name] = parse_rec-style objects.  Returns the list
    of per-class APs (-1 for a class without detections), equal to voc_eval_class's."""
    from .. import ops
    n_img = len(names)
    bb_all, members, grp_start, gt_all, diff_all, gt_start = [], [], [0], [], [], [0]
    per_class = []      # (offset of the class's detections in bb_all, count, npos)
    off = 0
    for ci, cls in enumerate(labelmap):
        gts = []
        for n in names:
            objs = [o for o in recs[n] if o["name"] == cls]
            gts.append((np.array([o["bbox"] for o in objs], dtype=np.float64).reshape(-1, 4),
                        np.array([o["difficult"] for o in objs], dtype=bool)))
        npos = int(sum((~g[1]).sum() for g in gts))
        img_of, dets = [], []
        for ii in range(n_img):
            d = box_list[ci][ii]
            if len(d):
                r = round_like_results_file_fast(d)
                dets.append(r)
                img_of.append(np.full(len(r), ii, dtype=np.int64))
        if not dets:
            per_class.append((off, 0, npos))
            continue
        dets, img_of = np.concatenate(dets), np.concatenate(img_of)
        order = np.argsort(-dets[:, 4])            # the reference's call: its tie order is this numpy's
        dets, img_of = dets[order], img_of[order]  # position = rank
        by_img = np.argsort(img_of, kind="stable") # ranks grouped by image, ascending rank inside an image
        counts = np.bincount(img_of, minlength=n_img)
        members.append(by_img + off)
        pos = 0
        for ii in range(n_img):
            if counts[ii]:
                pos += int(counts[ii])
                grp_start.append(off + pos)
                gt_all.append(gts[ii][0])
                diff_all.append(gts[ii][1])
                gt_start.append(gt_start[-1] + len(gts[ii][0]))
        bb_all.append(dets[:, :4])
        per_class.append((off, len(dets), npos))
        off += len(dets)
    if off == 0:
        return [-1.0] * len(labelmap)
    dev = torch.device(device)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    flags = ops.voc_match(t(np.concatenate(bb_all), torch.float64), t(np.concatenate(members), torch.int32),
                          t(np.array(grp_start), torch.int32),
                          t(np.concatenate(gt_all).reshape(-1, 4) if gt_all else np.zeros((0, 4)), torch.float64),
                          t(np.concatenate(diff_all) if diff_all else np.zeros(0, bool), torch.uint8),
                          t(np.array(gt_start), torch.int32), ovthresh).cpu().numpy()
    aps = []
    for o, n, npos in per_class:
        if n == 0:
            aps.append(-1.0)
            continue
        f = flags[o:o + n]
        tp, fp = np.cumsum((f == 1).astype(np.float64)), np.cumsum((f == 2).astype(np.float64))
        rec = tp / float(npos)
        prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
        aps.append(voc_ap(rec, prec, use_07_metric))
    return aps


def parse_rec(filename):
    """Objects of one PASCAL VOC annotation file (reference :95-112)."""
    objects = []
    for obj in ET.parse(filename).findall("object"):
        box = obj.find("bndbox")
        objects.append({
            "name": obj.find("name").text,
            "pose": obj.find("pose").text,
            "truncated": int(obj.find("truncated").text),
            "difficult": int(obj.find("difficult").text),
            "bbox": [int(box.find(k).text) for k in ("xmin", "ymin", "xmax", "ymax")],
        })
    return objects


class VOCAPIEvaluator:
    """VOC AP evaluation (reference :24-338).  `dataset` must offer __len__, `ids` (entries whose [1] is the
    image name) and `pull_item(i) -> (image tensor [3,S,S], gt, h, w)`; if it is not given, the reference's
    VOCDataset is expected at data.dataset (host-side I/O, not part of this package)."""

    def __init__(self, data_root, img_size, transform, labelmap, set_type="test", year="2007", display=False,
                 dataset=None, annotations=None):
        self.data_root, self.img_size, self.transform, self.labelmap = data_root, img_size, transform, labelmap
        self.set_type, self.year, self.display = set_type, year, display
        self.devkit_path = str(data_root) + "VOC" + year
        self.annopath = os.path.join(str(data_root), "VOC2007", "Annotations", "%s.xml")
        self.imgsetpath = os.path.join(str(data_root), "VOC2007", "ImageSets", "Main", set_type + ".txt")
        if dataset is None:
            raise ValueError("VOCAPIEvaluator needs a dataset object (VOC file I/O is outside this package)")
        self.dataset = dataset
        self.annotations = annotations  # optional {image name: parse_rec-style list}; else read from annopath
        self.map = None
        self.device = None   # set by evaluate(): a CUDA device sends the matching loop of the metric to the GPU

    # ---- detection --------------------------------------------------------------------------------
    @torch.no_grad()
    def evaluate(self, net, batch_size=1, device="cuda"):
        net.eval()
        self.device = torch.device(device)
        n_img = len(self.dataset)
        rank, world = ydist.world()
        lo, hi = ydist.shard_range(n_img, rank, world)
        max_keep = 0
        local = []  # (image index, [k,6] = x1,y1,x2,y2,score,cls in native pixels)
        for s in range(lo, hi, batch_size):
            idx = list(range(s, min(s + batch_size, hi)))
            items = [self.dataset.pull_item(i) for i in idx]
            x = torch.stack([it[0] for it in items]).to(device)
            if hasattr(net, "detect_batch"):
                boxes, keep, cls, best = net.detect_batch(x)
                for j, i in enumerate(idx):
                    k = keep[j].bool()
                    d = torch.cat([boxes[j][k], best[j][k, None], cls[j][k, None].float()], 1)
                    local.append((i, d, items[j][2], items[j][3]))
            else:
                for j, i in enumerate(idx):
                    b, sc, c = net(x[j:j + 1])
                    d = torch.from_numpy(np.concatenate([b, sc[:, None], c[:, None].astype(np.float32)], 1))
                    local.append((i, d.to(device), items[j][2], items[j][3]))
        for _, d, _, _ in local:
            max_keep = max(max_keep, d.shape[0])
        if world > 1:
            m = torch.tensor([max_keep], device=device)
            torch.distributed.all_reduce(m, op=torch.distributed.ReduceOp.MAX)
            max_keep = int(m.item())
        det = torch.zeros(len(local), max(max_keep, 1), 6, device=device)
        cnt = torch.zeros(len(local), dtype=torch.int32, device=device)
        ids = torch.zeros(len(local), dtype=torch.int64, device=device)
        for r, (i, d, h, w) in enumerate(local):
            d = d.clone()
            d[:, :4] *= torch.tensor([w, h, w, h], device=d.device, dtype=d.dtype)  # reference :68-69
            det[r, : d.shape[0]] = d
            cnt[r], ids[r] = d.shape[0], i
        det, cnt, ids = ydist.all_gather_detections(det, cnt, ids, total=n_img)  # the one exchange step: one collective, no size exchange
        det, cnt = det.cpu().numpy(), cnt.cpu().numpy()
        self.all_boxes = [[np.empty((0, 5), dtype=np.float32) for _ in range(n_img)] for _ in self.labelmap]
        for r, i in enumerate(ids.tolist()):
            d = det[r, : cnt[r]]
            for j in range(len(self.labelmap)):
                sel = d[:, 5] == j
                if sel.any():
                    self.all_boxes[j][i] = d[sel, :5].astype(np.float32)
        self.evaluate_detections(self.all_boxes)
        return self.map

    # ---- metric -----------------------------------------------------------------------------------
    def _ground_truth(self):
        names = [e[1] for e in self.dataset.ids]
        if self.annotations is not None:
            return names, {n: self.annotations[n] for n in names}
        return names, {n: parse_rec(self.annopath % n) for n in names}

    def evaluate_detections(self, box_list, use_07=True):
        names, recs = self._ground_truth()
        if self.device is not None and self.device.type == "cuda":
            aps = voc_eval_all_device(box_list, names, recs, self.labelmap, self.device, 0.5, use_07)
            if self.display:
                for cls, ap in zip(self.labelmap, aps):
                    print("AP for {} = {:.4f}".format(cls, ap))
            return self._finish(aps)
        aps = []
        for ci, cls in enumerate(self.labelmap):
            gts = {}
            for n in names:
                objs = [o for o in recs[n] if o["name"] == cls]
                gts[n] = (np.array([o["bbox"] for o in objs], dtype=float).reshape(-1, 4),
                          np.array([o["difficult"] for o in objs], dtype=bool))
            dets = {}
            for ii, n in enumerate(names):
                d = box_list[ci][ii]
                if len(d):
                    dets[n] = round_like_results_file(d)
            aps.append(voc_eval_class(dets, gts, 0.5, use_07)[2])
            if self.display:
                print("AP for {} = {:.4f}".format(cls, aps[-1]))
        return self._finish(aps)

    def _finish(self, aps):
        self.aps = aps
        self.map = float(np.mean(aps))
        rank, _ = ydist.world()
        if rank == 0:
            print("Mean AP = {:.4f}".format(self.map))
        return self.map

    def write_voc_results_file(self, all_boxes, out_dir):
        """Optional: the per-class text files in the reference's format (:137-152)."""
        os.makedirs(out_dir, exist_ok=True)
        for ci, cls in enumerate(self.labelmap):
            with open(os.path.join(out_dir, "det_" + self.set_type + "_%s.txt" % cls), "wt") as fh:
                for ii, entry in enumerate(self.dataset.ids):
                    d = all_boxes[ci][ii]
                    for k in range(len(d)):
                        fh.write("{:s} {:.3f} {:.1f} {:.1f} {:.1f} {:.1f}\n".format(
                            entry[1], d[k, -1], d[k, 0] + 1, d[k, 1] + 1, d[k, 2] + 1, d[k, 3] + 1))
