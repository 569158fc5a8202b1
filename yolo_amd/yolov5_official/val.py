"""Validation loop with the signature and return contract of the reference's yolov5_official/val.py
`run`: forward -> NMS -> rescale to native image space -> TP matrix -> per-class AP.

Differences in mechanism, not in results: forward, decode, NMS and the IoU matrices run on the GPU
through libydet.so; detections stay in padded device tensors (`[bs, max_det, 6]` + counts) until they
are needed on the host; when torch.distributed is initialised each rank evaluates one contiguous range of
whole batches (distributed.shard_range; a dataloader that walks an indexable dataset in order is re-built over
just that range, so a rank never loads an image it does not own; other iterables are drawn lazily up to the end
of the range) and the ranks exchange the padded detections with one all_gather_into_tensor, the
(variable-length) labels with a second, before the AP computation.
"""
from pathlib import Path

import numpy as np
import torch

from .. import distributed as ydist
from .. import ops
from .utils import (ap_per_class, non_max_suppression_padded, process_batch, scale_coords, xywh2xyxy, xyxy2xywh)


def save_one_txt(predn, save_conf, shape, file):
    gn = torch.tensor(shape)[[1, 0, 1, 0]]
    with open(file, "a") as fh:
        for *xyxy, conf, cls in predn.tolist():
            xywh = (xyxy2xywh(torch.tensor(xyxy).view(1, 4)) / gn).view(-1).tolist()
            line = (cls, *xywh, conf) if save_conf else (cls, *xywh)
            fh.write(("%g " * len(line)).rstrip() % line + "\n")


def _shard_batches(dataloader, rank, world):
    """-> (iterator of (global index of the batch's first image, batch) over the batches this rank owns,
           images per rank as a list when every rank can compute it without an exchange, else None).

    Ranks own contiguous ranges of whole batches (distributed.shard_range over the batch count).
    * A torch DataLoader that walks an indexable dataset in order (sequential sampler, default batch sampler, no
      drop_last) is re-built over just this rank's image range with the same batch size, collate function, workers and
      worker initialiser: a rank never loads an image it does not own, and every rank knows every shard's size.
    * Any other iterable with a length (a list of batches, a DataLoader with a sampler of its own) is walked lazily up to
      the end of this rank's range; batches in front of the range are drawn and dropped without being moved to the device.
    * An iterable without a length (a generator) has to be drawn completely first: its batch count is not known otherwise."""
    def running(batches, base=0):
        for b in batches:
            yield base, b
            base += b[0].shape[0]

    if world == 1:
        return running(dataloader), None
    ds, bs = getattr(dataloader, "dataset", None), getattr(dataloader, "batch_size", None)
    tud = torch.utils.data
    in_order = (isinstance(dataloader, tud.DataLoader) and bs and hasattr(ds, "__len__") and hasattr(ds, "__getitem__")
                and isinstance(dataloader.sampler, tud.SequentialSampler) and type(dataloader.batch_sampler) is tud.BatchSampler
                and not dataloader.drop_last)
    if in_order:
        n = len(ds)
        nbatch = (n + bs - 1) // bs
        bounds = [ydist.shard_range(nbatch, r, world) for r in range(world)]
        sizes = [min(b1 * bs, n) - min(b0 * bs, n) for b0, b1 in bounds]
        lo = min(bounds[rank][0] * bs, n)
        kw = dict(batch_size=bs, shuffle=False, num_workers=dataloader.num_workers, collate_fn=dataloader.collate_fn,
                  pin_memory=dataloader.pin_memory, worker_init_fn=dataloader.worker_init_fn, timeout=dataloader.timeout)
        if dataloader.num_workers > 0:
            kw.update(persistent_workers=dataloader.persistent_workers, prefetch_factor=dataloader.prefetch_factor)
        sub = tud.DataLoader(tud.Subset(ds, range(lo, lo + sizes[rank])), **kw)
        return running(sub, lo), sizes
    if not hasattr(dataloader, "__len__"):
        dataloader = list(dataloader)
    b0, b1 = ydist.shard_range(len(dataloader), rank, world)

    def owned():
        base = 0
        for i, b in enumerate(dataloader):
            if i >= b1:
                break
            if i >= b0:
                yield base, b
            base += b[0].shape[0]
    return owned(), None


@torch.no_grad()
def run(data,
        weights=None,
        conf_thres=0.001,
        iou_thres=0.6,
        device="cuda",
        single_cls=False,
        augment=False,
        save_txt=False,
        save_hybrid=False,
        save_conf=True,
        save_json=False,
        half=True,
        model=None,
        dataloader=None,
        save_dir="",
        plots=False,
        compute_loss=None,
        max_det=300,
        ):
    """Returns ((mp, mr, map50, map, *losses), maps, t) like the reference (val.py:27-186)."""
    dev = torch.device(device)
    half = half and dev.type != "cpu"
    if half:
        model.half()
    model.eval()
    nc = 1 if single_cls else int(data["nc"])
    iouv = torch.linspace(0.5, 0.95, 10).to(dev)
    niou = iouv.numel()
    names = dict(enumerate(model.names if hasattr(model, "names") else model.module.names))
    rank, world = ydist.world()
    loss = torch.zeros(3, device=dev)

    dets, counts, ids, tgts = [], [], [], []   # this rank's padded detections / labels in native image space
    batches, sizes = _shard_batches(dataloader, rank, world)
    for img_base, (img, targets, paths, shapes) in batches:
        nb = img.shape[0]
        img = img.to(dev, non_blocking=True)
        if not (img.dtype == torch.uint8 and img.is_cuda and not augment):
            img = (img.half() if half else img.float()) / 255.0
        # else: uint8 batches go to the model as they are; its first kernel applies the / 255 while loading
        targets = targets.to(dev).clone()
        _, _, height, width = img.shape
        out, train_out = model(img, augment=augment)
        if compute_loss:
            loss += compute_loss([t.float() for t in train_out], targets)[1]
        targets[:, 2:] *= torch.tensor([width, height, width, height], device=dev, dtype=targets.dtype)
        lb = [targets[targets[:, 0] == i, 1:] for i in range(nb)] if save_hybrid else []
        det, cnt = non_max_suppression_padded(out.float(), conf_thres, iou_thres, labels=lb, multi_label=True,
                                              agnostic=single_cls, max_det=max_det)
        if single_cls:
            det[..., 5] = 0
        # native-space boxes: the per-image gain/pad differ, so the rescale is done per image on the device
        for si in range(nb):
            scale_coords(img[si].shape[1:], det[si, :, :4], shapes[si][0], shapes[si][1])
        sel = targets.clone()
        if len(sel):
            box = xywh2xyxy(sel[:, 2:6])
            for si in range(nb):
                rows = sel[:, 0] == si
                if rows.any():
                    b = box[rows]
                    scale_coords(img[si].shape[1:], b, shapes[si][0], shapes[si][1])
                    box[rows] = b
            sel[:, 2:6] = box
            sel[:, 0] += img_base
        dets.append(det)
        counts.append(cnt)
        ids.append(torch.arange(img_base, img_base + nb, device=dev))
        tgts.append(sel)
        if save_txt:
            c_host = cnt.tolist()
            for si in range(nb):
                save_one_txt(det[si, :c_host[si]], save_conf, shapes[si][0],
                             file=str(save_dir) + "labels/" + Path(paths[si]).stem + ".txt")

    def _cat(xs, shape, dtype):
        return torch.cat(xs, 0) if xs else torch.zeros(shape, device=dev, dtype=dtype)

    det = _cat(dets, (0, max_det, 6), torch.float32)
    cnt = _cat(counts, (0,), torch.int32)
    iid = _cat(ids, (0,), torch.int64)
    tgt = _cat(tgts, (0, 6), torch.float32).float()
    # the one exchange step: padded detections + labels of every shard
    # (shard sizes known to every rank: no size exchange and no host synchronisation in front of the collective)
    det, cnt, iid = ydist.all_gather_detections(det, cnt, iid, sizes=sizes)
    tgt = ydist.all_gather_rows(tgt)

    stats = []
    seen = int(iid.numel())
    cnt_host = cnt.tolist()
    iid_host = iid.tolist()
    tgt_img = tgt[:, 0].long() if len(tgt) else tgt.new_zeros(0, dtype=torch.long)
    if dev.type == "cuda" and seen:
        # the TP matrices of every image in one launch; only booleans, confidences and classes leave the GPU
        pos = torch.searchsorted(iid, tgt_img)                       # row of each label's image (iid is sorted)
        order = torch.argsort(pos, stable=True)
        per_img = torch.bincount(pos, minlength=seen)
        start = torch.zeros(seen + 1, device=dev, dtype=torch.int32)
        start[1:] = torch.cumsum(per_img, 0)
        lab_sorted = tgt[order, 1:]
        correct_all = ops.process_batch(det, cnt, lab_sorted, start, iouv).cpu()
        conf_all, cls_all = det[..., 4].cpu(), det[..., 5].cpu()
        start_host, tcls_host = start.tolist(), lab_sorted[:, 0].tolist()
        for k in range(seen):
            c, tcls = cnt_host[k], tcls_host[start_host[k]:start_host[k + 1]]
            if c == 0:
                if tcls:
                    stats.append((torch.zeros(0, niou, dtype=torch.bool), torch.Tensor(), torch.Tensor(), tcls))
                continue
            stats.append((correct_all[k, :c], conf_all[k, :c], cls_all[k, :c], tcls))
    else:
        for k, image in enumerate(iid_host):
            pred = det[k, :cnt_host[k]]
            labels = tgt[tgt_img == image, 1:]
            nl = len(labels)
            tcls = labels[:, 0].tolist() if nl else []
            if len(pred) == 0:
                if nl:
                    stats.append((torch.zeros(0, niou, dtype=torch.bool), torch.Tensor(), torch.Tensor(), tcls))
                continue
            correct = process_batch(pred, labels, iouv) if nl else torch.zeros(pred.shape[0], niou, dtype=torch.bool)
            stats.append((correct.cpu(), pred[:, 4].cpu(), pred[:, 5].cpu(), tcls))

    p = r = f1 = mp = mr = map50 = map95 = 0.0
    ap, ap_class = [], []
    stats = [np.concatenate(x, 0) for x in zip(*stats)]
    if len(stats) and stats[0].any():
        p, r, ap, f1, ap_class = ap_per_class(*stats, plot=plots, save_dir=save_dir, names=names)
        ap50, ap = ap[:, 0], ap.mean(1)
        mp, mr, map50, map95 = p.mean(), r.mean(), ap50.mean(), ap.mean()
        nt = np.bincount(stats[3].astype(np.int64), minlength=nc)
    else:
        nt = np.zeros(1)
    if rank == 0:
        print(("%20s" + "%11s" * 6) % ("Class", "Images", "Labels", "P", "R", "mAP@.5", "mAP@.5:.95"))
        print(("%20s" + "%11i" * 2 + "%11.3g" * 4) % ("all", seen, nt.sum(), mp, mr, map50, map95))
    if half:
        model.float()
    maps = np.zeros(nc) + map95
    for i, c in enumerate(ap_class):
        maps[c] = ap[i]
    t = (0.0, 0.0, 0.0)
    n_batches = max(len(dataloader), 1) if hasattr(dataloader, "__len__") else 1
    return (mp, mr, map50, map95, *(loss.cpu() / n_batches).tolist()), maps, t
