#!/usr/bin/env python3
"""Diagnostic: the (boxes, scores) the per-class NMS of a family-A detector receives for image 0 of the synthetic batch,
with a few statistics of the pair test on its largest class (how many pairs a division-free verdict leaves open)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    tag, size, batch = sys.argv[1] if len(sys.argv) > 1 else "v4", 608, 4
    from helpers import build_fa
    from yolo_amd import engine, ops
    from yolo_amd.synthetic import images
    engine.set_compute_dtype("bf16")
    dev = torch.device("cuda:0")
    net = build_fa(tag, size).to(dev)
    x = images(batch, size, seed=1, kind="randn").to(dev)
    got = {}
    real = ops.nms_perclass

    def spy(boxes, scores, conf, thr, **kw):
        got["b"], got["s"], got["conf"], got["thr"] = boxes[0].float().cpu().numpy(), scores[0].float().cpu().numpy(), conf, thr
        return real(boxes, scores, conf, thr, **kw)

    ops.nms_perclass = spy
    with torch.no_grad():
        net.detect_batch(x)
    b, s, conf, thr = got["b"], got["s"], got["conf"], np.float32(got["thr"])
    print("conf", conf, "thr", thr, "boxes", b.shape, "min/max", b.min(), b.max(), "nan", np.isnan(b).sum(), "inf", np.isinf(b).sum())
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    print("area: zero", (area == 0).sum(), "neg", (area < 0).sum(), "min pos", area[area > 0].min(), "median", np.median(area))
    cls = s.argmax(1)
    best = s.max(1)
    on = best >= conf
    cnt = np.bincount(cls[on], minlength=s.shape[1])
    c = cnt.argmax()
    idx = np.nonzero(on & (cls == c))[0]
    print("largest class", c, "boxes", len(idx))
    bb = b[idx][:1024]
    ar = area[idx][:1024]
    xx1 = np.maximum(bb[:, None, 0], bb[None, :, 0]); yy1 = np.maximum(bb[:, None, 1], bb[None, :, 1])
    xx2 = np.minimum(bb[:, None, 2], bb[None, :, 2]); yy2 = np.minimum(bb[:, None, 3], bb[None, :, 3])
    w = np.maximum(np.float32(1e-28), xx2 - xx1); h = np.maximum(np.float32(1e-28), yy2 - yy1)
    inter = w * h
    uni = ar[:, None] + ar[None, :] - inter
    lo = thr * np.float32(0.99999952316284) * uni
    hi = thr * np.float32(1.00000047683716) * uni
    decided = (lo >= np.float32(1e-30)) & ((inter >= hi) | (inter <= lo))
    print("pairs", decided.size, "undecided", (~decided).sum(), "of which lo<1e-30", (lo < 1e-30).sum(),
          "rows with any undecided", (~decided).any(1).sum())
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "fa_boxes_img0.npz"), b=b, s=s)


if __name__ == "__main__":
    main()
