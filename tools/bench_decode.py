#!/usr/bin/env python3
"""Micro-benchmark of ydet_decode_v5 / ydet_nms_v5 at the headline size (64 x 25200 x 85)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import ops

dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator().manual_seed(0)
heads = []
for s in (80, 40, 20):
    buf = torch.zeros(N, s, s, 256, device=dev)
    buf[..., :255] = (torch.randn(N, s, s, 255, generator=g) * 1.5 - 3.0).to(dev)
    heads.append(buf[..., :255])
strides, anchors = [8., 16., 32.], [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

nbytes = N * 25200 * 85 * 4
t1 = timeit(lambda: ops.decode_v5(heads, strides, anchors, 3, 85, row_max=True))
t0 = timeit(lambda: ops.decode_v5(heads, strides, anchors, 3, 85, row_max=False))
print(f"decode with row_max {t1:.0f} us, without {t0:.0f} us  ({(nbytes + N * 8400 * 1024) / t1 / 1e6:.2f} TB/s)")
z = ops.decode_v5(heads, strides, anchors, 3, 85)
kw = dict(conf_thres=0.001, iou_thres=0.6, multi_label=True)
print(f"nms with hint {timeit(lambda: ops.nms_v5(z, **kw)):.0f} us, with scan {timeit(lambda: ops.nms_v5(z, row_max=None, **kw)):.0f} us")
kw = dict(conf_thres=0.25, iou_thres=0.45)
print(f"nms deploy regime: hint {timeit(lambda: ops.nms_v5(z, **kw)):.0f} us, scan {timeit(lambda: ops.nms_v5(z, row_max=None, **kw)):.0f} us")
