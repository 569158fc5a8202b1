#!/usr/bin/env python3
"""Host time of the pieces of one step (plan.run, Detect.finish, NMS, D2H copies) for the first steps after a
synchronize, per stream."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolo_amd import _ffi, engine  # noqa: E402
from yolo_amd.synthetic import images, init_deterministic  # noqa: E402
from yolo_amd.yolov5_official.model import Model  # noqa: E402
from yolo_amd.yolov5_official.utils import non_max_suppression_padded  # noqa: E402

dev = torch.device("cuda:0")
lib = _ffi.lib()
engine.set_compute_dtype("bf16")
model = Model()
init_deterministic(model, seed=0, head_convs=list(model.model[-1].m))
model = model.to(dev).eval()
x = images(64, 640, seed=100).to(dev)
S = 3
streams = [torch.cuda.Stream(dev) for _ in range(S)]
pdet = [torch.empty(64, 300, 6, pin_memory=True) for _ in range(S)]
pcnt = [torch.empty(64, dtype=torch.int32, pin_memory=True) for _ in range(S)]


def step(si, log=None):
    t = [time.perf_counter()]
    with torch.cuda.stream(streams[si]):
        key = (tuple(x.shape), "bf16", x.device.index)
        plan = model._plans().get(key, lambda: engine.build_plan(dev, "bf16", 64, lambda g, n: model._lower(g, n, *x.shape[1:])))
        t.append(time.perf_counter())
        outs = plan.run(x)
        t.append(time.perf_counter())
        z, _ = model.model[-1].finish(outs)
        t.append(time.perf_counter())
        det, cnt = non_max_suppression_padded(z, 0.001, 0.6, multi_label=True)
        t.append(time.perf_counter())
        pdet[si].copy_(det, non_blocking=True)
        pcnt[si].copy_(cnt, non_blocking=True)
        t.append(time.perf_counter())
    if log is not None:
        log.append([1e3 * (b - a) for a, b in zip(t, t[1:])])


with torch.no_grad():
    for rnd in range(3):
        for k in range(5 if rnd == 0 else 9):
            step(k % S)
        torch.cuda.synchronize()
        log = []
        for k in range(9):
            step(k % S, log)
        torch.cuda.synchronize()
        print(f"round {rnd}: host ms per step [plan lookup, plan.run, finish, nms, copies]")
        for k, l in enumerate(log):
            print(f"  step {k} stream {k % S}: " + " ".join(f"{v:6.2f}" for v in l))
