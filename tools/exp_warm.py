#!/usr/bin/env python3
"""Completion time of every step of a run of 60 pipelined steps (YOLOv5s 640 bs64, three streams) after W warm-up steps:
does the rate settle at once, or over tens of steps?   python tools/exp_warm.py [W]"""
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

W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
lib = _ffi.lib()
engine.set_compute_dtype("bf16")
model = Model()
init_deterministic(model, seed=0, head_convs=list(model.model[-1].m))
model = model.to(dev).eval()
x = images(64, 640, seed=100).to(dev)
S, K = 3, int(sys.argv[2]) if len(sys.argv) > 2 else 60
streams = [torch.cuda.Stream(dev) for _ in range(S)]
pdet = [torch.empty(64, 300, 6, pin_memory=True) for _ in range(S)]
pcnt = [torch.empty(64, dtype=torch.int32, pin_memory=True) for _ in range(S)]


def step(si=0):
    with torch.cuda.stream(streams[si]):
        key = (tuple(x.shape), "bf16", x.device.index)
        plan = model._plans().get(key, lambda: engine.build_plan(dev, "bf16", 64, lambda g, n: model._lower(g, n, *x.shape[1:])))
        outs = plan.run(x)
        z, _ = model.model[-1].finish(outs)
        det, cnt = non_max_suppression_padded(z, 0.001, 0.6, multi_label=True)
        pdet[si].copy_(det, non_blocking=True)
        pcnt[si].copy_(cnt, non_blocking=True)


with torch.no_grad():
    for k in range(W):
        step(k % S)
    torch.cuda.synchronize()
    print("allocator after warm-up:", torch.cuda.memory_reserved() >> 20, "MB reserved,", torch.cuda.memory_stats()["num_device_alloc"], "device allocations")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    host = []
    with torch.cuda.stream(streams[0]):
        ev[0].record()
    t0 = time.perf_counter()
    for k in range(K):
        step(k % S)
        with torch.cuda.stream(streams[k % S]):
            ev[k + 1].record()
        host.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    done = [ev[0].elapsed_time(e) for e in ev[1:]]
    print("allocator after the run:", torch.cuda.memory_reserved() >> 20, "MB reserved,", torch.cuda.memory_stats()["num_device_alloc"], "device allocations")
    print("step: GPU completion (ms since start), host enqueue done (ms)")
    for k in range(K):
        if k < 12 or k % 10 == 9:
            print(f"{k:3d} {done[k]:8.2f} {host[k]:8.2f}")
    for a, b in [(i, i + 20) for i in range(0, K, 20)]:
        print(f"steps {a}..{b - 1}: {(done[b - 1] - (done[a - 1] if a else 0.0)) / (b - a):.3f} ms per step")
