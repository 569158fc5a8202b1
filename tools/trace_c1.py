#!/usr/bin/env python3
"""YOLOv3-tiny 416 batch 1 (configuration C1 of bench.py) for a kernel trace:
    rocprofv3 --kernel-trace --stats -- python3 tools/trace_c1.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from yolo_amd.synthetic import images  # noqa: E402

from yolo_amd import engine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
engine.set_compute_dtype(sys.argv[2] if len(sys.argv) > 2 else "bf16")
engine.set_split_k(len(sys.argv) > 3 and sys.argv[3] == "splitk")
dev = torch.device("cuda:0")
net = bench.build_family_a("v3tiny", 416).to(dev)
x = images(1, 416, seed=1, kind="randn").to(dev)
with torch.no_grad():
    for _ in range(3):
        net.detect_batch(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        net.detect_batch(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"{steps} steps: host enqueue {(t1 - t0) / steps * 1e3:.3f} ms per step, with the GPU {(t2 - t0) / steps * 1e3:.3f} ms per step")
