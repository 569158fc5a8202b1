#!/usr/bin/env python3
"""The 80x80 prediction level (HBM-bound) on a side stream beside the rest of the neck (MFMA-bound 40x40 / 20x20
convolutions) of the same batch: ms per step, three pipelined batches, against the plain launch order.

    python tools/exp_head_overlap.py
"""
import ctypes as C
import os
import sys

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
S, K = int(os.environ.get("S", "3")), 120
streams = [torch.cuda.Stream(dev) for _ in range(S)]
sides = [torch.cuda.Stream(dev) for _ in range(S)]
pdet = [torch.empty(64, 300, 6, pin_memory=True) for _ in range(S)]
pcnt = [torch.empty(64, dtype=torch.int32, pin_memory=True) for _ in range(S)]
plans, outs = [], []
with torch.no_grad():
    for s in range(S):
        with torch.cuda.stream(streams[s]):
            plan = engine.build_plan(dev, "bf16", 64, lambda g, n: model._lower(g, n, *x.shape[1:]))
            plans.append(plan)
            outs.append(plan.run(x))
torch.cuda.synchronize()
n = len(plans[0].ops)
labels = [i.get("label", "") for i in plans[0].op_info]
h0 = labels.index("conv_head_decode")            # first (largest) prediction level
print("ops", n, "first head op", h0, "shape", plans[0].op_info[h0].get("shape"))
# the op that produces its input: the last op before the next stride-2 convolution of the neck; found by hand for
# YOLOv5s: op 33 (cv3 of the 80x80 C3)
FORK = int(os.environ.get("FORK", "34"))


def run_range(s, a, b, stream):
    p = C.cast(C.byref(plans[s].ops[a]), C.POINTER(_ffi.Op))
    _ffi.check(lib.ydet_run_ops(p, b - a, stream.cuda_stream), "ydet_run_ops")


def tail(s):
    z, _ = model.model[-1].finish(outs[s])
    det, cnt = non_max_suppression_padded(z, 0.001, 0.6, multi_label=True)
    pdet[s].copy_(det, non_blocking=True)
    pcnt[s].copy_(cnt, non_blocking=True)


def run(mode):
    done = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    ev = [(torch.cuda.Event(), torch.cuda.Event()) for _ in range(K)]
    for e in done:
        e.record()
    torch.cuda.synchronize()
    with torch.cuda.stream(streams[0]):
        done[0].record()
    with torch.no_grad():
        for k in range(K):
            s = k % S
            with torch.cuda.stream(streams[s]):
                if mode == "plain":
                    run_range(s, 0, n, streams[s])
                else:
                    run_range(s, 0, FORK, streams[s])
                    ev[k][0].record(streams[s])
                    sides[s].wait_event(ev[k][0])
                    run_range(s, h0, h0 + 1, sides[s])
                    ev[k][1].record(sides[s])
                    run_range(s, FORK, h0, streams[s])
                    run_range(s, h0 + 1, n, streams[s])
                    streams[s].wait_event(ev[k][1])
                tail(s)
                done[k + 1].record()
    torch.cuda.synchronize()
    t = [done[0].elapsed_time(e) for e in done[1:]]
    w = [(t[i + 19] - (t[i - 1] if i else 0.0)) / 20 for i in range(0, K, 20)]
    print(f"{mode:6s}: " + " ".join(f"{v:.3f}" for v in w) + f"   mean of windows 2.. {sum(w[1:]) / len(w[1:]):.3f} ms per step", flush=True)


ref = [o.clone() if torch.is_tensor(o) else o for o in [outs[0]["z"]]]
for rnd in range(3):
    run("plain")
    run("side")
torch.cuda.synchronize()
print("z unchanged:", torch.equal(ref[0], outs[0]["z"]))
