#!/usr/bin/env python3
"""Three pipelined streams with an enforced phase offset: step k may start only when step k - 1 (on the next stream)
has passed op CUT of its launch list.  Reports ms per step over windows of 20 steps, free-running against staggered.

    python tools/exp_stagger.py [CUT ...]
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

cuts = [int(v) for v in sys.argv[1:]] or [7]
dev = torch.device("cuda:0")
lib = _ffi.lib()
engine.set_compute_dtype("bf16")
model = Model()
init_deterministic(model, seed=0, head_convs=list(model.model[-1].m))
model = model.to(dev).eval()
x = images(64, 640, seed=100).to(dev)
S, K = 3, 120
streams = [torch.cuda.Stream(dev) for _ in range(S)]
pdet = [torch.empty(64, 300, 6, pin_memory=True) for _ in range(S)]
pcnt = [torch.empty(64, dtype=torch.int32, pin_memory=True) for _ in range(S)]
plans, outs = [], []
with torch.no_grad():
    for s in range(S):
        with torch.cuda.stream(streams[s]):
            plan = engine.build_plan(dev, "bf16", 64, lambda g, n: model._lower(g, n, *x.shape[1:]))
            plans.append(plan)
            outs.append(plan.run(x))          # pointers patched; the same output buffers serve every step of this stream
torch.cuda.synchronize()
n = len(plans[0].ops)


def run_range(s, a, b):
    p = C.cast(C.byref(plans[s].ops[a]), C.POINTER(_ffi.Op))
    _ffi.check(lib.ydet_run_ops(p, b - a, streams[s].cuda_stream), "ydet_run_ops")


def tail(s):
    z, _ = model.model[-1].finish(outs[s])
    det, cnt = non_max_suppression_padded(z, 0.001, 0.6, multi_label=True)
    pdet[s].copy_(det, non_blocking=True)
    pcnt[s].copy_(cnt, non_blocking=True)


def run(cut):
    marks = [torch.cuda.Event() for _ in range(K)]
    done = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    for e in done:
        e.record()                            # (events exist before the timed loop)
    torch.cuda.synchronize()
    with torch.cuda.stream(streams[0]):
        done[0].record()
    with torch.no_grad():
        for k in range(K):
            s = k % S
            with torch.cuda.stream(streams[s]):
                if cut > 0:
                    if k > 0:
                        streams[s].wait_event(marks[k - 1])
                    run_range(s, 0, cut)
                    marks[k].record()
                    run_range(s, cut, n)
                else:
                    run_range(s, 0, n)
                tail(s)
                done[k + 1].record()
    torch.cuda.synchronize()
    t = [done[0].elapsed_time(e) for e in done[1:]]
    w = [(t[i + 19] - (t[i - 1] if i else 0.0)) / 20 for i in range(0, K, 20)]
    print(f"cut {cut:2d}: " + " ".join(f"{v:.3f}" for v in w) + f"   mean of the last five windows {sum(w[1:]) / len(w[1:]):.3f} ms per step", flush=True)


for rnd in range(2):
    run(0)
    for c in cuts:
        run(c)
