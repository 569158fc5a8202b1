#!/usr/bin/env python3
"""Experiment: one batch-64 forward vs two batch-32 forwards on two streams (conv stack only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import engine
from yolo_amd.synthetic import images, init_deterministic
from yolo_amd.yolov5_official.model import Model

dev = "cuda:0"
m = Model()
init_deterministic(m, seed=0, head_convs=list(m.model[-1].m))
m = m.eval().to(dev).to(torch.bfloat16)
x = images(64, 640, seed=0).to(dev)

def plan_for(n):
    g = engine.Graph(dev, "bf16")
    m._lower(g, n, 3, 640, 640)
    return g.finalize()

def timeit(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k

p64 = plan_for(64)
print("1 x 64:", round(timeit(lambda: p64.run(x)), 3), "ms")
for parts in (2, 4):
    n = 64 // parts
    plans = [plan_for(n) for _ in range(parts)]
    xs = [x[i * n:(i + 1) * n] for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def run():
        cur = torch.cuda.current_stream()
        for s, p, xi in zip(streams, plans, xs):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                p.run(xi)
        for s in streams:
            cur.wait_stream(s)
    print(f"{parts} x {n} on {parts} streams:", round(timeit(run), 3), "ms")
    def run_seq():
        for p, xi in zip(plans, xs):
            p.run(xi)
    print(f"{parts} x {n} sequential:", round(timeit(run_seq), 3), "ms")

# ---- staggered start: the second half-batch starts when the first one has finished op k ----------------------------
import ctypes as C
from yolo_amd import _ffi
lib = _ffi.lib()
n = 32
plans = [plan_for(n) for _ in range(2)]
xs = [x[:n], x[n:]]
streams = [torch.cuda.Stream() for _ in range(2)]

def run_ops(plan, lo, hi):
    arr = (_ffi.Op * (hi - lo)).from_address(C.addressof(plan.ops) + lo * C.sizeof(_ffi.Op))
    _ffi.check(lib.ydet_run_ops(arr, hi - lo, _ffi.current_stream()), "run_ops")

for p, xi in zip(plans, xs):
    p.run(xi)   # binds the input / output pointers
nops = len(plans[0].ops)
for k in (0, 5, 9, 17, 26, 34):
    def run():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        for s in streams: s.wait_stream(cur)
        with torch.cuda.stream(streams[0]):
            run_ops(plans[0], 0, k)
            ev.record()
            run_ops(plans[0], k, nops)
        with torch.cuda.stream(streams[1]):
            streams[1].wait_event(ev)
            run_ops(plans[1], 0, nops)
        for s in streams: cur.wait_stream(s)
    print(f"2 x 32, second starts after op {k}:", round(timeit(run), 3), "ms")
