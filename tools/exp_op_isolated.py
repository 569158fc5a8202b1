#!/usr/bin/env python3
"""Every op of the YOLOv5s 640 bs64 plan timed three ways on its real buffers: alone in a burst of eight, alone with the
op before it run first each time (its input as fresh as in the network), and inside the whole list (per-op events).

    python tools/exp_op_isolated.py [--ops 3,7,31]
"""
import argparse
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolo_amd import _ffi, engine  # noqa: E402
from yolo_amd.synthetic import images, init_deterministic  # noqa: E402
from yolo_amd.yolov5_official.model import Model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="")
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _ffi.lib()
    engine.set_compute_dtype("bf16")
    model = Model()
    init_deterministic(model, seed=0, head_convs=list(model.model[-1].m))
    model = model.to(dev).eval()
    x = images(args.batch, 640, seed=100).to(dev)
    plan = engine.build_plan(dev, "bf16", x.shape[0], lambda g, n: model._lower(g, n, *x.shape[1:]))
    with torch.no_grad():
        outs = plan.run(x)       # patches every pointer; the buffers stay alive in `outs` / the arenas
    torch.cuda.synchronize()
    n = len(plan.ops)
    which = [int(i) for i in args.ops.split(",")] if args.ops else list(range(n))
    stream = torch.cuda.current_stream().cuda_stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def run(i, cnt=1):
        p = C.cast(C.byref(plan.ops[i]), C.POINTER(_ffi.Op))
        _ffi.check(lib.ydet_run_ops(p, cnt, stream), "ydet_run_ops")

    print("op label                                   burst-of-8   after-prev   (us)")
    for i in which:
        t_b, t_p = [], []
        for rnd in range(12):
            ev[0].record()
            for _ in range(8):
                run(i)
            ev[1].record()
            ev[1].synchronize()
            t_b.append(ev[0].elapsed_time(ev[1]) * 1e3 / 8)
        if i > 0:
            for rnd in range(12):
                ev[0].record()
                run(i - 1)
                ev[1].record()
                run(i)
                ev[2].record()
                ev[2].synchronize()
                t_p.append(ev[1].elapsed_time(ev[2]) * 1e3)
        info = plan.op_info[i]
        print(f"{i:2d} {info.get('label', ''):38s} {statistics.median(t_b[2:]):9.1f} {statistics.median(t_p[2:]) if t_p else 0.0:12.1f}", flush=True)


if __name__ == "__main__":
    main()
