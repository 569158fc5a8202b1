#!/usr/bin/env python3
"""Per-op timing table of a v2/v3/v4 detector's forward pass (events after every op, event cost not removed):
    python tools/perop_fa.py v3spp 608 64 [bf16|fp32]
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    tag, size, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    compute = sys.argv[4] if len(sys.argv) > 4 else "bf16"
    from helpers import build_fa
    from yolo_amd import _ffi, engine
    from yolo_amd.synthetic import images
    engine.set_compute_dtype(compute)
    lib = _ffi.lib()
    net = build_fa(tag, size).to("cuda:0")
    x = images(batch, size, seed=1, kind="randn").to("cuda:0")
    with torch.no_grad():
        for _ in range(3):
            net.head_maps_nhwc(x)
        plan = next(iter(net._plans().plans.values()))
        n = len(plan.ops)
        reps = 5
        t = lib.ydet_timer_create(reps * (n + 1))
        for r in range(reps):
            plan.run(x, timer=t, first_mark=r * (n + 1))
        torch.cuda.synchronize()
        acc = [0.0] * n
        buf = (C.c_float * n)()
        for r in range(reps):
            _ffi.check(lib.ydet_timer_elapsed(t, r * (n + 1), n, buf), "elapsed")
            for i in range(n):
                acc[i] += buf[i] / reps
    tot_f, tot_t = 0.0, 0.0
    print("op\tlabel\tshape\tus\tTFLOP/s\tGB/s")
    for i, info in enumerate(plan.op_info):
        us = acc[i] * 1e3
        tf = info["flops"] / us / 1e6 if info["flops"] else 0.0
        print(f"{i}\t{info['label']}\t{info.get('shape', '')}\t{us:.1f}\t{tf:.1f}\t{info['bytes'] / us / 1e3:.0f}")
        tot_f += info["flops"]
        tot_t += us
    print(f"total\t\t\t{tot_t:.1f}\t{tot_f / tot_t / 1e6:.1f}")


if __name__ == "__main__":
    main()
