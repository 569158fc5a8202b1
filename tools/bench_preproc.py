#!/usr/bin/env python3
"""Timing of the device pre-processing and of the uint8 input path (one JSON line each):
    python tools/bench_preproc.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timed(fn, n=20, warm=3):
    """Wall-clock per call (host work included)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def timed_gpu(fn, n=20):
    """GPU time per call between two events on the launch stream, calls queued back to back."""
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(n):
        a.record()
        fn()
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b) * 1e-3)
    return best


def main():
    from helpers import build_v5s
    from yolo_amd import engine
    from yolo_amd.preprocess import letterbox_batch
    from yolo_amd.data.transformers import BaseTransform
    dev = "cuda:0"
    rng = np.random.default_rng(0)
    for shape in ((720, 1280), (480, 640), (1080, 1920)):
        imgs = [torch.from_numpy(rng.integers(0, 256, size=shape + (3,), dtype=np.uint8)).to(dev) for _ in range(64)]
        src = 64 * shape[0] * shape[1] * 3
        for name, fn, out_b in (("letterbox->fp32 NCHW", lambda: letterbox_batch(imgs, (640, 640), auto=False), 64 * 3 * 640 * 640 * 4),
                                ("letterbox->uint8 NCHW", lambda: letterbox_batch(imgs, (640, 640), auto=False, scale=False), 64 * 3 * 640 * 640),
                                ("base_transform 416 fp32", lambda: BaseTransform(416).batch(imgs), 64 * 3 * 416 * 416 * 4)):
            dt, dg = timed(fn), timed_gpu(fn)
            print(json.dumps({"op": name, "source": f"64 x {shape[0]}x{shape[1]} uint8 HWC (resident)", "ms_wall": round(dt * 1e3, 3),
                              "ms_gpu_best": round(dg * 1e3, 3), "images_per_s": round(64 / dt),
                              "GBps_src_plus_out_gpu": round((src + out_b) / dg / 1e9, 1)}), flush=True)
    engine.set_compute_dtype("bf16")
    m = build_v5s(80).to(dev)
    u8 = torch.randint(0, 256, (64, 3, 640, 640), dtype=torch.uint8, device=dev)
    f32 = u8.float() / 255
    with torch.no_grad():
        for name, fn in (("forward, fp32 image", lambda: m(f32)), ("forward, uint8 image", lambda: m(u8)),
                         ("img.float()/255 + forward (what val.py does)", lambda: m(u8.float() / 255.0))):
            dt = timed(fn)
            print(json.dumps({"op": name, "ms": round(dt * 1e3, 3), "images_per_s": round(64 / dt)}), flush=True)


if __name__ == "__main__":
    main()
