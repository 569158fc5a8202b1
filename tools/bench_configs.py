#!/usr/bin/env python3
"""Throughput of the other benchmark configurations (forward + decode + per-class NMS of every image), one JSON line
each: YOLOv2-D19 416 bs32, YOLOv3-SPP 608 bs64, YOLOv4-CSPDarknet53 608 bs64, YOLOv3-tiny 416 bs1 / bs64.
The headline YOLOv5s line is bench.py's; this tool only adds the figures of the other models.

    python tools/bench_configs.py [--steps 20] [--compute bf16]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GFLOP = {"v3tiny": 6.18, "v2d19": 29.36, "v3spp": 140.90, "v4": 98.69, "v2r50": None, "v2slim": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--compute", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--only", default="", help="run only the configurations whose model tag is this (e.g. v4)")
    args = ap.parse_args()
    from helpers import build_fa
    from yolo_amd import engine
    from yolo_amd.synthetic import images
    engine.set_compute_dtype(args.compute)
    dev = torch.device("cuda:0")
    for tag, size, batch in (("v3tiny", 416, 1), ("v3tiny", 416, 64), ("v2d19", 416, 32), ("v3spp", 608, 64), ("v4", 608, 64),
                             ("v2r50", 416, 32), ("v2slim", 416, 32)):
        if args.only and tag != args.only:
            continue
        net = build_fa(tag, size).to(dev)
        x = images(batch, size, seed=1, kind="randn").to(dev)
        with torch.no_grad():
            for _ in range(args.warmup):
                net.detect_batch(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out = net.detect_batch(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
        gf = GFLOP[tag]
        print(json.dumps({"config": f"{tag} {size}x{size} bs{batch}", "dtype": args.compute, "images_per_s": round(batch / dt, 1),
                          "ms_per_step": round(dt * 1e3, 3),
                          "conv_tflops_lower_bound": None if gf is None else round(gf * batch / dt / 1e3, 1),
                          "kept_image0": int(out[1][0].sum())}), flush=True)
        del net


if __name__ == "__main__":
    main()
