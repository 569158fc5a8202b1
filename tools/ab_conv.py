#!/usr/bin/env python3
"""A/B timing of ydet_conv2d_nhwc between builds of libydet.so, in ONE process on ONE device.

    python tools/ab_conv.py [--libs yolo_amd/libydet.so tools/bin/libydet_head.so ...] [--only 12,18] [--rounds 30]

For every convolution shape of the list (the conv_dma launches of YOLOv5s 640 at batch 64 by default) each library
runs the same descriptor on the same buffers in interleaved rounds; reported: median and minimum microseconds per
library (hipEvents around a burst of launches), TFLOP/s of the median, and the largest output difference against
the first library.  Inputs are random (not zeros: the clock the chip holds depends on the data).
"""
import argparse
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolo_amd import _ffi  # noqa: E402

# (N, H, W, Cin, Cout, k, s, residual, act)  -- YOLOv5s 640 bs64 launches that go to conv_dma_kernel
V5S = [
    (64, 160, 160, 64, 128, 3, 2, 0),
    (64, 80, 80, 128, 128, 1, 1, 0),
    (64, 80, 80, 128, 256, 3, 2, 0),
    (64, 40, 40, 256, 256, 1, 1, 0),
    (64, 40, 40, 128, 128, 1, 1, 0),
    (64, 40, 40, 128, 128, 3, 1, 1),
    (64, 40, 40, 256, 512, 3, 2, 0),
    (64, 20, 20, 512, 256, 1, 1, 0),
    (64, 20, 20, 1024, 512, 1, 1, 0),
    (64, 20, 20, 512, 512, 1, 1, 0),
    (64, 20, 20, 256, 256, 1, 1, 0),
    (64, 20, 20, 256, 256, 3, 1, 1),
    (64, 40, 40, 512, 256, 1, 1, 0),
    (64, 40, 40, 256, 128, 1, 1, 0),
    (64, 80, 80, 256, 128, 1, 1, 0),
    (64, 80, 80, 128, 128, 3, 2, 0),
    (64, 40, 40, 256, 256, 3, 2, 0),
    (64, 320, 320, 32, 64, 3, 2, 0),
]


def load(path):
    h = C.CDLL(os.path.abspath(path))
    h.ydet_conv2d_nhwc.restype = C.c_int
    h.ydet_conv2d_nhwc.argtypes = [C.POINTER(_ffi.ConvDesc), C.c_void_p]
    h.ydet_last_error.restype = C.c_char_p
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", default=[os.path.join(ROOT, "yolo_amd", "libydet.so")])
    ap.add_argument("--only", default="")
    ap.add_argument("--rounds", type=int, default=25)
    ap.add_argument("--burst", type=int, default=8)
    ap.add_argument("--shapes", default="", help="extra shapes N,H,W,Cin,Cout,k,s,res;...")
    ap.add_argument("--cold", type=int, default=0, help="megabytes written between timed launches (one launch per "
                    "interval then): inside a network a layer's weights come from HBM and its input from whatever the "
                    "layer before left in the caches, not from eight runs of itself")
    ap.add_argument("--tiles", nargs="*", default=[], help="YDET_CONV_TILE per library ('-' for none): each library reads "
                    "the variable once, at its first launch, so copies of one build can run different forced tiles")
    ap.add_argument("--pw", nargs="*", default=[], help="YDET_PW per library ('-' for none; read once per library, as above): "
                    "0 keeps the 1x1 layers away from the streaming kernel of conv_pw.hip")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    libs = [(os.path.basename(p).replace("libydet", "").replace(".so", "") or "cur", load(p)) for p in args.libs]
    shapes = list(V5S)
    if args.shapes:
        shapes = [tuple(int(v) for v in s.split(",")) for s in args.shapes.split(";") if s]
    if args.only:
        shapes = [shapes[int(i)] for i in args.only.split(",")]
    flush = torch.zeros(max(args.cold, 1) * (1 << 18), device=dev) if args.cold else None
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(1)
    print("shape".ljust(38) + "".join(f"{n:>26s}" for n, _ in libs))
    tot = [0.0] * len(libs)
    for shp in shapes:
        N, H, W, Cin, Cout, k, s, res = shp
        p = k // 2
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        kp = (k * k * Cin + 31) // 32 * 32
        cpad = (Cout + 127) // 128 * 128
        x = torch.randn(N, H, W, Cin, generator=g).to(dev).bfloat16()
        w = (torch.randn(cpad, kp, generator=g) * (2.0 / (k * k * Cin)) ** 0.5).to(dev).bfloat16()
        w[Cout:] = 0
        b = (torch.randn(cpad, generator=g) * 0.1).to(dev)
        r = torch.randn(N, Ho, Wo, Cout, generator=g).to(dev).bfloat16() if res else None
        ys = [torch.zeros(N, Ho, Wo, Cout, device=dev, dtype=torch.bfloat16) for _ in libs]
        ds = []
        for y in ys:
            d = _ffi.ConvDesc()
            d.x, d.w, d.bias, d.y = x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr()
            d.res = r.data_ptr() if res else None
            d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad, d.Ho, d.Wo = N, H, W, Cin, Cout, k, k, s, p, Ho, Wo
            d.x_ld, d.x_coff, d.y_ld, d.y_coff, d.res_ld, d.res_coff = Cin, 0, Cout, 0, Cout if res else 0, 0
            d.kp, d.cout_pad, d.act, d.slope = kp, cpad, _ffi.ACT_SILU, 0.0
            d.res_mode = _ffi.RES_AFTER_ACT if res else _ffi.RES_NONE
            d.in_dtype = d.out_dtype = _ffi.BF16
            ds.append(d)
        for li, (name, h) in enumerate(libs):      # first launch of every library under its own environment
            tile = args.tiles[li] if li < len(args.tiles) else "-"
            if tile != "-":
                os.environ["YDET_CONV_TILE"] = tile
            else:
                os.environ.pop("YDET_CONV_TILE", None)
            pw = args.pw[li] if li < len(args.pw) else "-"
            os.environ.pop("YDET_PW", None)
            if pw != "-":
                os.environ["YDET_PW"] = pw
            if h.ydet_conv2d_nhwc(ds[li], stream) != 0:
                raise RuntimeError(f"{name}: {h.ydet_last_error().decode()}")
        torch.cuda.synchronize()
        times = [[] for _ in libs]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        burst = 1 if args.cold else args.burst
        for rnd in range(args.rounds + 3):
            for li, (name, h) in enumerate(libs):
                if args.cold:
                    flush.add_(1.0)
                    x.add_(0)             # the input as the layer before would have left it: last thing written
                ev[0].record()
                for _ in range(burst):
                    rc = h.ydet_conv2d_nhwc(ds[li], stream)
                    if rc != 0:
                        raise RuntimeError(f"{name}: {h.ydet_last_error().decode()}")
                ev[1].record()
                ev[1].synchronize()
                if rnd >= 3:
                    times[li].append(ev[0].elapsed_time(ev[1]) * 1e3 / burst)
        flop = 2.0 * N * Ho * Wo * Cout * Cin * k * k
        line = f"{str(shp):38s}"
        for li in range(len(libs)):
            med, mn = statistics.median(times[li]), min(times[li])
            diff = (ys[li].float() - ys[0].float()).abs().max().item()
            tot[li] += med
            line += f"  {med:7.1f} {mn:6.1f} {flop / med / 1e6:5.0f}T d{diff:.0e}"
        print(line, flush=True)
        for li, (name, h) in enumerate(libs):
            if hasattr(h, "ydet_debug_patch_trace"):
                buf = (C.c_ulonglong * 128)()
                torch.cuda.synchronize()
                h.ydet_debug_patch_trace(buf)
                print(f"   [{name}] workgroup 0, first tile, cycles per wave: K loop = work + waits for the ring + barriers | "
                      "K loop, wait for next patch, patch to LDS, epilogue, barrier")
                for w in range(8):
                    b = buf[w * 16:w * 16 + 8]
                    print(f"     wave {w}: {b[0]:7d} {b[1]:7d} {b[2]:7d} | " + " ".join(f"{v:7d}" for v in b[3:8]))
    print("total (median us)".ljust(38) + "".join(f"{t:26.1f}" for t in tot))


if __name__ == "__main__":
    main()
