#!/usr/bin/env python3
"""Random wide convolutions (1x1 and 3x3, stride 1 and 2, Cin a multiple of 64, 256 output channels and more) through
ydet_conv2d_nhwc twice -- in this process (software-pipelined 256 x 256 / 256 x 192 tiles where they apply) and in a child
process started with YDET_CONV_PIPE=0 (sixteen-wave 256 x 256 tile or 128-wide tiles) -- outputs compared bit for bit.
Shapes cover ragged pixel and channel tiles, input / residual / output slices of wider buffers, the three residual modes,
all activations, the second epilogue stage and padded stores.

    python tools/fuzz_wide.py [cases] [seed]
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolo_amd import _ffi  # noqa: E402


def cases(n, seed):
    from yolo_amd import engine        # (imported late: the parent starts its child before anything can touch the GPU)
    rng = np.random.RandomState(seed)
    out = []
    saved = os.environ.pop("YDET_CONV_PIPE", None)     # (the shape rule, not this process's switch: parent and child must agree)
    try:
        return _cases(n, rng, engine)
    finally:
        if saved is not None:
            os.environ["YDET_CONV_PIPE"] = saved


def _cases(n, rng, engine):
    out = []
    while len(out) < n:
        cin = int(rng.choice([64, 128, 192, 256, 512, 1024]))
        cout = int(rng.choice([256, 264, 392, 512, 700, 1024]))
        k = int(rng.choice([1, 3]))
        s = int(rng.choice([1, 1, 2])) if k == 3 else 1
        if k * k * cin < 512:
            continue
        H, W = int(rng.randint(5, 60)), int(rng.randint(5, 60))
        Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        N = int(rng.randint(1, 5)) * max(1, int(np.ceil(20000 / (Ho * Wo))))
        M = N * Ho * Wo
        if N * H * W * cin * 2 > 2.5e8 or M * cout * 2 > 2.5e8:
            continue
        res = int(rng.choice([0, 1, 1, 2]))
        lab = engine.conv_kernel_label(cin, cout, k, k, k // 2, _ffi.BF16, _ffi.BF16, M=M, stride=s, res=res != 0)
        if not lab.startswith("conv_dma<256x"):
            continue
        out.append(dict(N=N, H=H, W=W, cin=cin, cout=cout, k=k, s=s, x_extra=int(rng.choice([0, 0, 64])), y_extra=int(rng.choice([0, 0, 32])),
                        res=res, act=int(rng.choice([_ffi.ACT_SILU, _ffi.ACT_LEAKY, _ffi.ACT_NONE])), post=bool(rng.randint(0, 5) == 0), label=lab))
    return out


def run(cs, seed):
    lib = _ffi.lib()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    res = []
    for ci, c in enumerate(cs):
        g = torch.Generator(device=dev).manual_seed(seed * 1000 + ci)      # (on the device: the host would take seconds per case)
        rn = lambda *shape: torch.randn(*shape, generator=g, device=dev)
        N, H, W, cin, cout, k, st = c["N"], c["H"], c["W"], c["cin"], c["cout"], c["k"], c["s"]
        Ho, Wo = (H + 2 * (k // 2) - k) // st + 1, (W + 2 * (k // 2) - k) // st + 1
        x_ld = cin + c["x_extra"]
        x = rn(N, H, W, x_ld).bfloat16()
        cpad = (cout + 127) // 128 * 128
        w = (rn(cpad, k * k * cin) * (2.0 / (k * k * cin)) ** 0.5).bfloat16()
        w[cout:] = 0
        b = rn(cpad) * 0.1
        b[cout:] = 0
        d = _ffi.ConvDesc()
        d.x, d.w, d.bias = x.data_ptr(), w.data_ptr(), b.data_ptr()
        d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad, d.Ho, d.Wo = N, H, W, cin, cout, k, k, st, k // 2, Ho, Wo
        d.x_ld, d.x_coff = x_ld, c["x_extra"] // 2 // 8 * 8
        d.kp, d.cout_pad, d.act, d.slope = k * k * cin, cpad, c["act"], 0.1
        d.res_mode = _ffi.RES_NONE
        d.in_dtype = d.out_dtype = _ffi.BF16
        keep = [x, w, b]
        cst = (cout + 7) // 8 * 8
        y_ld = cst + c["y_extra"]
        y = torch.zeros(N, Ho, Wo, y_ld, device=dev, dtype=torch.bfloat16)
        d.y, d.y_ld, d.y_coff = y.data_ptr(), y_ld, c["y_extra"] // 2 // 8 * 8
        if cout % 8:
            d.cout_store = cst
        if c["res"]:
            r_ld = cst + 16
            r = rn(N, Ho, Wo, r_ld).bfloat16()
            d.res, d.res_ld, d.res_coff = r.data_ptr(), r_ld, 8
            d.res_mode = _ffi.RES_AFTER_ACT if c["res"] == 1 else _ffi.RES_BEFORE_ACT
            keep.append(r)
        if c["post"]:
            ps = rn(cpad).abs() + 0.5
            pb = rn(cpad) * 0.1
            d.post_scale, d.post_shift, d.post_act, d.post_slope = ps.data_ptr(), pb.data_ptr(), _ffi.ACT_LEAKY, 0.1
            keep += [ps, pb]
        if os.environ.get("FUZZ_VERBOSE"):
            print(f"  case {ci}: {c}", flush=True)
        rc = lib.ydet_conv2d_nhwc(C.byref(d), stream)
        if rc != 0:
            raise RuntimeError(f"case {ci} {c}: {lib.ydet_last_error().decode()}")
        torch.cuda.synchronize()
        res.append([y.view(torch.int16).cpu().numpy()])
        if ci % 10 == 9:
            print(f"  {ci + 1} cases run", flush=True)
    return res


def main():
    if os.environ.get("FUZZ_VERBOSE"):
        import faulthandler
        faulthandler.dump_traceback_later(40, exit=False)      # (where a silent run is stuck)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if os.environ.get("FUZZ_CHILD"):
        cs = cases(n, seed)
        res = run(cs, seed)
        np.savez(os.environ["FUZZ_CHILD"], **{f"c{i}_{j}": a for i, r in enumerate(res) for j, a in enumerate(r)})
        return
    path = "/tmp/fuzz_wide_ref.npz"
    env = dict(os.environ, FUZZ_CHILD=path, YDET_CONV_PIPE="0")
    subprocess.run([sys.executable, os.path.abspath(__file__), str(n), str(seed)], check=True, env=env)
    ref = np.load(path)
    cs = cases(n, seed)
    res = run(cs, seed)
    bad = 0
    for i, r in enumerate(res):
        for j, a in enumerate(r):
            same = np.array_equal(a, ref[f"c{i}_{j}"])
            nz = float((a != 0).mean())
            if not same or nz < 0.3:
                bad += 1
                print("MISMATCH" if not same else "EMPTY?", i, j, cs[i], "nonzero fraction", nz)
    print(f"{n} cases, {bad} bad")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
