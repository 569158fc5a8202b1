#!/usr/bin/env python3
"""Random 1x1 convolutions through ydet_conv2d_nhwc twice -- in this process (streaming kernel where it applies) and in a
child process started with YDET_PW=0 (tiled kernel only) -- outputs compared bit for bit.  Shapes cover input slices of
wider buffers, two destinations, padded channel counts, ragged pixel counts, upsampled sources, all activations.

    python tools/fuzz_pw.py [cases] [seed]
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
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        cin = int(rng.choice([128, 256]))
        cout = int(rng.choice([72, 128, 136, 200, 256, 264, 384]))
        H, W = int(rng.randint(20, 60)) * 2, int(rng.randint(20, 60)) * 2
        N = int(np.ceil(65536 / (H * W))) + int(rng.randint(0, 3))
        x_extra = int(rng.choice([0, 0, 64]))          # input is a slice of a wider buffer
        split = bool(rng.randint(0, 2)) and cout % 16 == 0
        up = int(rng.choice([0, 0, 32, 64, cin // 2]))
        act = int(rng.choice([_ffi.ACT_SILU, _ffi.ACT_LEAKY, _ffi.ACT_NONE]))
        post = bool(rng.randint(0, 4) == 0)
        out.append(dict(N=N, H=H, W=W, cin=cin, cout=cout, x_extra=x_extra, split=split, up=up, act=act, post=post))
    return out


def run(cs, seed):
    lib = _ffi.lib()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    res = []
    for ci, c in enumerate(cs):
        g = torch.Generator().manual_seed(seed * 1000 + ci)
        N, H, W, cin, cout = c["N"], c["H"], c["W"], c["cin"], c["cout"]
        x_ld = cin + c["x_extra"]
        x = torch.randn(N, H, W, x_ld, generator=g).to(dev).bfloat16()
        cpad = (cout + 127) // 128 * 128
        w = (torch.randn(cpad, cin, generator=g) * (2.0 / cin) ** 0.5).to(dev).bfloat16()
        w[cout:] = 0
        b = (torch.randn(cpad, generator=g) * 0.1).to(dev)
        b[cout:] = 0
        d = _ffi.ConvDesc()
        d.x, d.w, d.bias = x.data_ptr(), w.data_ptr(), b.data_ptr()
        d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad, d.Ho, d.Wo = N, H, W, cin, cout, 1, 1, 1, 0, H, W
        d.x_ld, d.x_coff = x_ld, c["x_extra"] // 2 // 8 * 8
        d.kp, d.cout_pad, d.act, d.slope = cin, cpad, c["act"], 0.1
        d.res_mode = _ffi.RES_NONE
        d.in_dtype = d.out_dtype = _ffi.BF16
        keep = [x, w, b]
        if c["split"]:
            ysp = cout // 2 // 8 * 8
            y = torch.zeros(N, H, W, ysp + 8, device=dev, dtype=torch.bfloat16)
            y2 = torch.zeros(N, H, W, cout - ysp + 16, device=dev, dtype=torch.bfloat16)
            d.y, d.y_ld, d.y_coff = y.data_ptr(), ysp + 8, 8
            d.y2, d.y2_ld, d.y2_coff, d.y_split = y2.data_ptr(), cout - ysp + 16, 8, ysp
            outs = [y, y2]
        else:
            y_ld = (cout + 7) // 8 * 8
            y = torch.zeros(N, H, W, y_ld, device=dev, dtype=torch.bfloat16)
            d.y, d.y_ld, d.y_coff = y.data_ptr(), y_ld, 0
            if cout % 8:
                d.cout_store = y_ld
            outs = [y]
        if c["up"]:
            xu = torch.randn(N, H // 2, W // 2, c["up"] + 8, generator=g).to(dev).bfloat16()
            d.x_up, d.x_up_ld, d.x_up_coff, d.x_up_C = xu.data_ptr(), c["up"] + 8, 8, c["up"]
            keep.append(xu)
        if c["post"]:
            ps = (torch.rand(cpad, generator=g) + 0.5).to(dev)
            pb = (torch.randn(cpad, generator=g) * 0.1).to(dev)
            d.post_scale, d.post_shift, d.post_act, d.post_slope = ps.data_ptr(), pb.data_ptr(), _ffi.ACT_LEAKY, 0.1
            keep += [ps, pb]
        rc = lib.ydet_conv2d_nhwc(C.byref(d), stream)
        if rc != 0:
            raise RuntimeError(f"case {ci} {c}: {lib.ydet_last_error().decode()}")
        torch.cuda.synchronize()
        res.append([o.view(torch.int16).cpu().numpy() for o in outs])
    return res


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cs = cases(n, seed)
    if os.environ.get("FUZZ_CHILD"):
        res = run(cs, seed)
        np.savez(os.environ["FUZZ_CHILD"], **{f"c{i}_{j}": a for i, r in enumerate(res) for j, a in enumerate(r)})
        return
    path = "/tmp/fuzz_pw_ref.npz"
    env = dict(os.environ, FUZZ_CHILD=path, YDET_PW="0")
    subprocess.run([sys.executable, os.path.abspath(__file__), str(n), str(seed)], check=True, env=env)
    ref = np.load(path)
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
