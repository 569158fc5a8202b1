#!/usr/bin/env python3
"""Calibration: what stock elementwise kernels reach on this GPU for a 550 MB fp32 copy / sigmoid / fill."""
import torch
dev = "cuda:0"
n = 64 * 25200 * 85
x = torch.randn(n, device=dev)
y = torch.empty_like(x)

def timeit(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3

for name, fn, nb in (("copy", lambda: y.copy_(x), 8 * n), ("sigmoid", lambda: torch.sigmoid(x, out=y), 8 * n),
                     ("fill", lambda: y.fill_(1.0), 4 * n), ("sum", lambda: x.sum(), 4 * n)):
    t = timeit(fn)
    print(f"{name}: {t:.0f} us  {nb / t / 1e6:.2f} TB/s")

# the size of one 80x80x128-channel bf16 map at batch 64 (a layer's input is what the layer before just wrote)
m = 64 * 80 * 80 * 128 // 2
xs, ys = x[:m], y[:m]
t = timeit(lambda: ys.copy_(xs), 50)
print(f"copy 105 MB -> 105 MB: {t:.1f} us  {8 * m / t / 1e6:.2f} TB/s")

# ping-pong: every copy reads what the copy before it wrote (does written data stay in the memory-side cache?)
a, b = x[:m], y[:m]
def pingpong():
    b.copy_(a); a.copy_(b)
t = timeit(pingpong, 25) / 2
print(f"ping-pong copy 105 MB <-> 105 MB: {t:.1f} us per copy  {8 * m / t / 1e6:.2f} TB/s")
# a chain over four buffers (420 MB working set): each copy reads the buffer written by the one before
bufs = [torch.empty(m, device=dev) for _ in range(4)]
def chain():
    for i in range(4):
        bufs[(i + 1) % 4].copy_(bufs[i])
t = timeit(chain, 15) / 4
print(f"chain of copies over 4 x 105 MB: {t:.1f} us per copy  {8 * m / t / 1e6:.2f} TB/s")
