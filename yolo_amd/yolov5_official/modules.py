"""Building blocks of the YOLOv5-style model family, with the class names, constructor signatures
and state_dict keys of the reference (yolov5_official/modules.py).

Every block has two faces:
  forward(x)    stock torch ops on NCHW tensors -- used in training mode and on CPU tensors (for
                example the 256x256 dry run that derives the Detect strides at construction time);
  lower(g, x)   emits the block into an `engine.Graph` as fused HIP launches on NHWC slices -- this is
                what runs for CUDA tensors in eval mode.
"""
import math

import torch
import torch.nn as nn

from .. import _ffi


def autopad(k, p=None):
    """'same' padding for odd kernels (reference yolov5_official/modules.py:12-16)."""
    if p is not None:
        return p
    return k // 2 if isinstance(k, int) else [v // 2 for v in k]


def activation_code(act):
    """Map an activation module to (ydet_act, slope)."""
    if act is None or isinstance(act, nn.Identity):
        return _ffi.ACT_NONE, 0.0
    if isinstance(act, nn.SiLU):
        return _ffi.ACT_SILU, 0.0
    if isinstance(act, nn.LeakyReLU):
        return _ffi.ACT_LEAKY, float(act.negative_slope)
    if isinstance(act, nn.ReLU):          # the leaky form with slope 0
        return _ffi.ACT_LEAKY, 0.0
    if type(act).__name__ == "SiLU":  # hand-written x*sigmoid(x) modules
        return _ffi.ACT_SILU, 0.0
    raise NotImplementedError(f"activation {type(act).__name__} has no fused epilogue")


class Conv(nn.Module):
    """Conv2d(bias=False) + BatchNorm2d + SiLU (reference yolov5_official/modules.py:24-41)."""
    default_act = staticmethod(lambda: nn.SiLU(inplace=True))

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p), groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = self.default_act() if act is True else None

    def forward(self, x):
        y = self.bn(self.conv(x))
        return y if self.act is None else self.act(y)

    def fuseforward(self, x):
        return self.act(self.conv(x))

    def lower(self, g, x, out=None, res=None, res_mode=_ffi.RES_AFTER_ACT):
        act, slope = activation_code(self.act)
        return g.conv(x, self.conv, getattr(self, "bn", None), act=act, slope=slope, res=res, res_mode=res_mode,
                      out=out)


def _cb(m):
    """(convolution, BatchNorm or None) of a Conv block: Model.fuse() folds the BatchNorm into the convolution's
    weights and bias and deletes the attribute."""
    return m.conv, getattr(m, "bn", None)


def DWConv(c1, c2, k=1, s=1, act=True):
    return Conv(c1, c2, k, s, g=math.gcd(c1, c2), act=act)


class Bottleneck(nn.Module):
    """1x1 -> 3x3 with optional identity shortcut: x + cv2(cv1(x)) (reference :208-218)."""

    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y

    def lower(self, g, x, out=None):
        a1, a2 = activation_code(self.cv1.act), activation_code(self.cv2.act)
        if a1 == a2 and a1[0] != _ffi.ACT_NONE:
            return g.bottleneck(x, _cb(self.cv1), _cb(self.cv2), *a1, add=self.add, out=out)
        return self.cv2.lower(g, self.cv1.lower(g, x), out=out, res=x if self.add else None)


def _lower_chain(g, mods, x):
    for m in mods:
        x = m.lower(g, x)
    return x


class BottleneckCSP(nn.Module):
    """CSP bottleneck whose concat is followed by BatchNorm + LeakyReLU(0.1) (reference :221-239).
    On the GPU path each half of that BatchNorm is folded into the plain 1x1 conv that produces it."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = nn.Conv2d(c1, c_, 1, 1, bias=False)
        self.cv3 = nn.Conv2d(c_, c_, 1, 1, bias=False)
        self.cv4 = Conv(2 * c_, c2, 1, 1)
        self.bn = nn.BatchNorm2d(2 * c_)
        self.act = nn.LeakyReLU(0.1, inplace=True)
        self.m = nn.Sequential(*[Bottleneck(c_, c_, shortcut, g, e=1.0) for _ in range(n)])

    def forward(self, x):
        y1 = self.cv3(self.m(self.cv1(x)))
        y2 = self.cv2(x)
        return self.cv4(self.act(self.bn(torch.cat((y1, y2), dim=1))))

    def lower(self, g, x):
        c_ = self.cv3.out_channels
        act, slope = activation_code(self.act)
        y1 = g.conv(_lower_chain(g, self.m, self.cv1.lower(g, x)), self.cv3, self.bn, (0, c_), act=act, slope=slope)
        y2 = g.conv(x, self.cv2, self.bn, (c_, 2 * c_), act=act, slope=slope)
        return self.cv4.lower(g, g.cat([y1, y2]))


class C3(nn.Module):
    """CSP bottleneck with three convolutions (reference :242-253)."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*[Bottleneck(c_, c_, shortcut, g, e=1.0) for _ in range(n)])

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), dim=1))

    def lower(self, g, x):
        c_ = self.cv1.conv.out_channels
        a1, a2 = activation_code(self.cv1.act), activation_code(self.cv2.act)
        chain = isinstance(self.m, nn.Sequential)      # (C3SPP puts a single SPP block there)
        if chain and a1 == a2 and c_ % g.vec == 0 and all(isinstance(b, Bottleneck) and b.cv2.conv.out_channels == c_ for b in self.m):
            # cv1 and cv2 read the same input: one convolution writes [cv1(x) | cv2(x)] straight into the concat buffer;
            # the bottleneck chain then updates the first half in place (its residual is read by the thread that
            # overwrites it), so the concat never exists as an operation.
            if c_ in (32, 64) and g.dt == _ffi.BF16 and x.C % 32 == 0:
                # 32- and 64-channel bottlenecks run as one kernel that must not work in place (it reads a halo): the sibling
                # conv sends cv1(x) to a buffer of its own and cv2(x) to what becomes the second half of the concat
                # buffer; the last bottleneck's output becomes the first half
                _, (y, y2) = g.conv_siblings(x, [_cb(self.cv1), _cb(self.cv2)], *a1, separate=True)
                for b in list(self.m)[:-1]:
                    y = b.lower(g, y)
                last = list(self.m)[-1]
                la = (activation_code(last.cv1.act), activation_code(last.cv2.act), activation_code(self.cv3.act))
                if c_ == 32 and la[0] == la[1] == la[2] == a1 and a1[0] != _ffi.ACT_NONE:
                    # the last bottleneck, the concat and cv3 as one kernel (32-channel stage only)
                    out = g.c3_tail(y, _cb(last.cv1), _cb(last.cv2), y2, _cb(self.cv3), *a1, add=last.add)
                    if out is not None:
                        return out
                y = last.lower(g, y)
                return self.cv3.lower(g, g.cat([y, y2]))
            cat, (y, _) = g.conv_siblings(x, [_cb(self.cv1), _cb(self.cv2)], *a1)
            for b in self.m:
                b.lower(g, y, out=y)
            return self.cv3.lower(g, cat)
        y1 = _lower_chain(g, self.m if chain else [self.m], self.cv1.lower(g, x))
        y2 = self.cv2.lower(g, x)
        return self.cv3.lower(g, g.cat([y1, y2]))


class SPP(nn.Module):
    """Spatial pyramid pooling: cv2(cat(x, pool5, pool9, pool13)) with x = cv1(input) (reference :271-282)."""

    def __init__(self, c1, c2, k=(5, 9, 13)):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * (len(k) + 1), c2, 1, 1)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])

    def forward(self, x):
        x = self.cv1(x)
        return self.cv2(torch.cat([x] + [m(x) for m in self.m], 1))

    def lower(self, g, x):
        ks = tuple(int(m.kernel_size) for m in self.m)
        if ks != (5, 9, 13):
            raise NotImplementedError(f"SPP kernels {ks}: the fused pooling kernel implements (5, 9, 13)")
        return self.cv2.lower(g, g.spp(self.cv1.lower(g, x)))


class C3SPP(C3):
    """C3 whose middle is one SPP block instead of the bottleneck chain (reference :262-268)."""

    def __init__(self, c1, c2, k=(5, 9, 13), n=1, shortcut=True, g=1, e=0.5):
        super().__init__(c1, c2, n=n, shortcut=shortcut, g=g, e=e)
        c_ = int(c2 * e)
        self.m = SPP(c_, c_, k)


class Focus(nn.Module):
    """Space-to-depth by 2 then Conv; channel order (0,0),(1,0),(0,1),(1,1) (reference :285-293)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        self.conv = Conv(c1 * 4, c2, k, s, p, g, act)

    @staticmethod
    def slice4(x):
        return torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)

    def forward(self, x):
        return self.conv(self.slice4(x))

    def lower(self, g, x):
        # as the first layer the slicing is folded into the NCHW->NHWC edge kernel (see Model._lower)
        return self.conv.lower(g, g.space_to_depth(x, 1))


class Contract(nn.Module):
    """Spatial -> channel rearrangement (reference :296-306); torch-only, not on the accelerated path."""

    def __init__(self, gain=2):
        super().__init__()
        self.gain = gain

    def forward(self, x):
        n, c, h, w = x.size()
        s = self.gain
        x = x.view(n, c, h // s, s, w // s, s).permute(0, 3, 5, 1, 2, 4).contiguous()
        return x.view(n, c * s * s, h // s, w // s)


class Expand(nn.Module):
    """Channel -> spatial rearrangement (reference :309-319); torch-only."""

    def __init__(self, gain=2):
        super().__init__()
        self.gain = gain

    def forward(self, x):
        n, c, h, w = x.size()
        s = self.gain
        x = x.view(n, s, s, c // s ** 2, h, w).permute(0, 3, 4, 1, 5, 2).contiguous()
        return x.view(n, c // s ** 2, h * s, w * s)


class Concat(nn.Module):
    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x):
        return torch.cat(x, self.d)

    def lower(self, g, xs):
        if self.d != 1:
            raise NotImplementedError("only channel concatenation is lowered")
        return g.cat(list(xs))


def lower_upsample(g, m, x):
    """nn.Upsample(scale_factor=2) in 'nearest' or 'bilinear' (align_corners=True) mode."""
    sf = m.scale_factor
    if m.size is not None or float(sf if not isinstance(sf, (tuple, list)) else sf[0]) != 2.0:
        raise NotImplementedError("only scale_factor=2 upsampling is lowered")
    if m.mode == "nearest":
        return g.upsample2x(x, 0)
    if m.mode == "bilinear" and m.align_corners:
        return g.upsample2x(x, 1)
    raise NotImplementedError(f"upsample mode {m.mode} (align_corners={m.align_corners})")
