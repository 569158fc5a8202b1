"""YOLOv5-style model: `Detect`, `Model`, `parse_model` with the interface of the reference's
yolov5_official/model.py (constructor signatures, attributes, state_dict keys, return contracts).

`Model.forward(x)`:
  * CUDA tensor + eval mode -> one pre-planned launch list of HIP kernels (engine.Plan) followed by
    the fused Detect decode kernel; returns `(z [bs, n, no], [x_i (bs, na, ny, nx, no)])`;
  * anything else (training mode, CPU tensors) -> stock torch ops, exactly the reference's module
    semantics (this is what the constructor's 256x256 stride probe uses).
"""
import math
import os
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn
import torch.nn.functional as F
import yaml

from .. import _ffi, engine, ops
from .modules import C3, C3SPP, SPP, Bottleneck, BottleneckCSP, Concat, Contract, Conv, DWConv, Expand, Focus, activation_code

DEFAULT_CFG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "yamls", "yolov5s.yaml")


class Detect(nn.Module):
    """Detection head (reference yolov5_official/model.py:18-85).

    Per level: 1x1 conv to na*no channels; eval mode decodes
    xy = (2*sigmoid - 0.5 + grid) * stride, wh = (2*sigmoid)^2 * anchor_grid, rest = sigmoid.
    `grid_transposed` reproduces the np.meshgrid variant of models/yolo.py:68-71.
    """
    stride = None
    onnx_dynamic = False
    grid_transposed = False

    def __init__(self, nc=80, anchors=(), ch=(), inplace=True, use_gpu=False):
        super().__init__()
        self.nc = nc
        self.no = nc + 5
        self.nl = len(anchors)
        self.na = len(anchors[0]) // 2
        self.grid = [torch.zeros(1)] * self.nl
        a = torch.tensor(anchors, dtype=torch.float32).view(self.nl, -1, 2)
        self.register_buffer("anchors", a)
        self.register_buffer("anchor_grid", a.clone().view(self.nl, 1, -1, 1, 1, 2))
        self.m = nn.ModuleList(nn.Conv2d(c, self.no * self.na, 1) for c in ch)
        self.inplace = inplace
        self.use_gpu = use_gpu

    # ---- stock-torch face ---------------------------------------------------------------------
    def _make_grid(self, nx=20, ny=20, use_gpu=False):
        if self.grid_transposed:
            # np.meshgrid(arange(ny), arange(nx)) has shape (nx, ny); stacked and re-read as (ny, nx) the flat
            # cell f holds (f // ny, f % ny), i.e. (row, col) on square maps
            yv, xv = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing="xy")
        else:
            yv, xv = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing="ij")
        return torch.stack((xv, yv), 2).reshape(1, 1, ny, nx, 2).float()

    def forward(self, x):
        if not self.training and x[0].is_cuda:
            return self._forward_hip(x)
        z = []
        for i in range(self.nl):
            x[i] = self.m[i](x[i])
            bs, _, ny, nx = x[i].shape
            x[i] = x[i].view(bs, self.na, self.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
            if not self.training:
                if self.grid[i].shape[2:4] != x[i].shape[2:4] or self.onnx_dynamic:
                    self.grid[i] = self._make_grid(nx, ny).to(x[i].device)
                y = x[i].sigmoid()
                xy = (y[..., 0:2] * 2. - 0.5 + self.grid[i]) * self.stride[i]
                wh = (y[..., 2:4] * 2) ** 2 * self.anchor_grid[i].view(1, self.na, 1, 1, 2)
                z.append(torch.cat((xy, wh, y[..., 4:]), -1).view(bs, -1, self.no))
        return x if self.training else (torch.cat(z, 1), x)

    # ---- HIP face --------------------------------------------------------------------------------
    def lower(self, g, xs):
        """Emit the bias-only 1x1 head convs; outputs are fp32 NHWC buffers handed to the caller.  When the head fits
        the fused kernel (na*no <= 256, no <= 128, input channels a multiple of the K step) every level is ONE launch
        that also decodes: the plan then has the outputs "z" and "row_max" next to "head{i}"."""
        bk = 32 if g.dt == _ffi.BF16 else 16
        fused = (self.na * self.no <= 256 and self.no <= 128 and self.no > 5 and self.na <= 8
                 and all(x.C % bk == 0 and m.kernel_size == (1, 1) for x, m in zip(xs, self.m)))
        heads = []
        if fused:
            rows = sum(self.na * x.H * x.W for x in xs)
            z = g.new(xs[0].N, 1, rows, self.no, dtype=_ffi.F32, external=True, ld=self.no)
            rm = g.new(xs[0].N, 1, rows, 1, dtype=_ffi.F32, external=True, ld=1)
            strides, anchors = self._host_constants()
            off = 0
            for i, x in enumerate(xs):
                heads.append(g.head_decode(x, self.m[i], z, rm, off, self.na, self.no, strides[i], anchors[i],
                                           self.grid_transposed))
                off += self.na * x.H * x.W
            g.output("z", z)
            g.output("row_max", rm)
        else:
            for i, x in enumerate(xs):
                heads.append(g.conv(x, self.m[i], out_dtype=_ffi.F32, external=True))
        for i, h in enumerate(heads):
            g.output(f"head{i}", h)
        return heads

    def _host_constants(self):
        """Host copies of stride / anchor_grid, refreshed when the buffer changes (reading them from the device on
        every call would stall the stream)."""
        ver = (self.anchor_grid._version, self.anchor_grid.data_ptr(), tuple(float(s) for s in self.stride))
        cached = self.__dict__.get("_ydet_anchor_host")
        if cached is None or cached[0] != ver:
            cached = (ver, self.anchor_grid.detach().float().cpu().view(self.nl, -1).tolist())
            self.__dict__["_ydet_anchor_host"] = cached
        return list(ver[2]), cached[1]

    def finish(self, outs):
        """Plan outputs -> (z, [x_i]) as the reference returns them: the decoded tensor comes from the fused head
        kernels when the plan has it, from the stand-alone decode kernel otherwise."""
        heads = [outs[f"head{i}"] for i in range(self.nl)]
        if "z" not in outs:
            return self.decode(heads)
        z = outs["z"].view(heads[0].shape[0], -1, self.no)
        z._ydet_row_max = (outs["row_max"].view(z.shape[0], -1), z._version)
        xs = [h.view(h.shape[0], h.shape[1], h.shape[2], self.na, self.no).permute(0, 3, 1, 2, 4) for h in heads]
        return z, xs

    def decode(self, heads):
        """heads: NHWC fp32 maps [N, ny, nx, na*no] -> (z, [x_i]) as the reference returns them."""
        strides, anchors = self._host_constants()
        z = ops.decode_v5(heads, strides, anchors, self.na, self.no, self.grid_transposed)
        xs = [h.view(h.shape[0], h.shape[1], h.shape[2], self.na, self.no).permute(0, 3, 1, 2, 4) for h in heads]
        return z, xs

    def _forward_hip(self, x):
        """Stand-alone call with CUDA NCHW feature maps."""
        key = tuple(tuple(t.shape) for t in x) + (engine.compute_dtype_for(self),)
        cache = self.__dict__.setdefault("_ydet_plan_cache", engine.PlanCache(self))

        def build():
            g = engine.Graph(x[0].device, key[-1])
            self.lower(g, [g.input_nchw(*t.shape) for t in x])
            return g.finalize()

        outs = cache.get(key, build).run(*x)
        z, xs = self.finish(outs)
        for i in range(self.nl):
            x[i] = xs[i]
        return z, x

    def _apply(self, fn, *a, **k):
        self.__dict__.pop("_ydet_plan_cache", None)
        return super()._apply(fn, *a, **k)


_MODULES = {
    "Conv": Conv, "Bottleneck": Bottleneck, "SPP": SPP, "DWConv": DWConv, "Focus": Focus,
    "BottleneckCSP": BottleneckCSP, "C3": C3, "C3SPP": C3SPP, "Concat": Concat, "Contract": Contract, "Expand": Expand,
    "Detect": Detect, "nn.Upsample": nn.Upsample, "nn.BatchNorm2d": nn.BatchNorm2d,
}
_CHANNEL_SCALED = (Conv, Bottleneck, SPP, Focus, BottleneckCSP, C3)   # (the reference's list at :505 leaves C3SPP out too)
_REPEAT_INSIDE = (BottleneckCSP, C3)


def make_divisible(x, divisor):
    return math.ceil(x / divisor) * divisor


def _resolve_arg(a, names):
    """Config strings: symbolic names (nc, anchors), Python literals, or plain strings ('nearest')."""
    if not isinstance(a, str):
        return a
    if a in names:
        return names[a]
    try:
        return {"None": None, "True": True, "False": False}[a]
    except KeyError:
        pass
    try:
        return float(a) if "." in a else int(a)
    except ValueError:
        return a


def parse_model(par, inch):
    """Build the nn.Sequential described by `par` (reference yolov5_official/model.py:486-544).
    Returns (model, sorted indices of layers whose outputs later layers consume)."""
    anchors, nc, gd, gw = par["anchors"], par["nc"], par["depth_multiple"], par["width_multiple"]
    na = len(anchors[0]) // 2 if isinstance(anchors, list) else anchors
    n_out = na * (nc + 5)
    names = {"nc": nc, "anchors": anchors}

    layers, save, c2 = [], [], inch[-1]
    for i, (f, n, m, args) in enumerate(par["backbone"] + par["head"]):
        mod = m if not isinstance(m, str) else _MODULES.get(m)
        if mod is None:
            raise KeyError(f"layer {i}: module '{m}' is not available in this package")
        args = [_resolve_arg(a, names) for a in args]
        n = max(round(n * gd), 1) if n > 1 else n
        if mod in _CHANNEL_SCALED or mod is DWConv:
            c1, c2 = inch[f], args[0]
            if c2 != n_out:
                c2 = make_divisible(c2 * gw, 8)
            args = [c1, c2, *args[1:]]
            if mod in _REPEAT_INSIDE:
                args.insert(2, n)
                n = 1
        elif mod is nn.BatchNorm2d:
            args = [inch[f]]
        elif mod is Concat:
            c2 = sum(inch[j] for j in f)
        elif mod is Detect:
            args.append([inch[j] for j in f])
            if isinstance(args[1], int):
                args[1] = [list(range(args[1] * 2))] * len(f)
        elif mod is Contract:
            c2 = inch[f] * args[0] ** 2
        elif mod is Expand:
            c2 = inch[f] // args[0] ** 2
        else:
            c2 = inch[f]

        layer = nn.Sequential(*[mod(*args) for _ in range(n)]) if n > 1 else mod(*args)
        layer.i, layer.f = i, f
        layer.type = str(mod)[8:-2].replace("__main__.", "")
        layer.np = sum(p.numel() for p in layer.parameters())
        save.extend(j % i for j in ([f] if isinstance(f, int) else f) if j != -1)
        layers.append(layer)
        if i == 0:
            inch = []
        inch.append(c2)
    return nn.Sequential(*layers), sorted(save)


def check_anchor_order(m):
    """Flip the anchor sets if their area order disagrees with the stride order."""
    area = m.anchor_grid.prod(-1).view(-1)
    if (area[-1] - area[0]).sign() != (m.stride[-1] - m.stride[0]).sign():
        print("Reversing anchor order")
        m.anchors[:] = m.anchors.flip(0)
        m.anchor_grid[:] = m.anchor_grid.flip(0)


def initialize_weights(model):
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
        elif isinstance(m, (nn.LeakyReLU, nn.ReLU, nn.ReLU6)):
            m.inplace = True


def scale_img(img, ratio=1.0, same_shape=False, gs=32):
    if ratio == 1.0:
        return img
    h, w = img.shape[2:]
    s = (int(h * ratio), int(w * ratio))
    img = F.interpolate(img, size=s, mode="bilinear", align_corners=False)
    if not same_shape:
        h, w = [math.ceil(v * ratio / gs) * gs for v in (h, w)]
    return F.pad(img, [0, w - s[1], 0, h - s[0]], value=0.447)


def lower_layer(g, m, x):
    """Lower one parsed layer (a block, an nn.Sequential of blocks, or an nn.Upsample)."""
    return engine.lower_any(g, m, x)


class Model(engine.AcceleratedModule):
    """Reference yolov5_official/model.py:88-265."""

    def __init__(self, cfg=DEFAULT_CFG, ch=3, nc=None, anchors=None):
        super().__init__()
        if isinstance(cfg, dict):
            self.yaml = cfg
        else:
            self.yaml_file = Path(cfg).name
            with open(cfg) as fh:
                self.yaml = yaml.safe_load(fh)
        ch = self.yaml["ch"] = self.yaml.get("ch", ch)
        if nc and nc != self.yaml["nc"]:
            print(f"Overriding model.yaml nc={self.yaml['nc']} with nc={nc}")
            self.yaml["nc"] = nc
        if anchors:
            print("Overriding model.yaml anchors")
            self.yaml["anchors"] = round(anchors)

        self.model, self.save = parse_model(deepcopy(self.yaml), inch=[ch])
        self.names = [str(i) for i in range(self.yaml["nc"])]
        self.inplace = self.yaml.get("inplace", True)

        m = self.model[-1]
        if isinstance(m, Detect):
            s = 256
            m.inplace = self.inplace
            m.stride = torch.tensor([s / t.shape[-2] for t in self.forward(torch.zeros(1, ch, s, s))])
            m.anchors /= m.stride.view(-1, 1, 1)
            check_anchor_order(m)
            self.stride = m.stride
            self._initialize_biases()
        initialize_weights(self)
        self.trainable = True

    def _initialize_biases(self, cf=None):
        m = self.model[-1]
        for conv, s in zip(m.m, m.stride):
            b = conv.bias.view(m.na, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:] += math.log(0.6 / (m.nc - 0.99)) if cf is None else torch.log(cf / cf.sum())
            conv.bias = nn.Parameter(b.view(-1), requires_grad=True)

    # ---- forward -----------------------------------------------------------------------------
    def forward(self, x, augment=False):
        if augment:
            if x.dtype == torch.uint8:     # the scaled / flipped copies are resampled in floating point
                x = x.float() / 255.0
            return self.forward_augment(x)
        if x.is_cuda and not self.training:
            return self._forward_hip(x)
        if x.dtype == torch.uint8:     # raw pixels (accepted on the accelerated path, scaled there while loading)
            x = x.float() / 255.0
        return self.forward_once(x)

    def forward_once(self, x):
        y = []
        for m in self.model:
            if m.f != -1:
                x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
            x = m(x)
            y.append(x if m.i in self.save else None)
        return x

    def forward_augment(self, x):
        img_size = x.shape[-2:]
        y = []
        for si, fi in zip([1, 0.83, 0.67], [None, 3, None]):
            xi = scale_img(x.flip(fi) if fi else x, si, gs=int(self.stride.max()))
            yi = self.forward(xi)[0]
            y.append(self._descale_pred(yi, fi, si, img_size))
        return torch.cat(y, 1), None

    def _descale_pred(self, p, flips, scale, img_size):
        x, y, wh = p[..., 0:1] / scale, p[..., 1:2] / scale, p[..., 2:4] / scale
        if flips == 2:
            y = img_size[0] - y
        elif flips == 3:
            x = img_size[1] - x
        return torch.cat((x, y, wh, p[..., 4:]), -1)

    # ---- HIP path ----------------------------------------------------------------------------------
    def _lower(self, g, N, C, H, W):
        layers = list(self.model)
        first = layers[0]
        if isinstance(first, Focus) and first.f == -1:
            act, slope = activation_code(first.conv.act)
            x = g.focus_stem(N, C, H, W, first.conv.conv, getattr(first.conv, "bn", None), act=act, slope=slope)
        else:
            x = lower_layer(g, first, g.input_nchw(N, C, H, W))
        y = [x if first.i in self.save else None]
        for m in layers[1:]:
            if m.f != -1:
                x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
            x = lower_layer(g, m, x)
            y.append(x if m.i in self.save else None)
        return x

    def _forward_hip(self, x):
        det = self.model[-1]
        if not isinstance(det, Detect):
            raise NotImplementedError("the accelerated path expects a Detect head as the last layer")
        compute = engine.compute_dtype_for(self)
        key = (tuple(x.shape), compute, x.device.index)

        def build():
            return engine.build_plan(x.device, compute, x.shape[0], lambda g, n: self._lower(g, n, *x.shape[1:]))

        outs = self._plans().get(key, build).run(x)
        z, xs = det.finish(outs)
        p = next(self.parameters())
        if p.dtype != torch.float32:  # model.half(): hand back tensors in the model's dtype
            z, xs = z.to(p.dtype), [t.to(p.dtype) for t in xs]
        return z, xs

    # ---- utilities of the reference class --------------------------------------------------------
    def fuse(self):
        """Fold every Conv's BatchNorm into its convolution (reference :258-265)."""
        from .utils import fuse_conv_and_bn
        for m in self.model.modules():
            if type(m) is Conv and hasattr(m, "bn"):
                m.conv = fuse_conv_and_bn(m.conv, m.bn)
                delattr(m, "bn")
                m.forward = m.fuseforward
        self.invalidate_plans()
        return self

    def autoshape(self):
        from .utils import AutoShape
        m = AutoShape(self)
        for k in ("yaml", "nc", "hyp", "names", "stride"):
            if hasattr(self, k):
                setattr(m, k, getattr(self, k))
        return m
