"""The author's second YOLOv5 re-creation ("v5s drawn from the structure diagram"): SiLU ConvBlocks, CSP
modules with a post-concat BatchNorm + LeakyReLU, SPP, FPN + PAN; forward returns the three raw prediction
maps -- reference models/v5v2.py:10-259, same class names, attribute names and state_dict keys.
`computeLoss` (training) is not part of this package."""
import torch
import torch.nn as nn

from .. import _ffi, engine
from ..utils.moduleUtils import SPP
from ..utils.utils import create_gridv3
from ..yolov5_official.modules import activation_code
from ._raw_heads import RawHeadDetector


class SiLU(nn.Module):
    @staticmethod
    def forward(x):
        return x * torch.sigmoid(x)


class ConvBlock(nn.Module):
    """Conv2d(bias=False by default) + BatchNorm2d + SiLU in `self.conv`; channel counts scaled by e (:16-30)."""

    def __init__(self, in_ch, out_ch, k=1, s=1, p=None, g=1, bias=False, e=1.):
        super().__init__()
        in_ch, out_ch = round(e * in_ch), round(out_ch * e)
        p = k // 2 if p is None else p
        self.conv = nn.Sequential(nn.Conv2d(in_ch, out_ch, k, s, p, groups=g, bias=bias), nn.BatchNorm2d(out_ch), SiLU())

    def forward(self, x):
        return self.conv(x)

    def lower(self, g, x, res=None, post=None):
        act, slope = activation_code(self.conv[2])
        return g.conv(x, self.conv[0], self.conv[1], act=act, slope=slope, res=res,
                      res_mode=_ffi.RES_AFTER_ACT if res is not None else _ffi.RES_NONE, post=post)


class Focus(nn.Module):
    """Focus slicing + ConvBlock (:33-50); on the GPU path the slicing is fused into the input conversion."""

    def __init__(self, cin, cout, k=3, s=1, p=1, g=1, e=1.0):
        super().__init__()
        self.main = ConvBlock(cin * 4, round(cout * e), k, s, p, g=g)

    def forward(self, x):
        return self.main(torch.cat([x[..., 0::2, 0::2], x[..., 1::2, 0::2], x[..., 0::2, 1::2], x[..., 1::2, 1::2]], 1))

    def lower(self, g, x):
        if not getattr(x, "focused", False):
            raise NotImplementedError("Focus must be the first layer (its slicing is fused into the input conversion)")
        return self.main.lower(g, x)


class CSPResUnit(nn.Module):
    """n applications of ONE (1x1, 3x3) ConvBlock pair -- shared weights -- plus an outer shortcut (:53-65)."""

    def __init__(self, cin, cout, n):
        super().__init__()
        self.shortcut = cin == cout
        res = nn.Sequential(ConvBlock(cin, cin, 1, 1, 0), ConvBlock(cin, cout, 3, 1, 1))
        self.main = nn.Sequential(*[res for _ in range(n)])

    def forward(self, x):
        return x + self.main(x) if self.shortcut else self.main(x)

    def lower(self, g, x, post=None):
        if len(self.main) == 0:
            raise NotImplementedError("CSPResUnit with n = 0")
        y = x
        for i, pair in enumerate(self.main):
            last = i == len(self.main) - 1
            y = pair[1].lower(g, pair[0].lower(g, y), res=x if last and self.shortcut else None,
                              post=post if last else None)
        return y


def _lower_last(g, last, y1_fn, y2_conv, x, h):
    """cat(y1, y2) -> BatchNorm(2h) -> LeakyReLU -> ConvBlock, with the BatchNorm + LeakyReLU split per half:
    y1_fn(post) builds the first half with the second epilogue stage, y2_conv is a plain conv that takes its half
    folded."""
    bn, act_m, block = last
    act, slope = activation_code(act_m)
    y1 = y1_fn((bn, (0, h), act, slope))
    y2 = g.conv(x, y2_conv, bn, (h, 2 * h), act=act, slope=slope)
    return block.lower(g, g.cat([y1, y2]))


class CSPModule1(nn.Module):
    """last(cat(res(cbl(x)), conv(x))) (:68-98)."""

    def __init__(self, cin, cout, k=1, s=1, p=None, g=1, n=1, e=None):
        super().__init__()
        e = [0.5, 0.5] if e is None else e
        cin, cout = round(cin * e[1]), round(cout * e[1])
        hdim = cout // 2
        n = round(n * e[0])
        p = k // 2 if p is None else p
        self.cbl = ConvBlock(cin, hdim, k, s, p, g=1)
        self.res = CSPResUnit(hdim, hdim, n)
        self.conv = nn.Conv2d(cin, hdim, 1, 1, 0, bias=False)
        self.last = nn.Sequential(nn.BatchNorm2d(hdim * 2), nn.LeakyReLU(inplace=True), ConvBlock(hdim * 2, cout, 1, 1, 0))

    def forward(self, x):
        return self.last(torch.cat([self.res(self.cbl(x)), self.conv(x)], 1))

    def lower(self, g, x):
        h = self.conv.out_channels
        return _lower_last(g, self.last, lambda post: self.res.lower(g, self.cbl.lower(g, x), post=post), self.conv, x, h)


class CSPModule2(nn.Module):
    """last(cat(end(main(start(x))), conv(x))) with `main` = n applications of one shared ConvBlock pair (:101-131)."""

    def __init__(self, cin, cout, e=.5, n=1):
        super().__init__()
        cin, cout = round(cin * e), round(cout * e)
        hdim = cin // 2
        self.start = ConvBlock(cin, hdim, 1, 1, 0)
        recur_block = nn.Sequential(ConvBlock(hdim, hdim, 1, 1, 0), ConvBlock(hdim, hdim, 3, 1, 1))
        self.main = nn.Sequential(*[recur_block for _ in range(n)])
        self.end = nn.Conv2d(hdim, hdim, 1, 1, 0, bias=False)
        self.conv = nn.Conv2d(cin, hdim, 1, 1, 0, bias=False)
        self.last = nn.Sequential(nn.BatchNorm2d(hdim * 2), nn.LeakyReLU(inplace=True), ConvBlock(hdim * 2, cout, 1, 1, 0))

    def forward(self, x):
        return self.last(torch.cat([self.end(self.main(self.start(x))), self.conv(x)], 1))

    def lower(self, g, x):
        h = self.conv.out_channels
        bn, act_m, block = self.last
        act, slope = activation_code(act_m)
        y = self.start.lower(g, x)
        for pair in self.main:
            y = pair[1].lower(g, pair[0].lower(g, y))
        y1 = g.conv(y, self.end, bn, (0, h), act=act, slope=slope)
        y2 = g.conv(x, self.conv, bn, (h, 2 * h), act=act, slope=slope)
        return block.lower(g, g.cat([y1, y2]))


class CSPDarknet(nn.Module):
    """Focus stem + two CSPModule1 stages, stride 8 (:134-147)."""

    def __init__(self, e1=0.33, e2=0.5):
        super().__init__()
        r = lambda c: round(c * e2)
        self.main = nn.Sequential(Focus(3, 64, e=e2), ConvBlock(r(64), r(128), k=3, s=2, p=1),
                                  CSPModule1(128, 128, n=3, e=[e1, e2]), ConvBlock(r(128), r(256), k=3, s=2, p=1),
                                  CSPModule1(256, 256, n=9, e=[e1, e2]))

    def forward(self, x):
        return self.main(x)

    def lower(self, g, x):
        return engine.lower_any(g, self.main, x)


class SPPModule(nn.Module):
    """ConvBlock -> SPP(5, 9, 13) -> ConvBlock (:150-164)."""

    def __init__(self, cin, cout, e=1.):
        super().__init__()
        cin, cout = round(e * cin), round(e * cout)
        hdim = cin // 2
        self.cbl1 = ConvBlock(cin, hdim, 1, 1)
        self.spp = SPP()
        self.cbl2 = ConvBlock(hdim * 4, cout, 1, 1)

    def forward(self, x):
        return self.cbl2(self.spp(self.cbl1(x)))

    def lower(self, g, x):
        return self.cbl2.lower(g, g.spp(self.cbl1.lower(g, x)))


class YOLOv5(RawHeadDetector):
    """(:167-259).  `anchors` must hold a multiple of 3 (w, h) pairs; the reference's default list of five
    cannot be reshaped to [3, n, 2] and fails in its constructor as well."""

    def __init__(self, nc=80, e1=0.33, e2=0.5, anchors=None, pretrained=False, input_size=3, stride=None):
        super().__init__()
        if pretrained:
            raise NotImplementedError("no pretrained backbone weights ship with this package")
        if anchors is None:
            anchors = [[1.19, 1.98], [2.79, 4.59], [4.53, 8.92], [8.06, 5.29], [10.32, 10.65]]
        self.num_classes = nc
        self.anchor_size = torch.Tensor(anchors).view(3, len(anchors) // 3, 2)
        self.num_anchors = len(anchors) // 3
        self.input_size = input_size
        self.stride = stride if stride is not None else [8, 16, 32]
        self.trainable = True
        self.grid_cell, self.stride_tensor, self.all_anchors_wh = create_gridv3(
            input_size, strides=self.stride, num_anchors=self.num_anchors, anchor_size=self.anchor_size)

        self.base = CSPDarknet(e1, e2)
        self.first_layer = nn.Sequential(ConvBlock(256, 512, 3, 2, 1, e=e2), CSPModule1(512, 512, n=9, e=[e1, e2]))
        self.second_layer = nn.Sequential(ConvBlock(512, 1024, k=3, s=2, p=1, e=e2), SPPModule(1024, 1024, e=e2),
                                          CSPModule2(1024, 1024, n=1, e=e2), ConvBlock(1024, 512, 1, 1, 0, e=e2))
        self.upsample_medium = nn.Upsample(scale_factor=2)
        self.third_layer = nn.Sequential(CSPModule2(1024, 512, n=1, e=e2), ConvBlock(512, 256, 1, 1, 0, e=e2))
        self.upsample_large = nn.Upsample(scale_factor=2)
        self.up = CSPModule2(512, 256, n=1, e=e2)
        self.up2middle = ConvBlock(256, 256, 3, 2, 1, e=e2)
        self.middle = CSPModule2(512, 512, n=1, e=e2)
        self.middle2down = ConvBlock(512, 512, 3, 2, 1, e=e2)
        self.down = CSPModule2(1024, 1024, n=1, e=e2)
        self.det_large = nn.Conv2d(round(e2 * 256), 3 * (5 + nc), 1, 1, 0)
        self.det_medium = nn.Conv2d(round(512 * e2), 3 * (5 + nc), 1, 1, 0)
        self.det_small = nn.Conv2d(round(1024 * e2), 3 * (5 + nc), 1, 1, 0)

    def _heads(self, F, x):
        y1 = F.run(self.base, x)
        y2 = F.run(self.first_layer, y1)
        y3 = F.run(self.second_layer, y2)
        y4 = F.run(self.third_layer, F.cat([F.run(self.upsample_medium, y3), y2]))
        y5 = F.run(self.up, F.cat([F.run(self.upsample_large, y4), y1]))
        y6 = F.run(self.middle, F.cat([F.run(self.up2middle, y5), y4]))
        y7 = F.run(self.down, F.cat([F.run(self.middle2down, y6), y3]))
        return F.run(self.det_small, y7), F.run(self.det_medium, y6), F.run(self.det_large, y5)

    def computeLoss(self, preds, target=None, eva=False):
        raise NotImplementedError("training losses are outside this package's scope (inference hot path)")
