"""Blocks of the v2/v3/v4 detectors with the names, signatures and state_dict keys of the reference's
utils/moduleUtils.py.  `forward` = stock torch ops (training / CPU tensors); `lower` = emission into
an `engine.Graph` of fused HIP launches (CUDA tensors in inference)."""
import torch
import torch.nn as nn

from .. import _ffi
from ..yolov5_official.modules import activation_code


class Conv(nn.Module):
    """Conv2d(bias=True) + BatchNorm2d [+ LeakyReLU(0.1)], held in `self.conv` (an nn.Sequential, so the
    keys are conv.0.* / conv.1.*) -- reference utils/moduleUtils.py:8-24."""

    def __init__(self, in_ch, out_ch, k=1, p=0, s=1, d=1, g=1, act=True):
        super().__init__()
        stack = [nn.Conv2d(in_ch, out_ch, k, s, p, dilation=d, groups=g), nn.BatchNorm2d(out_ch)]
        if act:
            stack.append(nn.LeakyReLU(0.1, inplace=True))
        self.conv = nn.Sequential(*stack)

    def forward(self, x):
        return self.conv(x)

    def lower(self, g, x, out=None, res=None, res_mode=_ffi.RES_AFTER_ACT):
        act, slope = activation_code(self.conv[2] if len(self.conv) > 2 else None)
        return g.conv(x, self.conv[0], self.conv[1], act=act, slope=slope, res=res, res_mode=res_mode, out=out)


class reorg_layer(nn.Module):
    """Space to depth: out[:, (sy*s+sx)*C + c, i, j] = in[:, c, s*i+sy, s*j+sx] (reference :28-41)."""

    def __init__(self, stride):
        super().__init__()
        self.stride = stride

    def forward(self, x):
        b, c, h, w = x.shape
        s = self.stride
        x = x.view(b, c, h // s, s, w // s, s).permute(0, 3, 5, 1, 2, 4)
        return x.reshape(b, s * s * c, h // s, w // s)

    def lower(self, g, x):
        if self.stride != 2:
            raise NotImplementedError("reorg_layer: only stride 2 is lowered")
        return g.space_to_depth(x, 0)


class resblock(nn.Module):
    """nblocks x [x + Conv3x3(Conv1x1(x))] (reference :45-59)."""

    def __init__(self, ch, nblocks=1):
        super().__init__()
        self.module_list = nn.ModuleList(
            nn.Sequential(Conv(ch, ch // 2, 1), Conv(ch // 2, ch, 3, 1)) for _ in range(nblocks))

    def forward(self, x):
        for pair in self.module_list:
            x = pair(x) + x
        return x

    def lower(self, g, x):
        for pair in self.module_list:
            c1, c2 = pair[0].conv, pair[1].conv
            a1, a2 = activation_code(c1[2] if len(c1) > 2 else None), activation_code(c2[2] if len(c2) > 2 else None)
            if a1 == a2:
                # (one kernel for the 64 -> 32 -> 64 block of the first stage, two convolutions for the wider ones)
                x = g.bottleneck(x, (c1[0], c1[1]), (c2[0], c2[1]), act=a1[0], slope=a1[1], add=True)
            else:
                x = pair[1].lower(g, pair[0].lower(g, x), res=x)
        return x


class SPP(nn.Module):
    """cat(x, maxpool5(x), maxpool9(x), maxpool13(x)) (reference :63-75)."""

    def __init__(self):
        super().__init__()
        self.mp1 = nn.MaxPool2d(5, 1, 2)
        self.mp2 = nn.MaxPool2d(9, 1, 4)
        self.mp3 = nn.MaxPool2d(13, 1, 6)

    def forward(self, x):
        return torch.cat([x, self.mp1(x), self.mp2(x), self.mp3(x)], dim=1)

    def lower(self, g, x):
        return g.spp(x)


class CSPResBlock(nn.Module):
    """act(x + Conv3x3_noact(Conv1x1(x))) (reference :79-91)."""

    def __init__(self, in_ch, out_ch=None):
        super().__init__()
        out_ch = in_ch if out_ch is None else out_ch
        self.conv1 = Conv(in_ch, in_ch, k=1)
        self.conv2 = Conv(in_ch, out_ch, k=3, p=1, act=False)
        self.act = nn.LeakyReLU(0.1, inplace=True)

    def forward(self, x):
        return self.act(x + self.conv2(self.conv1(x)))

    def lower(self, g, x, out=None):
        act, slope = activation_code(self.act)
        c1, c2 = self.conv1.conv, self.conv2.conv
        if activation_code(c1[2]) == (act, slope) and c2[0].out_channels == x.C:
            # (one kernel for the 32- and 64-channel blocks when `out` is not `x`, two convolutions otherwise)
            return g.bottleneck(x, (c1[0], c1[1]), (c2[0], c2[1]), act=act, slope=slope, add=True, out=out,
                                res_mode=_ffi.RES_BEFORE_ACT)
        return g.conv(self.conv1.lower(g, x), c2[0], c2[1], act=act, slope=slope, res=x,
                      res_mode=_ffi.RES_BEFORE_ACT, out=out)


class CSPStage(nn.Module):
    """conv3(cat(conv1(x), resblocks(conv2(x)))) (reference :95-107)."""

    def __init__(self, c, n=1):
        super().__init__()
        h = c // 2
        self.conv1 = Conv(c, h, 1)
        self.conv2 = Conv(c, h, 1)
        self.resblock = nn.Sequential(*[CSPResBlock(in_ch=h) for _ in range(n)])
        self.conv3 = Conv(2 * h, c, 1)

    def forward(self, x):
        return self.conv3(torch.cat([self.conv1(x), self.resblock(self.conv2(x))], dim=1))

    def lower(self, g, x):
        h = self.conv1.conv[0].out_channels
        act, slope = activation_code(self.conv1.conv[2])
        same = all(b.conv2.conv[0].out_channels == h for b in self.resblock)
        if same and h in (32, 64) and g.dt == _ffi.BF16 and x.C % 32 == 0:
            # 32- and 64-channel residual blocks run as one kernel each, which must not work in place (it reads a halo):
            # the sibling convolution sends its halves to buffers of their own, every block writes a new map, and the concat
            # in front of conv3 joins conv1's half with the last block's output
            _, (y1, y2) = g.conv_siblings(x, [(self.conv1.conv[0], self.conv1.conv[1]),
                                              (self.conv2.conv[0], self.conv2.conv[1])], act, slope, separate=True)
            for blk in self.resblock:
                y2 = blk.lower(g, y2)
            return self.conv3.lower(g, g.cat([y1, y2]))
        if h % g.vec == 0 and same:
            # conv1 and conv2 read the same input: one convolution writes [conv1(x) | conv2(x)] into the concat
            # buffer and the residual blocks update the second half in place
            act, slope = activation_code(self.conv1.conv[2])
            cat, (_, y2) = g.conv_siblings(x, [(self.conv1.conv[0], self.conv1.conv[1]),
                                               (self.conv2.conv[0], self.conv2.conv[1])], act, slope)
            for blk in self.resblock:
                blk.lower(g, y2, out=y2)
            return self.conv3.lower(g, cat)
        y1 = self.conv1.lower(g, x)
        y2 = self.conv2.lower(g, x)
        for blk in self.resblock:
            y2 = blk.lower(g, y2)
        return self.conv3.lower(g, g.cat([y1, y2]))


class Bottleneck(nn.Module):
    """Conv1x1 -> Conv3x3 with optional shortcut (reference :110-135)."""

    def __init__(self, cin, cout, shortcut=True, g=1, e=0.5, n=1):
        super().__init__()
        h = int(cout * e)
        self.conv1 = Conv(cin, h, k=1)
        self.conv2 = Conv(h, cout, k=3, p=1, g=g)
        self.shortcut = shortcut and cin == cout

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        return x + y if self.shortcut else y

    def lower(self, g, x):
        return self.conv2.lower(g, self.conv1.lower(g, x), res=x if self.shortcut else None)


class CSPBottleneck(nn.Module):
    """conv4(leaky(bn(cat(conv3(bottlenecks(conv1(x))), conv2(x))))) (reference :138-171).
    conv2/conv3 are plain 1x1 convs; on the GPU path the post-concat BatchNorm + LeakyReLU is folded,
    half by half, into them."""

    def __init__(self, cin, cout, n=1, shortcut=True, g=1, e=.5):
        super().__init__()
        h = int(cout * e)
        self.conv1 = Conv(cin, h, k=1)
        self.conv2 = nn.Conv2d(cin, h, kernel_size=1, bias=False)
        self.conv3 = nn.Conv2d(h, h, kernel_size=1, bias=False)
        self.conv4 = Conv(h * 2, cout, k=1)
        self.bottleblock = nn.Sequential(*[Bottleneck(h, h, shortcut=shortcut, g=g, e=e) for _ in range(n)])
        self.act = nn.LeakyReLU(0.1, inplace=True)
        self.bn = nn.BatchNorm2d(h * 2)

    def forward(self, x):
        y1 = self.conv3(self.bottleblock(self.conv1(x)))
        y2 = self.conv2(x)
        return self.conv4(self.act(self.bn(torch.cat([y1, y2], 1))))

    def lower(self, g, x):
        h = self.conv3.out_channels
        act, slope = activation_code(self.act)
        y1 = self.conv1.lower(g, x)
        for blk in self.bottleblock:
            y1 = blk.lower(g, y1)
        y1 = g.conv(y1, self.conv3, self.bn, (0, h), act=act, slope=slope)
        y2 = g.conv(x, self.conv2, self.bn, (h, 2 * h), act=act, slope=slope)
        return self.conv4.lower(g, g.cat([y1, y2]))


class CSPResUnit(nn.Module):
    """The SAME conv-bn-leaky pair applied n times (one set of weights, hence one state_dict entry) with an
    outer shortcut (reference :174-190)."""

    def __init__(self, cin, cout, n):
        super().__init__()
        self.shortcut = cin == cout
        res = nn.Sequential(
            nn.Conv2d(cin, cin, 1, 1, 0), nn.BatchNorm2d(cin), nn.LeakyReLU(inplace=True),
            nn.Conv2d(cin, cout, 3, 1, 1), nn.BatchNorm2d(cout), nn.LeakyReLU(inplace=True))
        self.main = nn.Sequential(*[res for _ in range(n)])

    def forward(self, x):
        y = self.main(x)
        return x + y if self.shortcut else y

    def lower(self, g, x, post=None):
        if len(self.main) == 0:
            raise NotImplementedError("CSPResUnit with n = 0")
        y = x
        for i, (c1, b1, a1, c2, b2, a2) in enumerate(self.main):
            last = i == len(self.main) - 1
            act, slope = activation_code(a1)
            y = g.conv(y, c1, b1, act=act, slope=slope)
            act, slope = activation_code(a2)
            y = g.conv(y, c2, b2, act=act, slope=slope, res=x if last and self.shortcut else None,
                       res_mode=_ffi.RES_AFTER_ACT, post=post if last else None)
        return y


class CSPModule(nn.Module):
    """tie(cat(up(x), bottom(x))); `res` is constructed (and saved) but not used by forward
    (reference :193-228).  On the GPU path the post-concat BatchNorm + LeakyReLU of `tie` is split per half:
    folded into the plain `bottom` conv, and run as the second epilogue stage of `up` (which already ends in
    its own BatchNorm + LeakyReLU)."""

    def __init__(self, cin, cout, k=1, s=1, p=None, g=1, act=True, n=1, e=None):
        super().__init__()
        e = [0.5, 0.5] if e is None else e
        cin, cout = round(cin * e[1]), round(cout * e[1])
        h = cout // 2
        n = round(n * e[0])
        p = k // 2 if p is None else p
        up = [nn.Conv2d(cin, h, kernel_size=k, stride=s, groups=g, padding=p, bias=False), nn.BatchNorm2d(h)]
        if act:
            up.append(nn.LeakyReLU(inplace=True))
        self.up = nn.Sequential(*up)
        self.res = CSPResUnit(h, h, n)
        self.bottom = nn.Conv2d(cin, h, 1, 1, 0, bias=False)
        self.tie = nn.Sequential(nn.BatchNorm2d(h * 2), nn.LeakyReLU(inplace=True),
                                 nn.Conv2d(h * 2, cout, 1, 1, 0, bias=False))

    def forward(self, x):
        return self.tie(torch.cat([self.up(x), self.bottom(x)], 1))

    def lower(self, g, x):
        h = self.bottom.out_channels
        tie_bn, tie_act, tie_conv = self.tie
        act, slope = activation_code(tie_act)
        up_act, up_slope = activation_code(self.up[2] if len(self.up) > 2 else None)
        y1 = g.conv(x, self.up[0], self.up[1], act=up_act, slope=up_slope, post=(tie_bn, (0, h), act, slope))
        y2 = g.conv(x, self.bottom, tie_bn, (h, 2 * h), act=act, slope=slope)
        return g.conv(g.cat([y1, y2]), tie_conv)


class Focus(nn.Module):
    """Focus slicing + Conv2d(bias=False) + BN + LeakyReLU(0.01) (reference :231-255).  On the GPU path the
    slicing happens while the NCHW image is converted to NHWC, so `lower` takes the already sliced tensor."""

    def __init__(self, cin, cout, k=3, s=1, p=1, g=1, act=True, e=1.0):
        super().__init__()
        cout = round(cout * e)
        self.conv = nn.Conv2d(cin * 4, cout, kernel_size=k, stride=s, groups=g, padding=p, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.act = nn.LeakyReLU(inplace=True) if act else None

    def forward(self, x):
        y = torch.cat([x[..., 0::2, 0::2], x[..., 1::2, 0::2], x[..., 0::2, 1::2], x[..., 1::2, 1::2]], 1)
        y = self.bn(self.conv(y))
        return y if self.act is None else self.act(y)

    def lower(self, g, x):
        if not getattr(x, "focused", False):
            raise NotImplementedError("Focus must be the first layer (its slicing is fused into the input conversion)")
        act, slope = activation_code(self.act)
        return g.conv(x, self.conv, self.bn, act=act, slope=slope)
