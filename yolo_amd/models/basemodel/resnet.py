"""ResNet trunk of the v2r50 detector with the attribute names (hence state_dict keys) of the reference's
models/basemodel/resnet.py.  `forward` = stock torch ops; `lower` = emission into an `engine.Graph`:
every (conv, BatchNorm[, ReLU]) triple is one fused convolution, the shortcut addition rides in the
epilogue of the block's last convolution (added before the ReLU, as the reference does)."""
import torch.nn as nn

from ... import _ffi


def conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


_RELU = dict(act=_ffi.ACT_LEAKY, slope=0.0)   # max(x, 0) is the leaky form with slope 0


def _shortcut(g, block, x):
    if block.downsample is None:
        return x
    return g.conv(x, block.downsample[0], block.downsample[1])


class BasicBlock(nn.Module):
    """relu(bn2(conv2(relu(bn1(conv1(x))))) + shortcut(x)) (reference :24-54)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        return self.relu(out + identity)

    def lower(self, g, x):
        t = g.conv(x, self.conv1, self.bn1, **_RELU)
        return g.conv(t, self.conv2, self.bn2, res=_shortcut(g, self, x), res_mode=_ffi.RES_BEFORE_ACT, **_RELU)


class Bottleneck(nn.Module):
    """1x1 -> 3x3 (carries the stride) -> 1x1 (x4 channels), shortcut added before the last ReLU
    (reference :58-93)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv1x1(inplanes, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = conv3x3(planes, planes, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = conv1x1(planes, planes * self.expansion)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        return self.relu(self.bn3(self.conv3(out)) + identity)

    def lower(self, g, x):
        t = g.conv(x, self.conv1, self.bn1, **_RELU)
        t = g.conv(t, self.conv2, self.bn2, **_RELU)
        return g.conv(t, self.conv3, self.bn3, res=_shortcut(g, self, x), res_mode=_ffi.RES_BEFORE_ACT, **_RELU)


class ResNet(nn.Module):
    """7x7/2 stem + four BasicBlock stages; returns the outputs of stages 2, 3, 4.  As in the reference
    (:148-155) `self.maxpool` is constructed but not applied, so the stages sit at strides 2, 4, 8, 16."""

    def __init__(self, layers, zero_init_residual=False):
        super().__init__()
        self.inplanes = 64
        stem = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.first_layer = nn.Sequential(stem, nn.BatchNorm2d(64), nn.ReLU(inplace=True))
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for k, (width, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2))):
            setattr(self, f"layer{k + 1}", self.make_layer(BasicBlock, width, layers[k], stride=stride))
        self._reset_parameters(zero_init_residual)

    def _reset_parameters(self, zero_init_residual):
        """He-normal (fan-out) convolutions, unit BatchNorms; optionally a zero gain on each block's last BatchNorm so
        that a fresh block starts as the identity (reference :115-130)."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        if zero_init_residual:
            for m in self.modules():
                last = m.bn3 if isinstance(m, Bottleneck) else m.bn2 if isinstance(m, BasicBlock) else None
                if last is not None:
                    nn.init.zeros_(last.weight)

    def make_layer(self, block, planes, blocks, stride=1):
        """One stage: the first block carries the stride (and a 1x1 + BatchNorm projection when the shape changes),
        the rest keep it (reference :132-146)."""
        width = planes * block.expansion
        project = None
        if stride != 1 or self.inplanes != width:
            project = nn.Sequential(conv1x1(self.inplanes, width, stride), nn.BatchNorm2d(width))
        chain = [block(self.inplanes, planes, stride, project)] + [block(width, planes) for _ in range(blocks - 1)]
        self.inplanes = width
        return nn.Sequential(*chain)

    def _stages(self, run, x):
        y2 = run(self.layer2, run(self.layer1, x))
        y3 = run(self.layer3, y2)
        return y2, y3, run(self.layer4, y3)

    def forward(self, x):
        return self._stages(lambda m, t: m(t), self.first_layer(x))

    def lower(self, g, x):
        from ...engine import lower_any
        y = g.conv(x, self.first_layer[0], self.first_layer[1], **_RELU)
        return self._stages(lambda m, t: lower_any(g, m, t), y)
