"""Backbones of the v2/v3/v4 detectors with the attribute names (hence state_dict keys) of the
reference's models/basemodel/d19.py.  Each network's wiring is written once, in `_net(F, x)`, against a
two-method adapter: engine.TorchOps runs it eagerly, engine.LowerOps emits it as HIP launches."""
import torch.nn as nn

from ... import engine
from ...utils.moduleUtils import Conv, CSPModule, CSPStage, Focus, SPP, resblock


class _Backbone(nn.Module):
    def forward(self, x, targets=None):
        return self._net(engine.TorchOps, x)

    def lower(self, g, x):
        return self._net(engine.LowerOps(g), x)


def _c3(cin, cout):
    return Conv(cin, cout, k=3, p=1)


class Dark19(_Backbone):
    """Darknet-19 trunk: returns the stride-8 (256 ch), stride-16 (512) and stride-32 (1024) maps
    (reference :17-79)."""

    def __init__(self, num_classes=1000):
        super().__init__()
        pool = lambda: nn.MaxPool2d(2, 2)
        self.conv1 = nn.Sequential(_c3(3, 32), pool())
        self.conv2 = nn.Sequential(_c3(32, 64), pool())
        self.conv3 = nn.Sequential(_c3(64, 128), Conv(128, 64, k=1), _c3(64, 128), pool())
        self.conv4 = nn.Sequential(_c3(128, 256), Conv(256, 128, k=1), _c3(128, 256))
        self.maxpool4 = pool()
        self.conv5 = nn.Sequential(_c3(256, 512), Conv(512, 256, k=1), _c3(256, 512), Conv(512, 256, k=1),
                                   _c3(256, 512))
        self.maxpool5 = pool()
        self.conv6 = nn.Sequential(_c3(512, 1024), Conv(1024, 512, k=1), _c3(512, 1024), Conv(1024, 512, k=1),
                                   _c3(512, 1024))

    def _net(self, F, x):
        x = F.run(self.conv3, F.run(self.conv2, F.run(self.conv1, x)))
        x1 = F.run(self.conv4, x)
        x2 = F.run(self.conv5, F.run(self.maxpool4, x1))
        x3 = F.run(self.conv6, F.run(self.maxpool5, x2))
        return x1, x2, x3


class DarkNet_53(_Backbone):
    """Darknet-53: five stride-2 stages of residual blocks [1,2,8,8,4] (reference :83-123)."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.layer_1 = nn.Sequential(Conv(3, 32, 3, p=1), Conv(32, 64, 3, p=1, s=2), resblock(64, nblocks=1))
        self.layer_2 = nn.Sequential(Conv(64, 128, 3, p=1, s=2), resblock(128, nblocks=2))
        self.layer_3 = nn.Sequential(Conv(128, 256, 3, p=1, s=2), resblock(256, nblocks=8))
        self.layer_4 = nn.Sequential(Conv(256, 512, 3, p=1, s=2), resblock(512, nblocks=8))
        self.layer_5 = nn.Sequential(Conv(512, 1024, 3, p=1, s=2), resblock(1024, nblocks=4))

    def _net(self, F, x):
        c3 = F.run(self.layer_3, F.run(self.layer_2, F.run(self.layer_1, x)))
        c4 = F.run(self.layer_4, c3)
        return c3, c4, F.run(self.layer_5, c4)


class DarkNet_Tiny(_Backbone):
    """All-conv small Darknet, stride-2 convs instead of pools (reference :127-166)."""

    def __init__(self, num_classes=1000):
        super().__init__()
        def stage(cin, cout):
            return nn.Sequential(_c3(cin, cout), Conv(cout, cout, k=3, p=1, s=2))
        self.conv_1 = stage(3, 32)
        self.conv_2 = stage(32, 64)
        self.conv_3 = stage(64, 128)
        self.conv_4 = stage(128, 256)
        self.conv_5 = stage(256, 512)

    def _net(self, F, x):
        c3 = F.run(self.conv_3, F.run(self.conv_2, F.run(self.conv_1, x)))
        c4 = F.run(self.conv_4, c3)
        return c3, c4, F.run(self.conv_5, c4)


class DarkNet_Light(_Backbone):
    """Tiny-YOLOv3 trunk: 7 convs, 2x2 pools; the last pool has stride 1 behind a (0,1,0,1) zero pad
    (reference :170-212).  Returns the stride-16 (256 ch) and stride-32 (1024 ch) maps."""

    def __init__(self, num_classes=1000):
        super().__init__()
        widths = [3, 16, 32, 64, 128, 256, 512, 1024]
        for i in range(1, 8):
            setattr(self, f"conv_{i}", _c3(widths[i - 1], widths[i]))
            if i <= 5:
                setattr(self, f"maxpool_{i}", nn.MaxPool2d((2, 2), 2))
        self.maxpool_6 = nn.Sequential(nn.ZeroPad2d((0, 1, 0, 1)), nn.MaxPool2d((2, 2), 1))

    def _net(self, F, x):
        for i in range(1, 5):
            x = F.run(getattr(self, f"maxpool_{i}"), F.run(getattr(self, f"conv_{i}"), x))
        c4 = F.run(self.conv_5, x)
        x = F.run(self.conv_6, F.run(self.maxpool_5, c4))
        return c4, F.run(self.conv_7, F.run(self.maxpool_6, x))


class CSPDarknet(_Backbone):
    """CSPDarknet-53: CSP stages [1,2,8,8,4] (reference :215-265)."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.layer1 = nn.Sequential(_c3(3, 32), Conv(32, 64, k=3, p=1, s=2), CSPStage(c=64, n=1))
        self.layer2 = nn.Sequential(Conv(64, 128, k=3, p=1, s=2), CSPStage(128, n=2))
        self.layer3 = nn.Sequential(Conv(128, 256, k=3, p=1, s=2), CSPStage(256, n=8))
        self.layer4 = nn.Sequential(Conv(256, 512, k=3, p=1, s=2), CSPStage(512, n=8))
        self.layer5 = nn.Sequential(Conv(512, 1024, k=3, p=1, s=2), CSPStage(1024, n=4))

    def _net(self, F, x):
        c3 = F.run(self.layer3, F.run(self.layer2, F.run(self.layer1, x)))
        c4 = F.run(self.layer4, c3)
        return c3, c4, F.run(self.layer5, c4)


class CSPDarknetV5(_Backbone):
    """Backbone of the author's own v5 re-creation (reference :268-294): Focus stem, CSPModule stages, SPP."""

    def __init__(self, e1, e2):
        super().__init__()
        r = lambda c: round(c * e2)
        self.big = nn.Sequential(Focus(3, 64, e=e2), Conv(r(64), r(128), k=3, s=2, p=1),
                                 CSPModule(128, 128, n=3, e=[e1, e2]), Conv(r(128), r(256), k=3, s=2, p=1),
                                 CSPModule(256, 256, n=9, e=[e1, e2]))
        self.small = nn.Sequential(Conv(r(512), r(1024), k=3, s=2, p=1), Conv(r(1024), r(1024) // 2, k=1, s=1),
                                   SPP(), Conv(r(1024) // 2 * 4, r(1024), k=1, s=1))
        self.medium = nn.Sequential(Conv(r(256), r(512), k=3, s=2, p=1), CSPModule(512, 512, n=9, e=[e1, e2]))

    def _net(self, F, x):
        hb = F.run(self.big, x)
        hm = F.run(self.medium, hb)
        return hb, hm, F.run(self.small, hm)
