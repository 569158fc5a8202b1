"""The author's first YOLOv5 re-creation: CSPDarknetV5 + FPN/PAN of CSPModules, returning the three raw
prediction maps (small, medium, large objects) -- reference models/v5.py:11-87, same attribute names and
state_dict keys."""
import torch.nn as nn

from ..utils.moduleUtils import Conv, CSPModule
from ._raw_heads import RawHeadDetector
from .basemodel.d19 import CSPDarknetV5


class YOLOv5(RawHeadDetector):
    def __init__(self, nc=80, e1=0.33, e2=0.5, pretrained=False):
        super().__init__()
        if pretrained:
            raise NotImplementedError("no pretrained backbone weights ship with this package")
        r = lambda c: round(c * e2)
        e = [e1, e2]
        self.base = CSPDarknetV5(e1, e2)
        self.neck_small = nn.Sequential(CSPModule(1024, 1024, n=3, e=e), Conv(r(1024), r(512), k=1, s=1, p=0))
        self.neck_medium = nn.Sequential(CSPModule(1024, 512, n=3, e=e), Conv(r(512), r(256), 1, s=1, p=0))
        self.tie_large = nn.Sequential(CSPModule(512, 256, n=3, e=e))
        self.tie_medium = nn.Sequential(CSPModule(512, 512, n=3, e=e))
        self.tie_small = nn.Sequential(CSPModule(1024, 1024, n=3, e=e))
        self.pan_medium = nn.Sequential(Conv(r(256), r(256), k=3, s=2, p=1))
        self.pan_small = nn.Sequential(Conv(r(512), r(512), k=3, s=2, p=1))
        self.upsample_medium = nn.Upsample(scale_factor=2)
        self.upsample_large = nn.Upsample(scale_factor=2)
        self.det_large = nn.Conv2d(r(256), 3 * (5 + nc), 1, 1, 0)
        self.det_medium = nn.Conv2d(r(512), 3 * (5 + nc), 1, 1, 0)
        self.det_small = nn.Conv2d(r(1024), 3 * (5 + nc), 1, 1, 0)

    def _heads(self, F, x):
        hl, hm, hs = F.run(self.base, x)
        ys = F.run(self.neck_small, hs)
        ym = F.run(self.neck_medium, F.cat([F.run(self.upsample_medium, ys), hm]))
        yl = F.run(self.tie_large, F.cat([F.run(self.upsample_large, ym), hl]))
        mid = F.run(self.tie_medium, F.cat([F.run(self.pan_medium, yl), ym]))
        small = F.run(self.tie_small, F.cat([F.run(self.pan_small, mid), ys]))
        return F.run(self.det_small, small), F.run(self.det_medium, mid), F.run(self.det_large, yl)
