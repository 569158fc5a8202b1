"""Three-scale YOLOv3 wiring shared by models/v3.py and models/v3spp.py of the reference (they differ
only in the first element of conv_set_3 and in the decode mode)."""
import torch
import torch.nn as nn

from ._anchor_detector import AnchorDetector
from .basemodel.d19 import DarkNet_53
from ..utils.moduleUtils import Conv, SPP
from ..utils.utils import create_gridv3


def _five(cin, mid, first=None):
    """conv set: [first or 1x1 cin->mid], 3x3 mid->2mid, 1x1, 3x3, 1x1 -> mid channels."""
    head = list(first) if first is not None else [Conv(cin, mid, k=1)]
    return nn.Sequential(*head, Conv(mid, 2 * mid, k=3, p=1), Conv(2 * mid, mid, k=1), Conv(mid, 2 * mid, k=3, p=1),
                         Conv(2 * mid, mid, k=1))


class ThreeScaleYOLOv3(AnchorDetector):
    with_spp = False

    def __init__(self, input_size=None, num_classes=20, trainable=False, conf_thresh=0.001, nms_thresh=0.50,
                 anchor_size=None, pretrained=False, hr=False):
        super().__init__()
        if pretrained:
            raise NotImplementedError("pretrained backbone weights are not distributed with this package")
        self.input_size, self.num_classes, self.trainable = input_size, num_classes, trainable
        self.conf_thresh, self.nms_thresh = conf_thresh, nms_thresh
        self.stride = [8, 16, 32]
        self.anchor_size = torch.tensor(anchor_size, dtype=torch.float32).view(3, len(anchor_size) // 3, 2)
        self.num_anchors = self.anchor_size.size(1)
        self.set_grid(input_size)
        nout = self.num_anchors * (1 + 4 + self.num_classes)

        self.base = DarkNet_53()
        first = [SPP(), Conv(1024 * 4, 512, k=1)] if self.with_spp else None
        self.conv_set_3 = _five(1024, 512, first)
        self.conv_1x1_3 = Conv(512, 256, k=1)
        self.extra_conv_3 = Conv(512, 1024, k=3, p=1)
        self.pred_3 = nn.Conv2d(1024, nout, kernel_size=1)
        self.conv_set_2 = _five(768, 256)
        self.conv_1x1_2 = Conv(256, 128, k=1)
        self.extra_conv_2 = Conv(256, 512, k=3, p=1)
        self.pred_2 = nn.Conv2d(512, nout, kernel_size=1)
        self.conv_set_1 = _five(384, 128)
        self.extra_conv_1 = Conv(128, 256, k=3, p=1)
        self.pred_1 = nn.Conv2d(256, nout, kernel_size=1)
        self.upsample3 = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.softmax = nn.Softmax(dim=1)
        self.upsample2 = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)

    def set_grid(self, input_size):
        self.input_size = input_size
        self.grid_cell, self.stride_tensor, self.all_anchors_wh = create_gridv3(input_size, self.stride,
                                                                                self.num_anchors, self.anchor_size)

    def _heads(self, F, x):
        c3, c4, c5 = F.run(self.base, x)
        y1 = F.run(self.conv_set_3, c5)
        out1 = F.run(self.pred_3, F.run(self.extra_conv_3, y1))
        y2 = F.run(self.conv_set_2, F.cat([c4, F.run(self.upsample3, F.run(self.conv_1x1_3, y1))]))
        out2 = F.run(self.pred_2, F.run(self.extra_conv_2, y2))
        y3 = F.run(self.conv_set_1, F.cat([c3, F.run(self.upsample2, F.run(self.conv_1x1_2, y2))]))
        out3 = F.run(self.pred_1, F.run(self.extra_conv_1, y3))
        return [out3, out2, out1]

    def _level_anchors(self):
        return [lvl.flatten().tolist() for lvl in self.anchor_size]
