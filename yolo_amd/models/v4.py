"""Simplified YOLOv4 (reference models/v4.py): CSPDarknet-53 + SPP + PAN with CSP bottlenecks,
nearest-neighbour upsampling, 'v2' box decode (v4.py:251).  The reference's eval branch stops on an
undefined `self.softmax` (v4.py:252); this class defines it, as the otherwise identical v3spp tail does."""
import torch
import torch.nn as nn

from ._anchor_detector import AnchorDetector
from .basemodel.d19 import CSPDarknet
from ..utils.moduleUtils import Conv, CSPBottleneck, SPP
from ..utils.utils import create_gridv3


class YOLOv4(AnchorDetector):
    decode_v2 = True

    def __init__(self, input_size=None, num_classes=20, trainable=False, conf_thresh=0.001, nms_thresh=0.50,
                 anchor_size=None, pretrained=False, hr=False):
        super().__init__()
        if pretrained:
            raise NotImplementedError("pretrained backbone weights are not distributed with this package")
        self.input_size, self.num_classes, self.trainable = input_size, num_classes, trainable
        self.conf_thresh, self.nms_thresh = conf_thresh, nms_thresh
        self.stride = [8, 16, 32]
        self.anchor_size = torch.tensor(anchor_size, dtype=torch.float32).view(3, len(anchor_size) // 3, 2)
        self.num_anchors = len(anchor_size) // 3
        self.set_grid(input_size)
        nout = self.num_anchors * (1 + self.num_classes + 4)

        self.base = CSPDarknet()
        self.spp = nn.Sequential(Conv(1024, 512, k=1), SPP(), CSPBottleneck(512 * 4, 1024, n=3, shortcut=False))
        self.head_conv_0 = Conv(1024, 512, k=1)
        self.head_upsample_0 = nn.Upsample(scale_factor=2)
        self.head_csp_0 = CSPBottleneck(512 * 2, 512, n=3, shortcut=False)
        self.head_conv_1 = Conv(512, 256, 1)
        self.head_csp_1 = CSPBottleneck(256 * 2, 256, n=3, shortcut=False)
        self.head_upsample_1 = nn.Upsample(scale_factor=2)
        self.head_conv_2 = Conv(256, 256, k=3, p=1, s=2)
        self.head_csp_2 = CSPBottleneck(256 * 2, 512, n=3, shortcut=False)
        self.head_conv_3 = Conv(512, 512, k=3, p=1, s=2)
        self.head_csp_3 = CSPBottleneck(512 * 2, 1024, n=3, shortcut=False)
        self.head_det_1 = nn.Conv2d(256, nout, 1)
        self.head_det_2 = nn.Conv2d(512, nout, 1)
        self.head_det_3 = nn.Conv2d(1024, nout, 1)
        self.softmax = nn.Softmax(dim=1)

    def set_grid(self, x):
        self.input_size = x
        self.grid_cell, self.stride_tensor, self.all_anchors_wh = create_gridv3(x, self.stride, self.num_anchors,
                                                                                self.anchor_size)

    def _heads(self, F, x):
        c3, c4, c5 = F.run(self.base, x)
        c5 = F.run(self.spp, c5)
        y0 = F.run(self.head_conv_0, c5)
        p4 = F.run(self.head_csp_0, F.cat([F.run(self.head_upsample_0, y0), c4]))
        y1 = F.run(self.head_conv_1, p4)
        p3 = F.run(self.head_csp_1, F.cat([F.run(self.head_upsample_1, y1), c3]))
        n4 = F.run(self.head_csp_2, F.cat([F.run(self.head_conv_2, p3), y1]))
        n5 = F.run(self.head_csp_3, F.cat([F.run(self.head_conv_3, n4), y0]))
        return [F.run(self.head_det_1, p3), F.run(self.head_det_2, n4), F.run(self.head_det_3, n5)]

    def _level_anchors(self):
        return [lvl.flatten().tolist() for lvl in self.anchor_size]
