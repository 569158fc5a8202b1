"""YOLOv2 on Darknet-19 (reference models/v2d19.py): constructor signature, attribute names and
state_dict keys of the reference; inference through the shared HIP tail of AnchorDetector."""
import torch
import torch.nn as nn

from ._anchor_detector import AnchorDetector
from .basemodel.d19 import Dark19
from ..utils.moduleUtils import Conv, reorg_layer
from ..utils.utils import create_grid


class YOLOv2(AnchorDetector):
    decode_v2 = True

    def __init__(self, input_size=None, num_classes=20, trainable=False, conf_thresh=0.001, nms_thresh=0.5,
                 anchor_size=None, hr=False, pretrain=False):
        super().__init__()
        if pretrain:
            raise NotImplementedError("pretrained Darknet-19 weights are not distributed with this package")
        self.input_size, self.num_classes, self.trainable = input_size, num_classes, trainable
        self.conf_thresh, self.nms_thresh = conf_thresh, nms_thresh
        # anchors are in grid units; the reference halves them unless the input is 640 (v2d19.py:24)
        a = torch.tensor(anchor_size, dtype=torch.float32)
        self.anchor_size = a if input_size == 640 else a / 2
        self.num_anchors = len(anchor_size)
        self.stride = 32
        self.grid_cell, self.all_anchor_wh = create_grid(input_size, self.stride, self.anchor_size)

        self.base = Dark19()
        self.convsets1 = nn.Sequential(Conv(1024, 1024, k=3, p=1), Conv(1024, 1024, k=3, p=1))
        self.route_layer = Conv(512, 64, k=1)
        self.reorg = reorg_layer(stride=2)
        self.convsets2 = Conv(1280, 1024, k=3, p=1)
        self.pred = nn.Conv2d(1024, self.num_anchors * (1 + 4 + self.num_classes), 1)

    def set_grid(self, size):
        self.input_size = size
        self.grid_cell, self.all_anchor_wh = create_grid(size, self.stride, self.anchor_size)

    def _heads(self, F, x):
        _, c5, c6 = F.run(self.base, x)
        p6 = F.run(self.convsets1, c6)
        p5 = F.run(self.reorg, F.run(self.route_layer, c5))
        return [F.run(self.pred, F.run(self.convsets2, F.cat([p5, p6])))]

    def _level_anchors(self):
        return [self.anchor_size.flatten().tolist()]
