"""Tiny YOLOv3 (reference models/v3tiny.py): two scales (stride 16 and 32), 'v3' box decode."""
import torch
import torch.nn as nn

from ._anchor_detector import AnchorDetector
from .basemodel.d19 import DarkNet_Light
from ..utils.moduleUtils import Conv
from ..utils.utils import create_gridv3


class YOLOv3(AnchorDetector):
    decode_v2 = False  # v3tiny.py:146 passes v2=False

    def __init__(self, input_size=None, num_classes=20, trainable=False, conf_thresh=0.001, nms_thresh=0.50,
                 anchor_size=None, pretrained=False, hr=False):
        super().__init__()
        if pretrained:
            raise NotImplementedError("pretrained backbone weights are not distributed with this package")
        self.input_size, self.num_classes, self.trainable = input_size, num_classes, trainable
        self.conf_thresh, self.nms_thresh = conf_thresh, nms_thresh
        self.stride = [16, 32]
        self.anchor_size = torch.tensor(anchor_size, dtype=torch.float32).view(2, len(anchor_size) // 2, 2)
        self.num_anchors = self.anchor_size.size(1)
        self.set_grid(input_size)

        self.base = DarkNet_Light()
        self.conv_set_2 = Conv(1024, 256, k=3, p=1)
        self.conv_1x1_2 = Conv(256, 128, k=1)
        self.extra_conv_2 = Conv(256, 512, k=3, p=1)
        self.pred_2 = nn.Conv2d(512, self.num_anchors * (1 + 4 + self.num_classes), kernel_size=1)
        self.upsample2 = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.conv_set_1 = Conv(384, 256, k=3, p=1)
        self.pred_1 = nn.Conv2d(256, self.num_anchors * (1 + 4 + self.num_classes), kernel_size=1)
        self.softmax = nn.Softmax(dim=1)

    def set_grid(self, x):
        self.input_size = x
        self.grid_cell, self.stride_tensor, self.all_anchors_wh = create_gridv3(x, self.stride, self.num_anchors,
                                                                                self.anchor_size)

    def _heads(self, F, x):
        c4, c5 = F.run(self.base, x)
        y1 = F.run(self.conv_set_2, c5)
        up = F.run(self.upsample2, F.run(self.conv_1x1_2, y1))
        fine = F.run(self.pred_1, F.run(self.conv_set_1, F.cat([c4, up])))
        coarse = F.run(self.pred_2, F.run(self.extra_conv_2, y1))
        return [fine, coarse]

    def _level_anchors(self):
        return [lvl.flatten().tolist() for lvl in self.anchor_size]
