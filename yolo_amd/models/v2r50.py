"""YOLOv2 on the ResNet trunk (reference models/v2r50.py; the trunk it builds is ResNet([2, 2, 2, 2])):
constructor signature, attribute names and state_dict keys of the reference; inference through the shared
HIP tail of AnchorDetector."""
import torch
import torch.nn as nn

from ._anchor_detector import AnchorDetector
from .basemodel.resnet import ResNet
from ..utils.moduleUtils import Conv, reorg_layer
from ..utils.utils import create_grid


class YOLOv2(AnchorDetector):
    decode_v2 = True

    def __init__(self, input_size=None, num_classes=20, trainable=False, conf_thresh=0.001, nms_thresh=0.6,
                 anchor_size=None, pretrain=False):
        super().__init__()
        if pretrain:
            raise NotImplementedError("pretrained ResNet weights are fetched from the network by the reference; "
                                      "load a state_dict instead")
        self.input_size, self.num_classes, self.trainable = input_size, num_classes, trainable
        self.conf_thresh, self.nms_thresh = conf_thresh, nms_thresh
        a = torch.tensor(anchor_size, dtype=torch.float32)
        self.anchor_size = a if input_size == 640 else a / 2            # reference :34
        self.num_anchors = len(anchor_size)
        # the trunk never applies its max-pool, so the head sits at stride 16; the reference's table says
        # 32 for a 640 input (:36), where its own grid no longer matches its head map
        self.stride = 32 if input_size == 640 else 16
        self.grid_cell, self.all_anchor_wh = create_grid(input_size, self.stride, self.anchor_size)

        self.base = ResNet([2, 2, 2, 2])
        self.convset1 = nn.Sequential(Conv(512, 256, k=1), Conv(256, 256, k=3, p=1), Conv(256, 256, k=3, p=1))
        self.route_layer = Conv(256, 128, k=1)
        self.reorg = reorg_layer(stride=2)
        self.convset2 = Conv(256 + 128 * 4, 512, k=3, p=1)
        self.pred = nn.Conv2d(512, self.num_anchors * (1 + 4 + self.num_classes), 1)

    def set_grid(self, input_size):
        self.input_size = input_size
        self.grid_cell, self.all_anchor_wh = create_grid(input_size, self.stride, self.anchor_size)

    def _heads(self, F, x):
        _, c5, c6 = F.run(self.base, x)
        p6 = F.run(self.convset1, c6)
        p5 = F.run(self.reorg, F.run(self.route_layer, c5))
        return [F.run(self.pred, F.run(self.convset2, F.cat([p5, p6])))]

    def _level_anchors(self):
        return [self.anchor_size.flatten().tolist()]
