"""YOLOv3-SPP (reference models/v3spp.py): YOLOv3 with an SPP block in front of the stride-32 conv set.
Its eval branch calls decode_boxes with the default v2=True (v3spp.py:180), so the pixel anchors are
multiplied by 32 as well -- reproduced as is."""
from ._fpn3 import ThreeScaleYOLOv3


class YOLOv3(ThreeScaleYOLOv3):
    with_spp = True
    decode_v2 = True
