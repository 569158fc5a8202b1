"""YOLOv3 on Darknet-53, three scales, 'v3' box decode (reference models/v3.py)."""
from ._fpn3 import ThreeScaleYOLOv3


class YOLOv3(ThreeScaleYOLOv3):
    with_spp = False
    decode_v2 = False  # v3.py:249 passes v2=False
