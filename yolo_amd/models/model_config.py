"""Per-detector configuration tables: the anchor sets are the reference's values
(models/model_config.py) -- they are part of each model's definition; the rest are its training
schedule constants, kept so code that reads these dicts keeps working."""

_V2_VOC = [[1.19, 1.98], [2.79, 4.59], [4.53, 8.92], [8.06, 5.29], [10.32, 10.65]]
_V2_COCO = [[0.53, 0.79], [1.71, 2.36], [2.89, 6.44], [6.33, 3.79], [9.03, 9.74]]
_V3_VOC = [[32.64, 47.68], [50.24, 108.16], [126.72, 96.32], [78.4, 201.92], [178.24, 178.56], [129.6, 294.72],
           [331.84, 194.56], [227.84, 325.76], [365.44, 358.72]]
_V3_COCO = [[12.48, 19.2], [31.36, 46.4], [46.4, 113.92], [97.28, 55.04], [133.12, 127.36], [79.04, 224.],
            [301.12, 150.4], [172.16, 285.76], [348.16, 341.12]]
_TINY_VOC = [[34.01, 61.79], [86.94, 109.68], [93.49, 227.46], [246.38, 163.33], [178.68, 306.55], [344.89, 337.14]]
_TINY_COCO = [[15.09, 23.25], [46.36, 61.47], [68.41, 161.84], [168.88, 93.59], [154.96, 257.45], [334.74, 302.47]]


def _cfg(backbone, voc, coco=None, lr_epoch=(100, 150), max_epoch=200):
    d = {"backbone": backbone, "train_size": 320, "val_size": 416, "random_size_range": [10, 19],
         "anchor_size_voc": voc, "lr_epoch": lr_epoch, "max_epoch": max_epoch, "ignore_thresh": 0.5}
    if coco is not None:
        d["anchor_size_coco"] = coco
    return d


yolov2_d19_cfg = _cfg("d19", _V2_VOC, _V2_COCO)
yolov2_r50_cfg = _cfg("r50", _V2_VOC, _V2_COCO)
yolov2_slim_cfg = _cfg("dtiny", _V2_VOC, _V2_COCO, lr_epoch=(90, 120), max_epoch=150)
yolov3_d53_cfg = _cfg("d53", _V3_VOC, _V3_COCO)
yolov4_cfg = _cfg("cspd53", _V3_VOC, _V3_COCO)
yolov3_tiny_cfg = _cfg("d-light", _TINY_VOC, _TINY_COCO)
yolov5_cfg = _cfg(None, _V3_VOC)
del yolov5_cfg["backbone"]
