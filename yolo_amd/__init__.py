"""yolo_amd: MI355X-native detection inference (conv stack -> box decode -> NMS) behind the module
API of the he13689/yolo reference.  Sub-packages mirror the reference's module paths:

    yolo_amd.yolov5_official.{modules,model,utils,val}   YOLOv5-style family
    yolo_amd.models.{common,yolo,v2d19,v2r50,v2slim,v3tiny,v3spp,v3,v4,v5,v5v2,basemodel.{d19,resnet}}
    yolo_amd.utils.{moduleUtils,utils}                    v2/v3/v4 blocks and post-processing
    yolo_amd.data.vocapi_evaluator                        VOC07 mAP

On a ROCm device in eval mode every forward pass runs hand-written gfx950 kernels from libydet.so
(include/ydet.h); there is no other GPU path.
"""
from .engine import set_compute_dtype  # noqa: F401

__all__ = ["set_compute_dtype", "install_reference_aliases"]


def install_reference_aliases():
    """Register this package's mirrors under the reference's top-level module names
    (`yolov5_official`, `models`, `utils`, `data`), so code and pickled checkpoints written against
    the reference resolve to the accelerated classes."""
    import importlib
    import sys
    for name in ("yolov5_official", "models", "utils", "data"):
        sys.modules.setdefault(name, importlib.import_module(f"{__name__}.{name}"))
    for sub in ("yolov5_official.modules", "yolov5_official.model", "yolov5_official.utils", "yolov5_official.val",
                "models.common", "models.yolo", "models.basemodel", "models.basemodel.d19", "models.basemodel.resnet", "models.v2d19",
                "models.v2r50", "models.v2slim",
                "models.v3tiny", "models.v3spp", "models.v3", "models.v4", "models.v5", "models.v5v2", "models.model_config",
                "utils.moduleUtils", "utils.utils", "data.vocapi_evaluator"):
        sys.modules.setdefault(sub, importlib.import_module(f"{__name__}.{sub}"))
