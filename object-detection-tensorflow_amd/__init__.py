"""odtk -- MI355X-native (gfx950) detector hot path behind the class surface of
Stick-To/Object-Detection-Tensorflow (SSD300 first).

Physical package directory: ``object-detection-tensorflow_amd/``; import name ``odtk``
(the repo-root ``odtk.py`` shim maps one onto the other).  Host code is Python on
PyTorch-ROCm tensors; all arithmetic on the hot path runs in hand-written HIP kernels
reached through the C-ABI in ``include/odtk.h`` (``libodtk.so``).
"""
from . import _lib                      # noqa: F401
from ._lib import BF16, F32, F32X3, OdtkError  # noqa: F401

__all__ = ["BF16", "F32", "OdtkError", "SSD300", "YOLOv3", "RetinaNet", "FCOS", "CenterNet", "SSD512", "RefineDet320", "PFPNetR", "YOLOv2", "LHRCNN"]


def __getattr__(name):
    if name == "SSD300":
        from .ssd300 import SSD300
        return SSD300
    if name == "YOLOv3":
        from .yolov3 import YOLOv3
        return YOLOv3
    if name == "RetinaNet":
        from .retinanet import RetinaNet
        return RetinaNet
    if name == "FCOS":
        from .fcos import FCOS
        return FCOS
    if name == "SSD512":
        from .ssd512 import SSD512
        return SSD512
    if name == "RefineDet320":
        from .refinedet import RefineDet320
        return RefineDet320
    if name == "YOLOv2":
        from .yolov2 import YOLOv2
        return YOLOv2
    if name == "PFPNetR":
        from .pfpnet import PFPNetR
        return PFPNetR
    if name == "CenterNet":
        from .centernet import CenterNet
        return CenterNet
    if name == "LHRCNN":
        from .lhrcnn import LHRCNN
        return LHRCNN
    raise AttributeError(name)
