"""ctypes binding of libodtk.so (the C-ABI declared in include/odtk.h).

The product path has NO fallback: if the HIP library is missing or a call fails,
this module raises.  Build with `python __graft_entry__.py build` or
`make -C object-detection-tensorflow_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ODTK_LIB") or os.path.join(_HERE, "libodtk.so")     # ODTK_LIB: A/B builds of the kernels (tools/)

BF16 = 0
F32 = 1
F32X3 = 2          # conv descriptors only: f32 tensors, convolutions as three bf16 MFMA products where that is faster (include/odtk.h)


class ConvDesc(C.Structure):
    """Mirror of `odtk_conv_desc` (include/odtk.h)."""
    _fields_ = [(n, C.c_int) for n in (
        "N", "H", "W", "C", "ldx", "Ho", "Wo", "K", "ldy", "R", "S", "stride", "dil",
        "pad_t", "pad_l", "dtype", "out_dtype")]


class AugPlan(C.Structure):
    """Mirror of `odtk_aug_plan` (include/odtk.h)."""
    _fields_ = ([("src", C.c_void_p)] +
                [(n, C.c_int) for n in ("src_u8", "src_chw", "in_h", "in_w", "resize", "resize_h", "resize_w", "crop_h", "crop_w",
                                        "flip_td", "flip_lr", "has_brightness", "has_contrast", "has_hue", "has_rotate")] +
                [(n, C.c_float) for n in ("brightness", "contrast", "hue", "angle", "ratio_y", "ratio_x")])


class OdtkError(RuntimeError):
    pass


_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong
_cd = C.POINTER(ConvDesc)

# name -> (restype, argtypes); every symbol include/odtk.h declares
SIGNATURES = {
    "odtk_last_error": (C.c_char_p, []),
    "odtk_version": (_i, []),
    "odtk_device_info": (_i, [C.POINTER(_i), C.c_char_p, _i]),
    "odtk_crc32c": (C.c_uint, [_vp, _ll, C.c_uint]),
    "odtk_debug_set": (_i, [_i, _i]),
    "odtk_scratch_slot": (_i, [_i]),
    "odtk_conv_last_kernel": (C.c_char_p, []),
    "odtk_conv2d_fwd": (_i, [_cd, _vp, _vp, _vp, _vp, _i, _vp]),
    "odtk_conv2d_fwd_pool2x2": (_i, [_cd, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "odtk_conv2d_fwd_pool2x2_fused": (_i, [_cd]),
    "odtk_conv2d_dgrad": (_i, [_cd, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "odtk_conv2d_relu_bits_supported": (_i, [_cd, _cd, _i]),
    "odtk_conv2d_fwd_bits": (_i, [_cd, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "odtk_conv2d_dgrad_bits": (_i, [_cd, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "odtk_conv2d_wgrad": (_i, [_cd, _vp, _vp, _i, _vp, _vp, _vp]),
    "odtk_conv2d_x3_supported": (_i, [_cd]),
    "odtk_filter_prepare": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "odtk_filter_prepare_batched": (_i, [_vp, _i, _i, _i, _vp]),
    "odtk_preprocess": (_i, [_vp, _ll, C.POINTER(_f), _i, _i, _vp, _vp]),
    "odtk_preprocess_norm": (_i, [_vp, _ll, _f, C.POINTER(_f), C.POINTER(_f), _i, _i, _vp, _vp]),
    "odtk_refinedet_loss": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _f, _vp, _vp, _vp, _vp,
                                _vp, _vp]),
    "odtk_refinedet_decode": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    "odtk_add_relu_fwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _ll, _i, _i, _vp]),
    "odtk_relu_bwd": (_i, [_vp, _vp, _i, _vp, _i, _ll, _i, _i, _i, _vp]),
    "odtk_avgpool2x2_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "odtk_avgpool2x2_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "odtk_adam": (_i, [_vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _f, _f, _vp, _vp, _i, _vp]),
    "odtk_maxpool_fwd": (_i, [_vp, _vp] + [_i] * 12 + [_vp]),
    "odtk_maxpool_bwd": (_i, [_vp, _vp, _vp, _vp] + [_i] * 12 + [_vp]),
    "odtk_maxpool2x2_fwd_idx": (_i, [_vp, _vp, _vp] + [_i] * 8 + [_vp]),
    "odtk_maxpool_fwd_argmax": (_i, [_vp, _vp, _vp] + [_i] * 12 + [_vp]),
    "odtk_maxpool_bwd_argmax": (_i, [_vp, _vp, _vp] + [_i] * 12 + [_vp]),
    "odtk_maxpool2x2_bwd_idx": (_i, [_vp, _vp, _vp] + [_i] * 8 + [_vp]),
    "odtk_bn_moments": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "odtk_bn_fwd_given": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _ll, _vp, _vp]),
    "odtk_bn_bwd_sums": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _ll, _vp, _vp, _i, _vp, _vp, _vp]),
    "odtk_bn_bwd_given": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _ll, _vp, _vp, _vp, _i, _vp, _ll, _vp, _vp, _vp]),
    "odtk_resize_bilinear_fwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "odtk_resize_bilinear_bwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "odtk_resize_bilinear2_fwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "odtk_resize_bilinear2_bwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "odtk_copy_channels": (_i, [_vp, _i, _i, _vp, _i, _i, _ll, _i, _i, _i, _vp, _vp]),
    "odtk_yolov2_loss": (_i, [_vp, _i, _i, _i, _i, _i, C.POINTER(_f), _f, _vp, _i, _f, _f, _f, _f, _f, _vp, _vp, _vp]),
    "odtk_yolov2_decode_candidates": (_i, [_vp, _i, _i, _i, _i, C.POINTER(_f), _f, _vp, _vp, _vp]),
    "odtk_rows_to_f32": (_i, [_vp, _i, _i, _vp, _i, _i, _ll, _ll, _i, _vp]),
    "odtk_rows_from_f32": (_i, [_vp, _i, _i, _ll, _vp, _i, _i, _ll, _i, _vp]),
    "odtk_gn_workspace_bytes": (_ll, [_i, _i]),
    "odtk_gn_fwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "odtk_gn_bwd": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "odtk_exp_rows_to_f32": (_i, [_vp, _i, _i, _vp, _ll, _i, _vp]),
    "odtk_exp_rows_bwd": (_i, [_vp, _vp, _vp, _i, _i, _ll, _i, _vp]),
    "odtk_add2d": (_i, [_vp, _i, _vp, _i, _vp, _i, _ll, _i, _i, _vp]),
    "odtk_upsample2x_fwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "odtk_upsample2x_bwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "odtk_bn_workspace_bytes": (_ll, [_i, _i]),
    "odtk_bn_fwd": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _ll,
                         _vp, _vp]),
    "odtk_bn_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _ll, _vp, _vp, _vp, _i, _vp, _vp, _vp,
                         _vp, _vp]),
    "odtk_l2norm_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "odtk_l2norm_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "odtk_colsum": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "odtk_sgd_blocks": (_i, [_ll]),
    "odtk_sgd_momentum": (_i, [_vp, _vp, _vp, _ll, _f, _f, _f, _f, _vp, _vp, _i, _vp]),
    "odtk_sum_f32": (_i, [_vp, _ll, _vp, _vp]),
    "odtk_zero": (_i, [_vp, _ll, _vp]),
    "odtk_loss_total": (_i, [_vp, _i, _i, _vp, _ll, _f, _f, _vp, _vp, _vp, _vp]),
    "odtk_cast_from_f32": (_i, [_vp, _vp, _ll, _i, _vp]),
    "odtk_cast_to_f32": (_i, [_vp, _i, _vp, _ll, _vp]),
    "odtk_ssd_priors": (_i, [_i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_f), _vp, _vp, _vp, _vp, _vp,
                             _vp]),
    "odtk_ssd_match": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "odtk_softmax_ce_const": (_i, [_vp, _ll, _i, _i, _i, _vp, _vp]),
    "odtk_nms_batched": (_i, [_vp, _ll, _vp, _ll, _i, _vp, _ll, _i, _i, _i, _i, _vp, _i, _i, _f, _vp, _i,
                              _vp, _vp]),
    "odtk_ssd_loss": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                           _vp, _f, _vp, _vp, _vp]),
    "odtk_retina_anchors": (_i, [_i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_f), _vp, _vp, _vp, _vp, _vp]),
    "odtk_retina_match_workspace_bytes": (_ll, [_i, _i, _i]),
    "odtk_retina_match": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "odtk_retina_loss": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "odtk_retina_decode": (_i, [_vp, _vp, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    "odtk_centernet_workspace_bytes": (_ll, [_i, _i, _i, _i]),
    "odtk_centernet_loss": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "odtk_centernet_decode": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "odtk_fcos_workspace_bytes": (_ll, [C.POINTER(_i), _i]),
    "odtk_fcos_loss": (_i, [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), _vp, _i, _i, _i, _f, _vp,
                            C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _vp, _vp]),
    "odtk_fcos_decode_candidates": (_i, [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), _i, _vp, _vp, _vp]),
    "odtk_yolov3_workspace_bytes": (_ll, [C.POINTER(_i), _i, _i]),
    "odtk_yolov3_loss": (_i, [C.POINTER(_vp), C.POINTER(_i), C.POINTER(_f), C.POINTER(_f), _vp, _i, _i, _i, _i, _f, _f, _f, _f, _f,
                              _vp, C.POINTER(_vp), _vp, _vp]),
    "odtk_yolov3_decode_candidates": (_i, [C.POINTER(_vp), C.POINTER(_i), C.POINTER(_f), C.POINTER(_f), _i, _i, _vp, _vp, _vp]),
    "odtk_augment_workspace_bytes": (_ll, [_i, _i, _i, _i]),
    "odtk_augment_boxes": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "odtk_augment_images": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "odtk_ssd_decode": (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    "odtk_depthwise_conv": (_i, [_vp, _i, _vp, _vp, _i] + [_i] * 9 + [_vp]),
    "odtk_depthwise_wgrad": (_i, [_vp, _i, _vp, _i, _vp] + [_i] * 7 + [_vp]),
    "odtk_lhrcnn_match": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _i] + [_vp] * 13),
    "odtk_lhrcnn_rpn_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i] + [_vp] * 8 + [_f, _i, _i] + [_vp] * 11),
    "odtk_crop_and_resize_fwd": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "odtk_crop_and_resize_bwd": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "odtk_lhrcnn_rcnn_loss": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "odtk_lhrcnn_rpn_decode": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "odtk_lhrcnn_gather_rois": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "odtk_lhrcnn_rcnn_decode": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "odtk_comm_unique_id": (_i, [_vp]),
    "odtk_comm_init": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "odtk_comm_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "odtk_comm_allreduce": (_i, [_vp, _vp, _vp, _ll, _i, _vp]),
    "odtk_comm_broadcast": (_i, [_vp, _vp, _ll, _i, _i, _vp]),
    "odtk_comm_destroy": (_i, [_vp]),
}

_lib = None


def load():
    """Load libodtk.so and attach signatures; raises OdtkError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OdtkError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` in the repo root.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise OdtkError(f"libodtk error {rc}: {load().odtk_last_error().decode()}")


# Recorded launch lists (SSD300 `use_graph='list'`): while a recorder is installed every C-ABI call is also appended to it as (bound ctypes function,
# argument tuple) -- the arguments are the already converted ctypes objects, raw pointers included, so a replay is `for f, a in cmds: f(*a)` at the cost of
# the foreign call alone (the Python wrappers of ops.py, the tensor -> pointer conversions and the stream lookups run once, at record time).  Plain kernel
# launches in stream order: unlike a HIP graph nothing changes on the device side.
_REC = None


def record_begin():
    global _REC
    assert _REC is None, 'a launch list is already being recorded'
    _REC = []
    return _REC


def record_end():
    global _REC
    cmds, _REC = _REC, None
    return cmds


def recording():
    return _REC


def call(name: str, *args):
    f = getattr(load(), name)
    if _REC is not None:
        _REC.append((f, args))
    check(f(*args))
