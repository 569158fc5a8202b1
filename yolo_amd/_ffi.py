"""ctypes binding of libydet.so (declared in include/ydet.h).

The library is the only compute path of this package on a GPU: if it cannot be loaded the first
call raises, there is no fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# YDET_LIB=<path>: another build of the library (A/B timing of kernel variants; the ABI check below still applies)
LIB_PATH = os.environ.get("YDET_LIB") or os.path.join(_HERE, "libydet.so")

F32, BF16, U8 = 0, 1, 2   # U8: output dtype of ydet_preprocess_u8 only
ACT_NONE, ACT_LEAKY, ACT_SILU = 0, 1, 2
RES_NONE, RES_AFTER_ACT, RES_BEFORE_ACT = 0, 1, 2
(OP_CONV, OP_MAXPOOL2, OP_MAXPOOL2S1, OP_SPP, OP_UPSAMPLE, OP_SPACE2DEPTH, OP_COPY, OP_NCHW2NHWC,
 OP_NHWC2NCHW, OP_FOCUS_CONV, OP_HEAD_DECODE, OP_BOTTLENECK, OP_C3_TAIL, OP_RGB_CONV, OP_FOCUS_PAIR, OP_RGB_PAIR) = range(16)
MAX_ANCHORS = 8
ABI_VERSION = 16   # YDET_ABI_VERSION of include/ydet.h these structs were written against


class ConvDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p), ("y", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("Cout", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("x_ld", C.c_int32), ("x_coff", C.c_int32),
        ("y_ld", C.c_int32), ("y_coff", C.c_int32),
        ("res_ld", C.c_int32), ("res_coff", C.c_int32),
        ("kp", C.c_int32), ("cout_pad", C.c_int32),
        ("act", C.c_int32), ("slope", C.c_float), ("res_mode", C.c_int32),
        ("in_dtype", C.c_int32), ("out_dtype", C.c_int32), ("cout_store", C.c_int32),
        ("post_scale", C.c_void_p), ("post_shift", C.c_void_p), ("post_act", C.c_int32), ("post_slope", C.c_float),
        ("y2", C.c_void_p), ("y2_ld", C.c_int32), ("y2_coff", C.c_int32), ("y_split", C.c_int32),
        ("x_up", C.c_void_p), ("x_up_ld", C.c_int32), ("x_up_coff", C.c_int32), ("x_up_C", C.c_int32),
        ("pw_w", C.c_void_p), ("pw_bias", C.c_void_p), ("pw_cout", C.c_int32), ("pw_kp", C.c_int32),
        ("groups", C.c_int32), ("dilation", C.c_int32), ("cin_group", C.c_int32),
        ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_int64),
    ]


class SliceDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
        ("x_ld", C.c_int32), ("x_coff", C.c_int32),
        ("y_ld", C.c_int32), ("y_coff", C.c_int32),
        ("dtype", C.c_int32),
    ]


class V5Level(C.Structure):
    _fields_ = [
        ("head", C.c_void_p), ("ny", C.c_int32), ("nx", C.c_int32), ("ld", C.c_int32),
        ("stride", C.c_float), ("anchor_wh", C.c_float * (2 * MAX_ANCHORS)),
    ]


class FALevel(C.Structure):
    _fields_ = [
        ("head", C.c_void_p), ("H", C.c_int32), ("W", C.c_int32), ("ld", C.c_int32),
        ("anchor_wh", C.c_float * (2 * MAX_ANCHORS)),
    ]


class HeadDesc(C.Structure):
    _fields_ = [
        ("z", C.c_void_p), ("row_max", C.c_void_p), ("rows_per_image", C.c_int64), ("level_row_offset", C.c_int32),
        ("na", C.c_int32), ("no", C.c_int32), ("ny", C.c_int32), ("nx", C.c_int32), ("grid_transposed", C.c_int32),
        ("stride", C.c_float), ("anchor_wh", C.c_float * (2 * MAX_ANCHORS)),
    ]


class U8Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("h", C.c_int32), ("w", C.c_int32), ("pitch", C.c_int32), ("rh", C.c_int32),
                ("rw", C.c_int32), ("top", C.c_int32), ("left", C.c_int32), ("reserved", C.c_int32)]


class PreprocDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("out_h", C.c_int32), ("out_w", C.c_int32), ("out_dtype", C.c_int32),
                ("layout", C.c_int32), ("swap_rb", C.c_int32), ("scale_mode", C.c_int32),
                ("pad_value", C.c_int32 * 3), ("mean", C.c_float * 3), ("std", C.c_float * 3)]


class Op(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("iarg", C.c_int32), ("iarg2", C.c_int32), ("iarg3", C.c_int32),
        ("conv", ConvDesc), ("slice", SliceDesc), ("head", HeadDesc), ("conv2", ConvDesc), ("conv3", ConvDesc),
    ]


# name -> (restype, argtypes); every symbol include/ydet.h declares
SYMBOLS = {
    "ydet_last_error": (C.c_char_p, []),
    "ydet_abi_version": (C.c_int, []),
    "ydet_conv2d_nhwc": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "ydet_conv_splitk_workspace_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "ydet_conv_kernel_name": (C.c_int, [C.POINTER(ConvDesc), C.c_char_p, C.c_int]),
    "ydet_focus_conv3x3_nchw": (C.c_int, [C.POINTER(ConvDesc), C.c_int, C.c_void_p]),
    "ydet_focus_conv_pair_nchw": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), C.c_int, C.c_void_p]),
    "ydet_rgb_conv3x3_nchw": (C.c_int, [C.POINTER(ConvDesc), C.c_int, C.c_void_p]),
    "ydet_rgb_conv_pair_nchw": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), C.c_int, C.c_void_p]),
    "ydet_maxpool2x2_nhwc": (C.c_int, [C.POINTER(SliceDesc), C.c_void_p]),
    "ydet_maxpool2x2s1_zeropad_nhwc": (C.c_int, [C.POINTER(SliceDesc), C.c_void_p]),
    "ydet_spp_nhwc": (C.c_int, [C.POINTER(SliceDesc), C.c_void_p]),
    "ydet_upsample2x_nhwc": (C.c_int, [C.POINTER(SliceDesc), C.c_int, C.c_void_p]),
    "ydet_space_to_depth_nhwc": (C.c_int, [C.POINTER(SliceDesc), C.c_int, C.c_void_p]),
    "ydet_copy_slice_nhwc": (C.c_int, [C.POINTER(SliceDesc), C.c_void_p]),
    "ydet_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 10 + [C.c_void_p]),
    "ydet_nhwc_to_nchw": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p]),
    "ydet_head_conv_decode": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(HeadDesc), C.c_void_p]),
    "ydet_process_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p]),
    "ydet_bottleneck_fused": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), C.c_void_p]),
    "ydet_c3_tail_fused": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), C.POINTER(ConvDesc), C.c_void_p]),
    "ydet_decode_v5": (C.c_int, [C.POINTER(V5Level), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "ydet_decode_v2v3": (C.c_int, [C.POINTER(FALevel), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                   C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ydet_nms_v5_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ydet_nms_v5": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_size_t, C.c_void_p]),
    "ydet_nms_v5_static": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "ydet_nms2_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ydet_nms2_out_cap": (C.c_int, [C.c_int, C.c_int]),
    "ydet_nms2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p,
                            C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ydet_nms_cluster_diou_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ydet_nms_cluster_diou": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ydet_nms_perclass_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ydet_nms_perclass_workspace_bytes_min": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ydet_nms_perclass": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ydet_box_iou": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ydet_voc_match_workspace_bytes": (C.c_size_t, [C.c_int]),
    "ydet_voc_match": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_double, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ydet_preprocess_u8": (C.c_int, [C.c_void_p, C.POINTER(PreprocDesc), C.c_void_p, C.c_void_p]),
    "ydet_run_ops": (C.c_int, [C.POINTER(Op), C.c_int, C.c_void_p]),
    "ydet_run_op_lists": (C.c_int, [C.POINTER(C.POINTER(Op)), C.POINTER(C.c_int), C.c_int, C.c_void_p]),
    "ydet_timer_create": (C.c_void_p, [C.c_int]),
    "ydet_timer_destroy": (None, [C.c_void_p]),
    "ydet_run_ops_timed": (C.c_int, [C.POINTER(Op), C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "ydet_timer_elapsed": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
}

_lib = None


def lib():
    """The loaded library; raises RuntimeError if libydet.so is missing or incomplete."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C yolo_amd/csrc` (or __graft_entry__.build()); "
                "the HIP library is the only GPU path of yolo_amd")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.ydet_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} has ABI version {handle.ydet_abi_version()}, this binding needs {ABI_VERSION}: "
                               "rebuild with `make -C yolo_amd/csrc`")
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().ydet_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libydet {what} failed ({rc}): {msg}")


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
