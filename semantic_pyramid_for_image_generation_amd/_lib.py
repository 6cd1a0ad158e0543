"""ctypes binding of libsempyr_hip.so (the C ABI declared in include/sempyr.h).

The product path has no CPU fallback: if the shared library is missing or does not export a
declared symbol, importing this module raises.  Only loading / symbol resolution happens here
(no GPU needed); compute entry points are only called by `ops.py` on CUDA(HIP) tensors.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsempyr_hip.so")

c_int, c_float, c_void_p, c_int64, c_size_t = (ctypes.c_int, ctypes.c_float, ctypes.c_void_p,
                                               ctypes.c_int64, ctypes.c_size_t)

MAX_PHASES = 4
CONV_FWD, CONV_DGRAD, CONV_WGRAD = 0, 1, 2
NORM_BATCH, NORM_INSTANCE, NORM_LAYER = 0, 1, 2
MATH_F32, MATH_BF16X3 = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3


class ConvDesc(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("Cin", ctypes.c_int32), ("Hin", ctypes.c_int32), ("Win", ctypes.c_int32),
                ("Cout", ctypes.c_int32), ("kh", ctypes.c_int32), ("kw", ctypes.c_int32),
                ("stride", ctypes.c_int32), ("pad", ctypes.c_int32), ("ups", ctypes.c_int32), ("math", ctypes.c_int32)]


class ConvPlanInfo(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int32), ("Ho", ctypes.c_int32), ("Wo", ctypes.c_int32),
                ("nphases", ctypes.c_int32),
                ("K", ctypes.c_int32 * MAX_PHASES), ("Kpad", ctypes.c_int32 * MAX_PHASES),
                ("tab_off", ctypes.c_int32 * MAX_PHASES), ("tab_entries", ctypes.c_int32),
                ("Mpad", ctypes.c_int32), ("wpack_floats", ctypes.c_int64), ("workspace_bytes", ctypes.c_int64),
                ("splits", ctypes.c_int32 * MAX_PHASES), ("cfg", ctypes.c_int32), ("stats_granule", ctypes.c_int32)]


class PackItem(ctypes.Structure):
    _fields_ = [("desc", ConvDesc), ("mode", ctypes.c_int32), ("w", c_void_p), ("wsrc", c_void_p), ("wpacked", c_void_p)]


class AdamTensor(ctypes.Structure):
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p),
                ("numel", ctypes.c_int64)]


class MaskTarget(ctypes.Structure):
    _fields_ = [("mask", c_void_p), ("rows", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32)]


class AccumItem(ctypes.Structure):
    _fields_ = [("dst", c_void_p), ("src", c_void_p * 4), ("numel", ctypes.c_int64), ("assign", ctypes.c_int32)]


ADAM_MAX_TENSORS = 64
CONCAT_MAX = 8
LINCOMB_MAX = 6
STORE_MAX_WORDS = 48
MASKS_MAX = 6

_P = c_void_p  # every device / host pointer is passed as an integer address
_DESC = ctypes.POINTER(ConvDesc)

# name -> argtypes ; every function returns int except where noted. Mirrors include/sempyr.h 1:1
# (tests/test_abi.py parses the header and checks this table against it).
SIGNATURES = {
    "sempyr_version": [],
    "sempyr_conv2d_plan": [_DESC, c_int, ctypes.POINTER(ConvPlanInfo)],
    "sempyr_conv2d_tables": [_DESC, c_int, _P, _P],
    "sempyr_conv2d_pack_weights": [_DESC, c_int, _P, _P, _P, _P],
    "sempyr_conv2d_pack_weights_many": [ctypes.POINTER(PackItem), c_int, _P],
    "sempyr_conv2d_fwd": [_DESC, _P, _P, _P, _P, _P, _P, c_int, c_int, c_float, _P, _P, _P, _P, _P, c_size_t, _P],
    "sempyr_conv2d_fwd_stats": [_DESC, _P, _P, _P, _P, _P, _P],
    "sempyr_conv2d_dgrad": [_DESC, _P, _P, _P, _P, _P, _P, c_size_t, _P],
    "sempyr_conv2d_wgrad": [_DESC, _P, _P, _P, _P, c_int, _P, c_size_t, _P],
    "sempyr_conv2d_dgrad_unpooled": [_DESC, _P, _P, c_float, _P, _P, _P, _P],
    "sempyr_stem_wgrad_workspace": [_DESC, _P],
    "sempyr_stem_wgrad": [_DESC, _P, _P, c_float, _P, _P, c_int, _P, c_int, _P, c_size_t, _P],
    "sempyr_norm_stats": [c_int, _P, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, c_float, c_int, _P, _P, _P],
    "sempyr_bn_stats_from_tiles": [_P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, c_float, c_int, _P, _P],
    "sempyr_norm_fwd": [c_int, _P, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, c_float, c_int, c_float, _P, _P, _P, _P],
    "sempyr_bn_eval_coeffs": [_P, _P, _P, _P, c_float, c_int, c_int, _P, _P, _P],
    "sempyr_norm_bwd": [c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P],
    "sempyr_norm_bwd_bwd": [c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P],
    "sempyr_affine_act": [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P],
    "sempyr_affine2_act": [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_int, _P],
    "sempyr_affine_act_bwd": [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P],
    "sempyr_mul": [_P, _P, _P, c_int64, _P],
    "sempyr_axpby": [_P, _P, c_float, c_float, _P, c_int64, _P],
    "sempyr_adam_step": [ctypes.POINTER(AdamTensor), c_int, c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                         ctypes.c_double, ctypes.c_double, c_int, _P],
    "sempyr_adam_scalars": [c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.POINTER(ctypes.c_float)],
    "sempyr_adam_step_dev": [ctypes.POINTER(AdamTensor), c_int, _P, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                             ctypes.c_double, c_int, _P],
    "sempyr_store_words": [ctypes.POINTER(ctypes.c_uint32), c_int, _P, _P],
    "sempyr_build_masks": [ctypes.POINTER(MaskTarget), c_int, _P, _P],
    "sempyr_concat": [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64), c_int, _P, _P],
    "sempyr_fill_const": [_P, c_float, c_int64, _P],
    "sempyr_bn_zero_input_add": [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, c_float, _P],
    "sempyr_lincomb": [ctypes.POINTER(c_void_p), ctypes.POINTER(c_float), c_int, c_float, c_float, _P, _P],
    "sempyr_accumulate_many": [ctypes.POINTER(AccumItem), c_int, _P],
    "sempyr_lerp_rows": [_P, _P, _P, _P, c_int, c_int, _P],
    "sempyr_normalize": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "sempyr_resize_crop_normalize": [_P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, ctypes.POINTER(c_float),
                                     ctypes.POINTER(c_float), _P, ctypes.POINTER(c_float), ctypes.POINTER(c_float), _P, _P, _P],
    "sempyr_pool2_down": [_P, _P, _P, _P, c_int, c_int, c_int, c_float, c_int, c_float, _P],
    "sempyr_pool2_up": [_P, _P, _P, _P, c_int, c_int, c_int, c_float, c_int, c_float, _P, _P],
    "sempyr_maxpool3s2_fwd": [_P, _P, _P, c_int, c_int, c_int, _P],
    "sempyr_maxpool3s2_bwd": [_P, _P, _P, c_int, c_int, c_int, _P],
    "sempyr_reduce_scratch_floats": [],
    "sempyr_mean": [_P, c_int64, c_float, _P, _P, _P],
    "sempyr_fill_scalar": [_P, c_float, _P, c_int64, _P],
    "sempyr_l1_mean": [_P, _P, _P, c_int64, _P, _P, _P],
    "sempyr_l1_mean_bwd": [_P, _P, _P, _P, c_int64, _P, _P, _P],
    "sempyr_gp_fwd": [_P, c_int, c_int, _P, _P, _P, _P],
    "sempyr_gp_fwd_static": [_P, c_int, c_int, _P, _P, _P],
    "sempyr_gp_bwd": [_P, _P, _P, c_int, c_int, _P, _P],
    "sempyr_channel_sum": [_P, c_int, c_int, c_int, c_float, _P, c_int, _P, _P],
    "sempyr_replay_create": [_P, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_int32)],
    "sempyr_replay_create_on": [_P, c_int, ctypes.POINTER(c_void_p), c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_int32)],
    "sempyr_replay_launch": [_P, _P],
    "sempyr_replay_destroy": [_P],
    "sempyr_profile_begin": [c_int],
    "sempyr_profile_marker": [c_int, _P],
    "sempyr_profile_end": [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)],
    "sempyr_profile_hbm_report": [ctypes.c_char_p, ctypes.c_int64],
}


class SempyrError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C semantic_pyramid_for_image_generation_amd/csrc`). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.sempyr_last_error_string.argtypes = []
    lib.sempyr_last_error_string.restype = ctypes.c_char_p
    return lib


lib = _load()


def check(rc, what):
    """Raise on a non-zero return code of an entry point (never continue silently)."""
    if rc != 0:
        msg = lib.sempyr_last_error_string().decode("utf-8", "replace")
        raise SempyrError(f"{what} failed (rc={rc}): {msg}")
