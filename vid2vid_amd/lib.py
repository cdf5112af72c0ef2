"""ctypes binding of libv2v_hip.so (C ABI declared in include/v2v_hip.h).

The library is the product; this module is plumbing.  There is NO fallback: if the shared
object is missing or a symbol cannot be resolved the import fails loudly, and every
non-zero return code becomes a RuntimeError carrying v2v_last_error() -- the same contract
the reference's pybind11 ops have (correlation_cuda.cc:81-83 -> AT_ERROR -> RuntimeError).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# V2V_LIB_PATH: another BUILD of the same library (A/B measurements of two builds on one box, scripts/gpu_r5.sh); never a fallback
LIB_PATH = os.environ.get("V2V_LIB_PATH") or os.path.join(_HERE, "libv2v_hip.so")

# dtype / mode codes (include/v2v_hip.h)
F32, BF16 = 0, 1
PAD_ZERO, PAD_REFLECT = 0, 1
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
OUT_RAW_F32_NHWC, OUT_ACT_NHWC, OUT_F32_NCHW, OUT_NORM_ACT_NHWC, OUT_RAW_ACT_NHWC = 0, 1, 2, 3, 4


class ConvDesc(C.Structure):
    """struct v2v_conv_desc"""
    _fields_ = [
        ("in_", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
        ("stats", C.c_void_p), ("zero_page", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("cin", C.c_int32), ("cin_stride", C.c_int32),
        ("cout", C.c_int32), ("cout_stride", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32),
        ("stride", C.c_int32), ("pad", C.c_int32), ("pad_mode", C.c_int32),
        ("transposed", C.c_int32),
        ("OH", C.c_int32), ("OW", C.c_int32),
        ("dtype", C.c_int32), ("out_mode", C.c_int32), ("act", C.c_int32),
        ("act_param", C.c_float), ("out_scale", C.c_float),
        ("tile", C.c_int32),
        ("fin_counter", C.c_void_p), ("fin_gamma", C.c_void_p), ("fin_beta", C.c_void_p),
        ("fin_scale_shift", C.c_void_p), ("fin_running_mean", C.c_void_p), ("fin_running_var", C.c_void_p),
        ("fin_eps", C.c_float), ("fin_momentum", C.c_float), ("fin_count", C.c_int64),
        ("splitk", C.c_int32), ("prefetch", C.c_int32), ("slabs", C.c_void_p), ("sk_counter", C.c_void_p),
        ("w_korder", C.c_int32), ("ablate", C.c_int32),
        ("res0", C.c_void_p), ("res1", C.c_void_p),
        ("act_split", C.c_int32), ("act_b", C.c_int32), ("act_param_b", C.c_float), ("out_scale_b", C.c_float),
        ("fin_workspace", C.c_void_p),
    ]


class OneHotNorm(C.Structure):
    """struct v2v_onehot_norm"""
    _fields_ = [
        ("counter", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("scale_shift", C.c_void_p),
        ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("eps", C.c_float), ("momentum", C.c_float),
        ("count", C.c_int64),
    ]


class WgradDesc(C.Structure):
    """struct v2v_wgrad_desc"""
    _fields_ = [
        ("p", C.c_void_p), ("q", C.c_void_p), ("grad", C.c_void_p), ("workspace", C.c_void_p), ("zero_page", C.c_void_p),
        ("N", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32), ("QH", C.c_int32), ("QW", C.c_int32),
        ("rows", C.c_int32), ("cols", C.c_int32), ("p_stride", C.c_int32), ("q_stride", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("pad_mode", C.c_int32),
        ("dtype", C.c_int32), ("accumulate", C.c_int32),
    ]


LOSS_MSE_CONST, LOSS_L1 = 0, 1

# name -> (restype, argtypes); mirrors include/v2v_hip.h one to one
_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
PROTOTYPES = {
    "v2v_conv_packed_elems": (_L, [_I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "v2v_conv_pack_weights": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "v2v_conv_stats_rows": (C.c_int, [C.POINTER(ConvDesc)]),
    "v2v_conv_debug_clocks": (C.c_int, [_P]),
    "v2v_conv_tile_config": (C.c_int, [C.POINTER(ConvDesc)]),
    "v2v_conv_fused_norm_max_workgroups": (C.c_int, []),
    "v2v_fastdiv_magic": (C.c_int, [C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(_I)]),
    "v2v_conv_splitk_workspace": (_L, [C.POINTER(ConvDesc), C.POINTER(_I)]),
    "v2v_conv2d": (C.c_int, [C.POINTER(ConvDesc), _P]),
    "v2v_conv2d_pair": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _P]),
    "v2v_conv_wgrad_workspace": (_L, [C.POINTER(WgradDesc)]),
    "v2v_conv_wgrad": (C.c_int, [C.POINTER(WgradDesc), _P]),
    "v2v_bn_backward_rows": (C.c_int, [_L]),
    "v2v_bn_backward": (C.c_int, [_P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _L, _I, _I, _I, _F, _I, _P]),
    "v2v_channel_sum": (C.c_int, [_P, _P, _I, _P, _L, _I, _I, _I, _P]),
    "v2v_act_backward": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _I, _P]),
    "v2v_avgpool3s2_nhwc_backward": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "v2v_pack_concat_nhwc": (C.c_int, [_P, _I, _P, _I, _F, _P, _I, _I, _I, _I, _I, _P]),
    "v2v_unpack_channels_nchw": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "v2v_reflect_pad_fold": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "v2v_warp_blend_backward": (C.c_int, [_P] * 15 + [_I] * 5 + [_P]),
    "v2v_resample_flow_backward": (C.c_int, [_P] * 7 + [_I] * 5 + [_P]),
    "v2v_flownet_normalize": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "v2v_warp_diff_norm": (C.c_int, [_P, _L, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "v2v_resize_planar": (C.c_int, [_P, _P, _L, _I, _I, _I, _I, _I, _F, _P]),
    "v2v_pack_channels_nhwc": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _I, _P]),
    "v2v_concat_channels_nhwc": (C.c_int, [_P, _I, _I, _P, _I, _I, _I, _L, _I, _P]),
    "v2v_loss_workspace_floats": (C.c_int, []),
    "v2v_loss_forward": (C.c_int, [_I, _P, _P, _P, _F, _F, _L, _I, _I, _L, _L, _L, _I, _P, _P, _I, _P]),
    "v2v_loss_backward": (C.c_int, [_I, _P, _P, _P, _F, _F, _L, _I, _I, _L, _L, _L, _I, _P, _P, _I, _P]),
    "v2v_adam_step": (C.c_int, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _I, _P]),
    "v2v_adam_step_dev": (C.c_int, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _P, _P]),
    "v2v_memset_zero": (C.c_int, [_P, _L, _P]),
    "v2v_bn_finalize": (C.c_int, [_P, _I, _I, _L, _P, _P, _F, _P, _P, _P, _F, _P, _P]),
    "v2v_bn_finalize_groups": (C.c_int, [_I]),
    "v2v_bn_apply": (C.c_int, [_P, _I, _P, _P, _P, _P, _L, _I, _I, _I, _F, _I, _P]),
    "v2v_bn_apply_pair": (C.c_int, [_P] * 10 + [_I, _L, _I, _I, _I, _F, _I, _P]),
    "v2v_bn_apply_raw": (C.c_int, [_P, _I, _I, _P, _P, _P, _P, _L, _I, _I, _I, _F, _I, _P]),
    "v2v_avgpool3s2_planar": (C.c_int, [_P, _P, _L, _I, _I, _P]),
    "v2v_avgpool3s2_nhwc": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "v2v_maxpool2_nhwc": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "v2v_maxpool2_nhwc_backward": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "v2v_avgpool2_planar": (C.c_int, [_P, _P, _L, _I, _I, _P]),
    "v2v_avgpool2_planar_backward": (C.c_int, [_P, _P, _L, _I, _I, _P]),
    "v2v_onehot_planar": (C.c_int, [_P, _P, _P, _I, _I, _I, _P]),
    "v2v_onehot_planar_u8": (C.c_int, [_P, _P, _P, _I, _I, _I, _P]),
    "v2v_instance_mean_workspace": (_L, [_I, _L]),
    "v2v_instance_mean_planar": (C.c_int, [_P, _P, _P, _P, _I, _L, _P]),
    "v2v_tensor2im": (C.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "v2v_tensor2label": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "v2v_tensor2flow": (C.c_int, [_P, _P, _P, _I, _I, _P]),
    "v2v_onehot_conv_table_bytes": (_L, [_I, _I, _I, _I, _I, _I]),
    "v2v_onehot_conv_pack_weights": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "v2v_onehot_conv_stats_rows": (C.c_int, [_I, _I]),
    "v2v_label_codes": (C.c_int, [_P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "v2v_onehot_conv7x7": (C.c_int, [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "v2v_onehot_conv7x7_norm": (C.c_int, [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(OneHotNorm), _P]),
    "v2v_encode_labels": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "v2v_encode_labels_u8": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "v2v_encode_labels_pooled": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P]),
    "v2v_fg_mask_nhwc": (C.c_int, [_P, _P, _L, _I, _I, _P, _I, _I, _P]),
    "v2v_pack_nchw_to_nhwc": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "v2v_unpack_nhwc_to_nchw": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "v2v_add_nhwc": (C.c_int, [_P, _P, _P, _L, _I, _P]),
    "v2v_warp_blend": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "v2v_resample_flow": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "v2v_correlation_out_size": (C.c_int, [_I, _I, _I, _I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "v2v_correlation_forward": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "v2v_correlation_nhwc": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "v2v_resample2d_forward": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "v2v_channelnorm_forward": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "v2v_plan_create": (_P, []),
    "v2v_plan_destroy": (None, [_P]),
    "v2v_plan_begin_record": (C.c_int, [_P]),
    "v2v_plan_end_record": (C.c_int, [_P]),
    "v2v_plan_num_ops": (C.c_int, [_P]),
    "v2v_plan_run": (C.c_int, [_P, _P]),
    "v2v_plan_instantiate_graph": (C.c_int, [_P, _P]),
    "v2v_plan_launch_graph": (C.c_int, [_P, _P]),
    "v2v_plan_segment_program": (C.c_int, [_P, C.POINTER(_I), _I, C.POINTER(_I), _I]),
    "v2v_plan_profile": (C.c_int, [_P, _P, C.POINTER(C.c_float), _I]),
    "v2v_plan_timeline": (C.c_int, [_P, _P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(_I), _I]),
    "v2v_plan_timeline_graph": (C.c_int, [_P, _P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(_I), _I]),
    "v2v_plan_op_name": (C.c_char_p, [_P, _I]),
    "v2v_plan_op_lane": (C.c_int, [_P, _I]),
    "v2v_plan_set_label": (C.c_int, [_P, C.c_char_p]),
    "v2v_plan_op_label": (C.c_char_p, [_P, _I]),
    "v2v_plan_set_lane": (C.c_int, [_I]),
    "v2v_plan_lane_wait": (C.c_int, [_I, _I]),
    "v2v_memcpy_d2d": (C.c_int, [_P, _P, _L, _P]),
    "v2v_split_x3": (C.c_int, [_P, _P, _L, _I, _I, _I, _P]),
    "v2v_bn_apply_x3": (C.c_int, [_P] * 12 + [_I, _L, _I, _I, _F, _P]),
    "v2v_set_dry_run": (C.c_int, [_I]),
    "v2v_get_dry_run": (C.c_int, []),
    "v2v_version": (C.c_int, []),
    "v2v_last_error": (C.c_char_p, []),
    "v2v_device_status": (C.c_int, [_I]),
    "v2v_device_info": (C.c_int, [C.POINTER(_I), C.POINTER(_I), C.POINTER(_L), C.c_char_p, _I]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "vid2vid_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C vid2vid_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc, what=""):
    """Raise RuntimeError for a non-zero v2v_* return code."""
    if rc != 0:
        msg = lib.v2v_last_error()
        raise RuntimeError("v2v %s failed (code %d): %s" % (what, rc, msg.decode() if msg else ""))


def exported_symbols():
    return sorted(PROTOTYPES.keys())
