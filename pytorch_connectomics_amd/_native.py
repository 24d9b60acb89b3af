"""ctypes binding of libpytc_hip.so (the C ABI declared in include/pytc_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, a RuntimeError
is raised.  The library is built in-tree by ``python -m pytorch_connectomics_amd.csrc.build``
(also run by ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libpytc_hip.so"
ABI_VERSION = 4          # include/pytc_hip.h PYTC_ABI_VERSION

F32, BF16 = 0, 1
OK = 0

VIEW_FLIP_Z, VIEW_FLIP_Y, VIEW_FLIP_X, VIEW_SWAP_YX, VIEW_SWAP_ZY, VIEW_SWAP_ZX = 1, 2, 4, 8, 16, 32
VIEW_SWAPS = {VIEW_SWAP_YX: (1, 2), VIEW_SWAP_ZY: (0, 1), VIEW_SWAP_ZX: (0, 2)}      # swap bit -> the two window axes it exchanges
PAD_MODES = {"constant": 0, "reflect": 1, "replicate": 2, "circular": 3}
BLEND_PRODUCT, BLEND_MIN = 0, 1
ACT_NONE, ACT_SIGMOID, ACT_TANH, ACT_GELU, ACT_SOFTMAX, ACT_RELU, ACT_LEAKY, ACT_ELU = 0, 1, 2, 3, 4, 5, 6, 7
RES_NONE, RES_ADD, RES_UPSAMPLE, RES_GELU_BWD, RES_NORM_BWD = 0, 1, 2, 3, 4


class PwArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w_packed", C.c_void_p), ("bias", C.c_void_p), ("ab", C.c_void_p),
        ("res", C.c_void_p), ("y", C.c_void_p),
        ("N", C.c_int), ("rows_per_sample", C.c_int64),
        ("C_in", C.c_int), ("C_out", C.c_int),
        ("in_dtype", C.c_int), ("out_dtype", C.c_int), ("w_dtype", C.c_int),
        ("act", C.c_int), ("res_mode", C.c_int), ("gather", C.c_int),
        ("Di", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int),
        ("res_low", C.c_void_p), ("res_bias", C.c_void_p), ("pre_act", C.c_int), ("w_paired", C.c_int),
    ]


class MlpArgs(C.Structure):
    _fields_ = [
        ("t", C.c_void_p), ("ab", C.c_void_p), ("w2_packed", C.c_void_p), ("b2", C.c_void_p),
        ("w3_packed", C.c_void_p), ("b3", C.c_void_p), ("res", C.c_void_p), ("res_low", C.c_void_p),
        ("res_bias", C.c_void_p), ("y", C.c_void_p),
        ("N", C.c_int), ("rows_per_sample", C.c_int64),
        ("C_in", C.c_int), ("C_hid", C.c_int), ("C_out", C.c_int), ("res_mode", C.c_int),
        ("Di", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int), ("w3_format", C.c_int), ("per_sample", C.c_int),
    ]


W3_BF16, W3_F16 = 0, 1
NORM_NONE, NORM_ZSCORE, NORM_MINMAX, NORM_DIVIDE = 0, 1, 2, 3
RAW_DTYPES = {"uint8": 0, "int8": 1, "uint16": 2, "int16": 3, "uint32": 4, "int32": 5, "float32": 6, "float64": 7}


class ReduceItem(C.Structure):
    _fields_ = [("part", C.c_void_p), ("out", C.c_void_p), ("n", C.c_int64), ("slots", C.c_int32), ("out_t", C.c_int32)]


class Conv3dArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w_packed", C.c_void_p), ("bias", C.c_void_p), ("ab", C.c_void_p), ("res", C.c_void_p),
        ("y", C.c_void_p),
        ("N", C.c_int), ("D", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C_in", C.c_int), ("C_out", C.c_int),
        ("kd", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
        ("act_in", C.c_int), ("act_param", C.c_float), ("res_mode", C.c_int), ("dtype", C.c_int),
    ]


_SIGS = {
    "pytc_abi_version": (C.c_int, []),
    "pytc_last_error": (C.c_char_p, []),
    "pytc_device_info": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "pytc_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "pytc_gather_windows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.c_void_p, C.c_int, C.c_void_p]),
    "pytc_blend_accumulate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_float, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                        C.c_int, C.c_void_p]),
    "pytc_blend_finalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_int, C.c_void_p]),
    "pytc_channel_activation": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_float, C.c_void_p]),
    "pytc_ensemble_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "pytc_dwconv3d_stat_slots": (C.c_int, [C.c_int] * 9),
    "pytc_dwconv3d_kernel_variant": (C.c_int, [C.c_int] * 9),
    "pytc_dwconv3d_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8
                          + [C.c_void_p]),
    "pytc_dwconv3d_fwd_wide": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8
                               + [C.c_void_p]),
    "pytc_dwconv3d_res_supported": (C.c_int, [C.c_int] * 7),
    "pytc_dwmix_supported": (C.c_int, [C.c_int] * 7),
    "pytc_dwmix_fwd": (C.c_int, [C.c_void_p] * 7 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 9 + [C.c_void_p]),
    "pytc_dwconv3d_fwd_res": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8
                              + [C.c_void_p]),
    "pytc_dwconvT3d_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7
                           + [C.c_void_p]),
    "pytc_groupnorm_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_float,
                                          C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pytc_groupnorm_fold_mlp": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pytc_pw_gemm_supported": (C.c_int, [C.c_int, C.c_int]),
    "pytc_pw_gemm_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p]),
    "pytc_pw_packed_elems": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "pytc_pw_pack_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "pytc_blend_accumulate_mapped": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                               C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p,
                                               C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pytc_blend_weight_shifted": (C.c_int, [C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_int, C.c_float, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                            C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pytc_normalize_covered": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "pytc_ensemble_update_masked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "pytc_ensemble_finalize_masked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "pytc_scale_cast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_int, C.c_void_p]),
    "pytc_resample_region": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]),
    "pytc_window_normalize_ws_elems": (C.c_int64, [C.c_int, C.c_int64]),
    "pytc_window_normalize": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "pytc_pw_conv_fwd": (C.c_int, [C.POINTER(PwArgs), C.c_void_p]),
    "pytc_pw_conv_paired_supported": (C.c_int, [C.POINTER(PwArgs)]),
    "pytc_pw_conv_rowmajor_supported": (C.c_int, [C.POINTER(PwArgs)]),
    "pytc_conv3d_packed_elems": (C.c_int64, [C.c_int] * 6),
    "pytc_conv3d_pack_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                          C.c_void_p]),
    "pytc_conv3d_fwd": (C.c_int, [C.POINTER(Conv3dArgs), C.c_void_p]),
    "pytc_channel_stats_slots": (C.c_int, [C.c_int64]),
    "pytc_channel_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "pytc_norm_finalize_groups": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_int,
                                            C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pytc_affine_act": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_float,
                                  C.c_int, C.c_void_p]),
    "pytc_maxpool3d_fwd": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 9 + [C.c_void_p]),
    "pytc_dwconvT3d_generic_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5
                                   + [C.POINTER(C.c_int32)] * 3 + [C.c_int, C.c_void_p]),
    "pytc_groupnorm_finalize_mr": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_float,
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pytc_gelu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "pytc_add_inplace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "pytc_copy_zero_front": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_void_p]),
    "pytc_pw_wgrad_slots": (C.c_int, [C.c_int64]),
    "pytc_pw_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pytc_pw_wgrad_partial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "pytc_pw_wgrad_dgrad_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pytc_pw_wgrad_dgrad_partial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64,
                                              C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "pytc_mixer_bwd_rc_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "pytc_mixer_bwd_rc_sps": (C.c_int, [C.c_int, C.c_int64, C.c_int]),
    "pytc_mixer_bwd_rc_ws_elems": (C.c_int64, [C.c_int, C.c_int64, C.c_int, C.c_int]),
    "pytc_mixer_bwd_rc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "pytc_dw_wgrad_partial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "pytc_reduce_slots_multi": (C.c_int, [C.POINTER(ReduceItem), C.c_int, C.c_void_p]),
    "pytc_dw_wgrad_slots": (C.c_int, [C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int,
                                      C.c_int]),
    "pytc_dw_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_void_p]),
    "pytc_pw_wgrad_groupnorm_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pytc_pw_wgrad_groupnorm_sps": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_int]),
    "pytc_pw_wgrad_groupnorm_ws_elems": (C.c_int64, [C.c_int, C.c_int64, C.c_int, C.c_int]),
    "pytc_pw_wgrad_groupnorm": (C.c_int, [C.c_void_p] * 6 + [C.c_float] + [C.c_void_p] * 3 + [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p]),
    "pytc_norm_bwd_apply": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_int,
                                      C.c_void_p, C.c_void_p]),
    "pytc_norm_bwd_ws_elems": (C.c_int, [C.c_int, C.c_int64, C.c_int]),
    "pytc_norm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int, C.c_int64, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "pytc_dwconv3d_bwd_data": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pytc_dwconv3d_bwd_data_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32),
                                             C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pytc_conv3d_wgrad_ws_elems": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int]),
    "pytc_conv3d_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_void_p]),
    "pytc_act_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                               C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "pytc_norm_finalize_groups_mr": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pytc_norm_bwd_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                                      C.c_int, C.c_void_p]),
    "pytc_act_norm_bwd_stats": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "pytc_act_norm_bwd_apply": (C.c_int, [C.c_void_p] * 7 + [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "pytc_act_norm_bwd_apply_cg": (C.c_int, [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 2 + [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "pytc_norm_bwd_apply_general": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "pytc_maxpool3d_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pytc_dwconv3d_generic_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int32), C.c_int, C.c_void_p]),
    "pytc_conv3d_direct_packed_elems": (C.c_int64, [C.c_int] * 6),
    "pytc_conv3d_pack_weight_direct": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                                 C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "pytc_conv3d_strided_fwd": (C.c_int, [C.POINTER(Conv3dArgs), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int32), C.c_int, C.c_void_p]),
    "pytc_convT3d_c1_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_int,
                                      C.c_int, C.c_void_p]),
    "pytc_convT3d_phase_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pytc_convT3d_phase_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pytc_convT3d_phase_fwd": (C.c_int, [C.POINTER(Conv3dArgs), C.POINTER(C.c_int32), C.c_void_p]),
    "pytc_conv3d_wgrad_strided_ws_elems": (C.c_int64, [C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    "pytc_conv3d_wgrad_strided": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int32), C.c_int, C.c_void_p]),
    "pytc_conv3d_pack_plan": (C.c_int, [C.c_int] * 7 + [C.c_void_p]),
    "pytc_conv3d_pack_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "pytc_conv3d_pack_weight_dgrad": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "pytc_norm_bwd_means": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_float, C.c_void_p]),
    "pytc_norm_bwd_means_cpg": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_float, C.c_void_p]),
    "pytc_norm_finalize_groups_cpg": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int,
                                                C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pytc_bn_train_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pytc_bn_train_finalize_cpad": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pytc_bn_update_running": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "pytc_layernorm_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "pytc_layernorm_rows_bwd_slots": (C.c_int, [C.c_int64, C.c_int, C.c_int]),
    "pytc_layernorm_rows_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float,
                                          C.c_int, C.c_void_p]),
    "pytc_grn_bwd_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int,
                                     C.c_void_p]),
    "pytc_bce_dice_ws_elems": (C.c_int64, [C.c_int, C.c_int, C.c_int64]),
    "pytc_bce_dice_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pytc_bce_dice_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_void_p]),
    "pytc_opt_chunk_elems": (C.c_int, []),
    "pytc_grad_norm_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pytc_adamw_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_void_p, C.c_void_p]),
    "pytc_pw_mlp_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pytc_pw_pack_weight_paired": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pytc_pack_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "pytc_pw_pack_weight_paired_f16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pytc_pw_mlp_fwd": (C.c_int, [C.POINTER(MlpArgs), C.c_void_p]),
    "pytc_pw_mlp_lds_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pytc_pw_mlp_lds_fwd": (C.c_int, [C.POINTER(MlpArgs), C.c_void_p]),
    "pytc_pw_mlp_dma_applies": (C.c_int, [C.POINTER(MlpArgs), C.c_int]),
    "pytc_pw_mlp_proj_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "pytc_pw_mlp_proj_fwd": (C.c_int, [C.POINTER(MlpArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pytc_pw_mlp_chunk_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pytc_pw_mlp_chunk_fwd": (C.c_int, [C.POINTER(MlpArgs), C.c_void_p]),
    "pytc_pw_mlp_stemres_fwd": (C.c_int, [C.POINTER(MlpArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pytc_stem_dwconv3d_stat_slots": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pytc_stem_dwconv3d_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pytc_convT3d_thin_supported": (C.c_int, [C.c_int, C.c_int]),
    "pytc_convT3d_thin_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int,
                                        C.c_int, C.c_void_p]),
    "pytc_stem_dwconv3d_mfma_image_bytes": (C.c_int, []),
    "pytc_stem_dwconv3d_pack_mfma": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pytc_stem_dwconv3d_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pytc_pw_mlp_train_fwd": (C.c_int, [C.POINTER(MlpArgs), C.c_void_p, C.c_void_p]),
    "pytc_pw_mlp_train_fwd_nostore": (C.c_int, [C.POINTER(MlpArgs), C.c_void_p]),
    "pytc_pw_mlp_bwd": (C.c_int, [C.POINTER(MlpArgs), C.c_void_p, C.c_void_p, C.c_void_p]),
    "pytc_pw_mlp_up_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pytc_pw_mlp_up_fwd": (C.c_int, [C.POINTER(MlpArgs), C.c_void_p, C.c_void_p, C.c_void_p]),
    "pytc_pw_mlp_head_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pytc_pw_mlp_head_fwd": (C.c_int, [C.POINTER(MlpArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
}

_lib = None


def exported_symbols():
    """Names every build of the library must export (checked by the CPU test-suite)."""
    return sorted(_SIGS)


def lib():
    """Load (once) and return the ctypes handle; raise loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"pytorch_connectomics_amd: HIP library {LIB_PATH} is missing. Build it with "
            "`python -m pytorch_connectomics_amd.csrc.build` (needs hipcc). There is no CPU fallback.")
    # torch first: it ships its own libamdhip64 -- loaded before ours, the library's HIP symbols bind to the runtime
    # PyTorch's tensors and streams live in; loaded after, the process ends up with two HIP runtimes and this library
    # sees no device
    import torch  # noqa: F401
    handle = C.CDLL(str(LIB_PATH))
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:  # pragma: no cover
            raise RuntimeError(f"{LIB_PATH} does not export {name}; rebuild the library") from e
        fn.restype = res
        fn.argtypes = args
    if handle.pytc_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libpytc_hip.so ABI version {handle.pytc_abi_version()} != {ABI_VERSION} (include/pytc_hip.h PYTC_ABI_VERSION); rebuild the library")
    _lib = handle
    # PYTC_TUNING="knob=value,knob=value": kernel-variant knobs (pytc_set_tuning) for A/B runs of unmodified commands
    for item in filter(None, (t.strip() for t in os.environ.get("PYTC_TUNING", "").split(","))):
        key, sep, val = item.partition("=")
        try:
            ok = bool(sep) and bool(key.strip()) and handle.pytc_set_tuning(key.strip().encode(), int(val)) == OK
        except ValueError:
            ok = False
        if not ok:
            raise RuntimeError(f"PYTC_TUNING: cannot set {item!r} (expected knob=integer[,knob=integer...])")
    return _lib


def check(status: int, what: str) -> None:
    if status != OK:
        msg = lib().pytc_last_error()
        raise RuntimeError(f"{what} failed (status {status}): {msg.decode() if msg else '?'}")
