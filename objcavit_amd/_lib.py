"""ctypes binding of libobjcavit_hip.so (declarations: include/objcavit_hip.h).

The library is the product: if it cannot be loaded the kernels raise -- there
is no CPU or PyTorch fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OCV_LIB_PATH") or os.path.join(_HERE, "lib", "libobjcavit_hip.so")   # env: diagnostic builds

_f32p = C.c_void_p      # device pointers travel as integers
_u8p = C.c_void_p
_stream = C.c_void_p
ABI_VERSION = 5          # include/objcavit_hip.h: OCV_ABI_VERSION


class EncoderLayerParams(C.Structure):
    """ocv_encoder_layer_params"""
    _fields_ = [("struct_size", C.c_size_t)] + [(n, C.c_void_p) for n in (
        "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "norm1_w", "norm1_b",
        "linear1_w", "linear1_b", "linear2_w", "linear2_b", "norm2_w", "norm2_b",
        "in_proj_p3", "out_proj_p3", "linear1_p3", "linear2_p3",
        "in_proj_h2", "out_proj_h2", "linear1_h2", "linear2_h2")]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(EncoderLayerParams)      # the library reads this many bytes, later fields = NULL


# name -> (restype, argtypes); every symbol declared in include/objcavit_hip.h
PROTOTYPES = {
    "ocv_abi_version": (C.c_int, []),
    "ocv_last_error": (C.c_char_p, []),
    "ocv_range_flag_set": (C.c_int, [C.c_void_p]),
    "ocv_range_flag_take_fwd": (C.c_int, [C.c_void_p, C.c_void_p, _stream]),
    "ocv_attention_set_fp32_range": (C.c_int, [C.c_int]),
    "ocv_attention_set_dispatch": (C.c_int, [C.c_int]),
    "ocv_linear_fwd": (C.c_int, [_f32p, C.c_int, C.c_long, _f32p, C.c_int, C.c_long, C.c_int, _f32p, _f32p, C.c_int,
                                 C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _stream]),
    "ocv_linear_residual_layernorm_fwd": (C.c_int, [_f32p, C.c_int, _f32p, C.c_int, _f32p, _f32p, C.c_int, _f32p, _f32p,
                                                    C.c_float, _u8p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _stream]),
    "ocv_layernorm_residual_fwd": (C.c_int, [_f32p, _f32p, _f32p, _f32p, C.c_float, _f32p, C.c_int, C.c_int, _stream]),
    "ocv_attention_fwd": (C.c_int, [_f32p, C.c_long, C.c_int, _f32p, C.c_long, C.c_int, _f32p, C.c_long, C.c_int, _u8p,
                                    _f32p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _stream]),
    "ocv_mha_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ocv_mha_fwd": (C.c_int, [_f32p, _f32p, _f32p, _u8p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, _stream]),
    "ocv_mha_split3_fwd": (C.c_int, [_f32p, _f32p, _f32p, _u8p, C.c_void_p, _f32p, C.c_void_p, _f32p, _f32p, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, _stream]),
    "ocv_split_h2_packed_elems": (C.c_size_t, [C.c_int, C.c_int]),
    "ocv_pack_split_h2_fwd": (C.c_int, [_f32p, C.c_int, C.c_int, C.c_int, C.c_void_p, _stream]),
    "ocv_mha_few_keys_h2_workspace_bytes": (C.c_size_t, [C.c_int]),
    "ocv_mha_few_keys_h2_set_dispatch": (C.c_int, [C.c_int]),
    "ocv_mha_few_keys_h2_fwd": (C.c_int, [_f32p, _f32p, _f32p, _u8p, C.c_void_p, _f32p, C.c_void_p, _f32p, _f32p, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, _stream]),
    "ocv_ffn_residual_layernorm_fwd": (C.c_int, [_f32p] * 7 + [C.c_float, _u8p, _f32p, C.c_int, C.c_int, C.c_int, _stream]),
    "ocv_split3_packed_elems": (C.c_size_t, [C.c_int, C.c_int]),
    "ocv_pack_split3_fwd": (C.c_int, [_f32p, C.c_int, C.c_int, C.c_int, C.c_void_p, _stream]),
    "ocv_linear_split3_fwd": (C.c_int, [_f32p, C.c_int, C.c_void_p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _stream]),
    "ocv_linear_residual_layernorm_split3_fwd": (C.c_int, [_f32p, C.c_int, C.c_void_p, _f32p, _f32p, C.c_int, _f32p, _f32p, C.c_float,
                                                           _u8p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _stream]),
    "ocv_ffn_split3_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "ocv_ffn_residual_layernorm_split3_fwd": (C.c_int, [_f32p, C.c_void_p, _f32p, C.c_void_p, _f32p, _f32p, _f32p, C.c_float, _u8p, _f32p,
                                                        C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, _stream]),
    "ocv_layer_tail_split3_fwd": (C.c_int, [_f32p, _f32p, C.POINTER(EncoderLayerParams), C.c_void_p, _f32p, C.c_float, _u8p, _f32p, _f32p,
                                            C.c_int, C.c_int, C.c_int, _stream]),
    "ocv_layer_tail_h2_fwd": (C.c_int, [_f32p, _f32p, C.POINTER(EncoderLayerParams), C.c_void_p, _f32p, C.c_float, _u8p, _f32p, _f32p,
                                            C.c_int, C.c_int, C.c_int, _stream]),
    "ocv_layer_tail_h2_groups": (C.c_int, [C.c_int, C.c_int]),
    "ocv_layer_tail_h2_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "ocv_layer_tail_h2_ws_fwd": (C.c_int, [_f32p, _f32p, C.POINTER(EncoderLayerParams), C.c_void_p, _f32p, C.c_float, _u8p, _f32p, _f32p,
                                           C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, _stream]),
    "ocv_encoder_stack_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ocv_encoder_stack_fwd": (C.c_int, [_f32p, C.POINTER(EncoderLayerParams), C.c_int, _u8p, C.c_int, _f32p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_size_t, _stream]),
    "ocv_encoder_layer_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ocv_encoder_layer_fwd": (C.c_int, [_f32p, C.POINTER(EncoderLayerParams), _u8p, C.c_int, _f32p, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_size_t, _stream]),
    "ocv_pointwise_conv_nhwc_split_hl_fwd": (C.c_int, [_f32p, _f32p, C.c_int, C.c_void_p, _f32p, _f32p, _f32p, C.c_void_p, C.c_long, C.c_int,
                                                       C.c_int, C.c_int, _stream]),
    "ocv_pointwise_split_workspace_bytes": (C.c_size_t, [C.c_long, C.c_int, C.c_int]),
    "ocv_pointwise_conv_nhwc_split_ws_fwd": (C.c_int, [_f32p, _f32p, C.c_int, C.c_void_p, _f32p, _f32p, _f32p, C.c_void_p, C.c_long, C.c_int,
                                                       C.c_int, C.c_int, C.c_void_p, C.c_size_t, _stream]),
    "ocv_pointwise_hl_set_dispatch": (C.c_int, [C.c_int, C.c_int]),
    "ocv_pointwise_hl_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_int, _f32p, _f32p, _f32p, C.c_void_p, C.c_long,
                                       C.c_int, C.c_int, _stream]),
    "ocv_depthwise_conv_nhwc_sum_hl_fwd": (C.c_int, [_f32p, _f32p, _f32p, _f32p, C.c_void_p, _f32p] + [C.c_int] * 10 + [_stream]),
    "ocv_se_gate_weights_fwd": (C.c_int, [_f32p, C.c_int, C.c_long, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_void_p, C.c_long, _f32p,
                                          _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _stream]),
    "ocv_patch_embed_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ocv_patch_embed_split_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ocv_patch_embed_split_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, _f32p, C.c_int, _f32p, _f32p, C.c_long, _f32p, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, _stream]),
    "ocv_patch_embed_fwd": (C.c_int, [_f32p, C.c_int, _f32p, _f32p, _f32p, C.c_long, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_void_p, C.c_size_t, _stream]),
    "ocv_pixel_dot_fwd": (C.c_int, [_f32p, C.c_int, _f32p, C.c_long, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _stream]),
    "ocv_bin_head_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ocv_bin_head_fold_fwd": (C.c_int, [_f32p, C.c_long, C.c_int, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _stream]),
    "ocv_bin_head_folded_fwd": (C.c_int, [_f32p, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _stream]),
    "ocv_bin_head_partials_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "ocv_bin_head_folded_ws_fwd": (C.c_int, [_f32p, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_void_p, C.c_size_t, _stream]),
    "ocv_bin_head_fwd": (C.c_int, [_f32p, C.c_int, _f32p, C.c_long, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, _stream]),
    "ocv_depthwise_conv_fwd": (C.c_int, [_f32p, _f32p, _f32p, _f32p] + [C.c_int] * 11 + [_stream]),
    "ocv_depthwise_sum_tiles": (C.c_int, [C.c_int] * 6),
    "ocv_depthwise_conv_nhwc_sum_fwd": (C.c_int, [_f32p] * 5 + [C.c_int] * 10 + [_stream]),
    "ocv_se_gate_partials_fwd": (C.c_int, [_f32p, C.c_int, C.c_long, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int,
                                           C.c_int, C.c_int, _stream]),
    "ocv_depth_metrics_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "ocv_depth_metrics_fwd": (C.c_int, [_f32p, _f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_float, C.c_float] + [C.c_int] * 4 +
                              [C.c_long, _f32p, C.c_int, C.c_void_p, C.c_size_t, _stream]),
    "ocv_conv3x3_few_channels_fwd": (C.c_int, [_f32p, C.c_long, C.c_long, C.c_long, C.c_long, _f32p, _f32p] + [C.c_int] * 5 + [_stream]),
    "ocv_stem_conv_fwd": (C.c_int, [_f32p, _f32p, _f32p, _f32p] + [C.c_int] * 12 + [_stream]),
    "ocv_pointwise_packed_weight_elems": (C.c_size_t, [C.c_int, C.c_int]),
    "ocv_pointwise_split_set_dispatch": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "ocv_pointwise_conv_nhwc_split_fwd": (C.c_int, [_f32p, _f32p, C.c_int, C.c_void_p, _f32p, _f32p, _f32p,
                                                    C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ocv_pointwise_conv_nhwc_fwd": (C.c_int, [_f32p, _f32p, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_long, C.c_int, C.c_int,
                                              C.c_int, _stream]),
    "ocv_depthwise_conv_nhwc_fwd": (C.c_int, [_f32p, _f32p, _f32p, _f32p] + [C.c_int] * 11 + [_stream]),
    "ocv_channel_mean_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_long]),
    "ocv_channel_mean_nhwc_fwd": (C.c_int, [_f32p, _f32p, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_size_t, _stream]),
    "ocv_se_gate_fwd": (C.c_int, [_f32p] * 7 + [C.c_int] * 3 + [_stream]),
    "ocv_conv_nhwc_fwd": (C.c_int, [_f32p, C.c_int, _f32p, C.c_int, C.c_void_p, C.c_void_p, _f32p, _f32p, _f32p] +
                          [C.c_int] * 6 + [_stream]),
    "ocv_conv_nhwc_exact_fwd": (C.c_int, [_f32p, C.c_int, _f32p, C.c_int, _f32p, _f32p, _f32p, _f32p] + [C.c_int] * 6 + [_stream]),
    "ocv_mbconv_expand_dw_tiles": (C.c_int, [C.c_int] * 4),
    "ocv_mbconv_expand_dw_fwd": (C.c_int, [_f32p, C.c_void_p, _f32p, _f32p, _f32p, _f32p, _f32p] + [C.c_int] * 11 + [_stream]),
    "ocv_regressor_bins_fwd": (C.c_int, [_f32p, C.c_long, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                         C.c_int, C.c_float, C.c_float, _f32p, _f32p, _f32p, C.c_int, _stream]),
    "ocv_bin_edges_fwd": (C.c_int, [_f32p, C.c_int, C.c_float, C.c_float, _f32p, _f32p, _f32p, C.c_int, C.c_int, _stream]),
    "ocv_object_tokens_pad_fwd": (C.c_int, [_f32p, C.c_void_p, C.c_float, _f32p, _u8p, C.c_int, C.c_int, C.c_int, _stream]),
    "ocv_object_front_pad_fwd": (C.c_int, [_f32p, C.c_void_p, C.c_int, C.c_int, C.c_float, _f32p, _u8p] + [C.c_int] * 4 + [_stream]),
    "ocv_split_act_elems": (C.c_size_t, [C.c_int] * 4),
    "ocv_conv_nhwc_split_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, _f32p, _f32p, _f32p, C.c_void_p]
                                + [C.c_int] * 6 + [_stream]),
    "ocv_conv_nhwc_split_workspace_bytes": (C.c_size_t, [C.c_int] * 6),
    "ocv_conv_nhwc_split_ws_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, _f32p, _f32p, _f32p, C.c_void_p]
                                   + [C.c_int] * 6 + [C.c_void_p, C.c_size_t, _stream]),
    "ocv_conv_nhwc_split_x_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, _f32p, C.c_int, _f32p, _f32p, _f32p, C.c_void_p]
                                  + [C.c_int] * 6 + [C.c_void_p, C.c_size_t, _stream]),
    "ocv_conv3x3_packed_taps_k": (C.c_int, [C.c_int]),
    "ocv_conv3x3_split_packed_taps_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, _f32p, C.c_int, _f32p, _f32p, _f32p, C.c_void_p]
                                          + [C.c_int] * 5 + [_stream]),
    "ocv_upsample_concat_split_x_fwd": (C.c_int, [_f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_void_p, C.c_int,
                                                  C.c_int, C.c_int, C.c_int, _stream]),
    "ocv_tap_interp_combine_x_fwd": (C.c_int, [_f32p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_void_p] + [C.c_int] * 6 + [_stream]),
    "ocv_pos_grid_sample_fwd": (C.c_int, [_f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                          C.c_int, _f32p, _f32p, _stream]),
    "ocv_conv3x3_winograd43_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "ocv_conv3x3_winograd43_split_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, _f32p, _f32p, _f32p, _f32p, C.c_void_p] + [C.c_int] * 6 +
                                         [C.c_void_p, C.c_size_t, _stream]),
    "ocv_tap_interp_supported": (C.c_int, [C.c_int] * 5),
    "ocv_tap_interp_combine_fwd": (C.c_int, [_f32p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_void_p] + [C.c_int] * 5 + [_stream]),
    "ocv_upsample_concat_split_fwd": (C.c_int, [_f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_void_p,
                                                C.c_int, C.c_int, C.c_int, _stream]),
}

_lib: Optional[C.CDLL] = None


class HipLibraryError(RuntimeError):
    pass


def load(path: Optional[str] = None) -> C.CDLL:
    """Load the library (once) and attach the prototypes.

    ``import torch`` must come first: libobjcavit_hip.so needs libamdhip64.so.7
    and has to share the HIP runtime instance PyTorch-ROCm already loaded
    (same SONAME -> the dynamic loader reuses it), otherwise torch's streams
    and device pointers would be foreign to our launches.
    """
    global _lib
    if _lib is not None and path is None:
        return _lib
    import torch  # noqa: F401  (loads torch/lib/libamdhip64.so first)
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise HipLibraryError(
            f"{p} not found: build it with `python -m objcavit_amd.build` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the ObjCAViT hot path.")
    try:
        lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise HipLibraryError(f"cannot load {p}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{p} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ocv_abi_version() != ABI_VERSION:
        raise HipLibraryError(f"ABI version mismatch: library {lib.ocv_abi_version()}, binding {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ocv_last_error().decode(errors="replace")
        raise HipLibraryError(f"{what} failed (rc={rc}): {msg}")
