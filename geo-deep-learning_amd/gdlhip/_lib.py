"""ctypes binding of libgdlhip.so (the C-ABI declared in include/gdlhip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is NO
fallback: if the shared object is missing, every op raises -- the product path never routes
through PyTorch math or the CPU oracle.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

# GDL_LIB_PATH: an A/B build of the same sources (csrc/Makefile: BUILD= OUT= EXTRA=), for same-box comparisons only
LIB_PATH = Path(os.environ.get("GDL_LIB_PATH") or Path(__file__).resolve().parent / "libgdlhip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_MUL_GELU_GRAD, ACT_RESID_RELU = 0, 1, 2, 3, 4

c_i, c_l, c_f, c_p = C.c_int, C.c_int64, C.c_float, C.c_void_p


class ConvArgs(C.Structure):
    """Mirror of gdl_conv_args (include/gdlhip.h)."""

    _fields_ = [
        ("inp", c_p), ("dtype", c_i), ("B", c_i), ("H", c_i), ("W", c_i), ("C", c_i),
        ("in_sB", c_l), ("in_sH", c_l), ("in_sW", c_l),
        ("Ho", c_i), ("Wo", c_i), ("R", c_i), ("S", c_i), ("stride", c_i), ("pad", c_i),
        ("w", c_p), ("w_sN", c_l), ("N", c_i),
        ("out", c_p), ("out_dtype", c_i), ("out_sB", c_l), ("out_sH", c_l), ("out_sW", c_l),
        ("alpha", c_f), ("bias", c_p), ("scale", c_p), ("shift", c_p), ("act", c_i),
        ("batch_scale", c_p), ("resid", c_p), ("resid_dtype", c_i),
        ("res_sB", c_l), ("res_sH", c_l), ("res_sW", c_l),
        ("nz", c_i), ("nz_inner", c_i),
        ("in_sZ0", c_l), ("in_sZ1", c_l), ("w_sZ0", c_l), ("w_sZ1", c_l),
        ("out_sZ0", c_l), ("out_sZ1", c_l),
        ("aux_out", c_p), ("stats_partial", c_p),
    ]


class WgradArgs(C.Structure):
    """Mirror of gdl_wgrad_args (include/gdlhip.h)."""

    _fields_ = [
        ("inp", c_p), ("dy", c_p), ("dtype", c_i), ("B", c_i), ("H", c_i), ("W", c_i), ("C", c_i),
        ("in_sB", c_l), ("in_sH", c_l), ("in_sW", c_l),
        ("Ho", c_i), ("Wo", c_i), ("R", c_i), ("S", c_i), ("stride", c_i), ("pad", c_i),
        ("N", c_i), ("dy_sB", c_l), ("dy_sH", c_l), ("dy_sW", c_l),
        ("dw", c_p), ("dw_sN", c_l), ("accumulate", c_i),
        ("workspace", c_p), ("workspace_bytes", c_l),
        ("nz", c_i), ("nz_inner", c_i),
        ("in_sZ0", c_l), ("in_sZ1", c_l), ("dy_sZ0", c_l), ("dy_sZ1", c_l), ("dw_sZ0", c_l), ("dw_sZ1", c_l),
    ]


# name -> (restype, argtypes); every symbol declared in include/gdlhip.h
SIGNATURES = {
    "gdl_version": (c_i, []),
    "gdl_last_error": (C.c_char_p, []),
    "gdl_conv_gemm": (c_i, [C.POINTER(ConvArgs), c_p]),
    "gdl_conv_gemm_plan": (c_i, [C.POINTER(ConvArgs), C.POINTER(c_l)]),
    "gdl_conv_gemm_stats_rows": (c_l, [C.POINTER(ConvArgs)]),
    "gdl_conv_wgrad_workspace": (c_l, [C.POINTER(WgradArgs)]),
    "gdl_conv_wgrad": (c_i, [C.POINTER(WgradArgs), c_p]),
    "gdl_layernorm_fwd": (c_i, [c_p, c_l, c_p, c_p, c_p, c_i, c_l, c_i, c_f, c_p]),
    "gdl_colreduce_workspace": (c_l, [c_l, c_i, c_i]),
    "gdl_layernorm_bwd": (c_i, [c_p, c_l, c_p, c_i, c_p, c_p, c_l, c_p, c_l, c_l, c_i, c_f, c_p, c_p, c_i, c_p, c_l,
                                c_p]),
    "gdl_colsum": (c_i, [c_p, c_i, c_l, c_i, c_l, c_p, c_i, c_p, c_l, c_p]),
    "gdl_layerscale_bwd": (c_i, [c_p, c_p, c_i, c_p, c_p, c_l, c_l, c_i, c_p, c_i, c_p, c_i, c_p, c_l, c_p]),
    "gdl_softmax_bwd_rows": (c_i, [c_p, c_p, c_p, c_i, c_l, c_i, c_i, c_f, c_p]),
    "gdl_dwconv3x3_gelu_bwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_l,
                                     c_p]),
    "gdl_col2im": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_l, c_l, c_l, c_p]),
    "gdl_chan_weights_fwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_p]),
    "gdl_chan_weights_bwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                   c_p]),
    "gdl_chan_pool_fwd": (c_i, [c_p, c_i, c_i, c_l, c_i, c_i, c_p, c_p, c_l, c_p, c_p, c_f, c_p, c_p, c_p]),
    "gdl_chan_pool_workspace": (c_l, [c_i, c_i, c_l, c_i, c_i]),
    "gdl_chan_pool_bwd": (c_i, [c_p, c_i, c_i, c_l, c_i, c_i, c_p, c_p, c_l, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_l,
                                c_p]),
    "gdl_maxpool3x3s2_fwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_p]),
    "gdl_maxpool3x3s2_bwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_l, c_l, c_l, c_l, c_l, c_l,
                                   c_p]),
    "gdl_nearest2x_fwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_p]),
    "gdl_nearest2x_bwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_p]),
    "gdl_add_relu": (c_i, [c_p, c_p, c_p, c_i, c_l, c_p]),
    "gdl_relu_bwd": (c_i, [c_p, c_p, c_p, c_i, c_l, c_p]),
    "gdl_pad_channels": (c_i, [c_p, c_i, c_l, c_i, c_p, c_i, c_i, c_p]),
    "gdl_bn_stats": (c_i, [c_p, c_i, c_l, c_i, c_l, c_p, c_p, c_p, c_p, c_f, c_p, c_l, c_p]),
    "gdl_bn_stats_workspace": (c_l, [c_l, c_i]),
    "gdl_bn_apply": (c_i, [c_p, c_p, c_i, c_l, c_i, c_l, c_l, c_p, c_p, c_p, c_p, c_f, c_i, c_p]),
    "gdl_bn_bwd_reduce": (c_i, [c_p, c_p, c_i, c_l, c_i, c_l, c_l, c_p, c_p, c_p, c_p, c_f, c_i,
                                c_p, c_p, c_p, c_l, c_p]),
    "gdl_bn_bwd_dx": (c_i, [c_p, c_p, c_p, c_i, c_l, c_i, c_l, c_l, c_l, c_p, c_p, c_p, c_p, c_f,
                            c_i, c_p, c_p, c_l, c_p]),
    "gdl_syncbn_pack": (c_i, [c_p, c_p, C.c_double, c_i, c_p, c_p]),
    "gdl_syncbn_unpack": (c_i, [c_p, c_i, c_p, c_p, c_p, c_p, c_f, c_p]),
    "gdl_bn_bwd_dx_sync": (c_i, [c_p, c_p, c_p, c_i, c_l, c_i, c_l, c_l, c_l, c_p, c_p, c_p, c_p, c_f, c_i, c_p, c_p, c_p, c_p]),
    "gdl_bn_small_fwd": (c_i, [c_p, c_p, c_i, c_l, c_i, c_l, c_l, c_p, c_p, c_f, c_i, c_p, c_p, c_p, c_p, c_f, c_p]),
    "gdl_bn_small_bwd": (c_i, [c_p, c_p, c_p, c_i, c_l, c_i, c_l, c_l, c_l, c_p, c_p, c_p, c_p, c_f, c_i, c_p, c_p, c_p]),
    "gdl_bilinear_fwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_p, c_i, c_i, c_i,
                               c_l, c_l, c_l, c_i, c_p]),
    "gdl_bilinear_fwd_add": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_p, c_p, c_i, c_i, c_i,
                                   c_l, c_l, c_l, c_p]),
    "gdl_resize_conv3x3_bwd_gather": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p]),
    "gdl_resize_conv3x3_bwd_gather_one_pass": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "gdl_resize_conv3x3_bwd_gather_bn": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_i, c_p, c_p,
                                               c_l, c_p, c_p]),
    "gdl_resize_conv3x3_bwd_gather_workspace": (c_l, [c_i, c_i, c_i, c_i, c_i]),
    "gdl_resize_conv3x3_bwd_gather2": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_l, c_p]),
    "gdl_resize_conv3x3_fwd_sum": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_p]),
    "gdl_resize_conv3x3_fwd_sum_any": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    "gdl_resize_conv3x3_fwd_sum_bn_rows": (c_l, [c_i, c_i, c_i]),
    "gdl_resize_conv3x3_fwd_sum_bn": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_f, c_p]),
    "gdl_bn_stats_finalize": (c_i, [c_p, c_i, c_i, c_l, c_p, c_p, c_p, c_p, c_f, c_p]),
    "gdl_copy_cast": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_p, c_i, c_l, c_l, c_l, c_p]),
    "gdl_bilinear_sum_fwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p]),
    "gdl_bilinear_bwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_p, c_i, c_i, c_i,
                               c_l, c_l, c_l, c_i, c_p]),
    "gdl_adaptive_avgpool_fwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_p, c_i, c_i,
                                       c_p]),
    "gdl_adaptive_avgpool_bwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_l, c_l, c_l,
                                       c_i, c_p]),
    "gdl_v_transpose": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_p, c_i, c_p]),
    "gdl_softmax_rows": (c_i, [c_p, c_p, c_i, c_l, c_i, c_i, c_p]),
    "gdl_flash_attn_fwd": (c_i, [c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p]),
    "gdl_flash_attn_fwd2": (c_i, [c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_i, c_i, c_i, c_i, c_f, c_p]),
    "gdl_flash_attn_bwd": (c_i, [c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_p,
                                 c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_i, c_i, c_i, c_i, c_f, c_p, c_l, c_p]),
    "gdl_flash_attn_bwd_workspace": (c_l, [c_i, c_i, c_i, c_i]),
    "gdl_patchify": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p]),
    "gdl_dwconv3x3": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_p]),
    "gdl_dofa_pack_kernel": (c_i, [c_p, c_i, c_i, c_i, c_f, c_p, c_i, c_i, c_p]),
    "gdl_dofa_unpack_grad": (c_i, [c_p, c_i, c_i, c_i, c_f, c_i, c_p, c_p]),
    "gdl_sincos_embed": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p]),
    "gdl_bn_fold": (c_i, [c_p, c_p, c_p, c_p, c_f, c_i, c_p, c_p, c_p]),
    "gdl_pack_dgrad": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "gdl_cast": (c_i, [c_p, c_i, c_p, c_i, c_l, c_p]),
    "gdl_scale_f32": (c_i, [c_p, c_p, c_l, c_f, c_p]),
    "gdl_add_rows": (c_i, [c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_i, c_p]),
    "gdl_normalize_u8": (c_i, [c_p, c_p, c_i, c_i, c_l, c_p, c_p, c_p]),
    "gdl_normalize_raw": (c_i, [c_p, c_i, c_p, c_i, c_i, c_l, c_p, c_p, c_p]),
    "gdl_augment": (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "gdl_scale_outer": (c_i, [c_p, c_i, c_p, c_l, c_l, c_p]),
    "gdl_head_1x1": (c_i, [c_p, c_i, c_l, c_i, c_l, c_p, c_p, c_p, c_l, c_p, c_i, c_p]),
    "gdl_head_1x1_bwd_workspace": (c_l, [c_l, c_i, c_i]),
    "gdl_head_1x1_bwd": (c_i, [c_p, c_i, c_p, c_l, c_i, c_l, c_p, c_p, c_l, c_p, c_l, c_p, c_p, c_i,
                               c_p, c_l, c_p]),
    "gdl_upsample_logits": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p]),
    "gdl_upsample_logits_bwd_workspace": (c_l, [c_i, c_i, c_i, c_i]),
    "gdl_upsample_logits_bwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_l, c_p]),
    "gdl_softmax_argmax": (c_i, [c_p, c_i, c_i, c_l, c_p, c_p]),
    "gdl_upsample_argmax": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p]),
    "gdl_class_probs": (c_i, [c_p, c_i, c_i, c_l, c_p, c_p]),
    "gdl_iou_counts": (c_i, [c_p, c_p, c_i, c_l, c_i, c_p, c_p]),
    "gdl_dice_loss_workspace": (c_l, [c_i, c_i, c_l]),
    "gdl_dice_loss_fwd": (c_i, [c_p, c_p, c_i, c_i, c_l, c_f, c_p, c_p, c_p, c_l, c_p]),
    "gdl_dice_loss_bwd": (c_i, [c_p, c_p, c_i, c_i, c_l, c_f, c_p, c_p, c_f, c_p, c_i, c_p]),
    "gdl_dice_loss_lowres_workspace": (c_l, [c_i, c_i, c_i, c_i]),
    "gdl_dice_loss_lowres_fwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_l, c_p]),
    "gdl_dice_loss_lowres_bwd_workspace": (c_l, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "gdl_dice_loss_lowres_bwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_f, c_p, c_p, c_l, c_p]),
    "gdl_pad_nhwc": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_p, c_i, c_i, c_i, c_p]),
    "gdl_subpix4_weights": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "gdl_dice_binary_loss_fwd": (c_i, [c_p, c_p, c_l, c_f, c_p, c_p, c_p, c_l, c_p]),
    "gdl_dice_binary_loss_bwd": (c_i, [c_p, c_p, c_l, c_f, c_p, c_p, c_f, c_p, c_i, c_p]),
    "gdl_sumsq": (c_i, [c_p, c_l, c_p, c_p]),
    "gdl_clip_coef": (c_i, [c_p, c_f, c_p, c_p]),
    "gdl_multi_sumsq": (c_i, [c_p, c_i, c_p, c_p]),
    "gdl_multi_adam": (c_i, [c_p, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_p]),
    "gdl_adam_tick": (c_i, [c_p, C.c_double, C.c_double, c_p]),
    "gdl_multi_adam_dev": (c_i, [c_p, c_i, c_p, c_p, c_p]),
    "gdl_multi_repack": (c_i, [c_p, c_i, c_l, c_p]),
    "gdl_adam_step": (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_p]),
}

_lib = None


class GdlHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libgdlhip.so and bind every declared symbol; raises if the build is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        msg = (f"{LIB_PATH} not found: the HIP extension is not built. Run "
               "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
               "There is no PyTorch/CPU fallback for the gdlhip ops.")
        raise GdlHipError(msg)
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if os.environ.get("GDL_CONV_SF") is not None:      # tuning hook: 0 disables the 3x3 shared-staging kernel
        lib.gdl_debug_set_conv_sf.argtypes = [C.c_int]
        lib.gdl_debug_set_conv_sf(int(os.environ["GDL_CONV_SF"]))
    if os.environ.get("GDL_CONV_EPILOGUE") is not None:   # tuning hook: 0 = the round-2 epilogue of the 256^2 tiles
        lib.gdl_debug_set_conv_epilogue.argtypes = [C.c_int]
        lib.gdl_debug_set_conv_epilogue(int(os.environ["GDL_CONV_EPILOGUE"]))
    if os.environ.get("GDL_CONV_NGROUP_KB") is not None:  # tuning hook: weight KiB per N-tile group of the GEMM tile order (0 = round-3 order)
        lib.gdl_debug_set_conv_ngroup_kb.argtypes = [C.c_int]
        lib.gdl_debug_set_conv_ngroup_kb(int(os.environ["GDL_CONV_NGROUP_KB"]))
    if os.environ.get("GDL_CONV_DUAL") is not None:       # tuning hook: 0 = never pick the dual-resident 256 x 128 tile
        lib.gdl_debug_set_conv_dual.argtypes = [C.c_int]
        lib.gdl_debug_set_conv_dual(int(os.environ["GDL_CONV_DUAL"]))
    if os.environ.get("GDL_CONV_STAGE4") is not None:     # tuning hook: 0 = never pick the four-stage 64^2 tile (few tiles, long K)
        lib.gdl_debug_set_conv_stage4.argtypes = [C.c_int]
        lib.gdl_debug_set_conv_stage4(int(os.environ["GDL_CONV_STAGE4"]))
    if os.environ.get("GDL_CONV_W4") is not None:         # tuning hook: 0 = never pick the one-wave-per-SIMD 256^2 tile
        lib.gdl_debug_set_conv_w4.argtypes = [C.c_int]
        lib.gdl_debug_set_conv_w4(int(os.environ["GDL_CONV_W4"]))
    if os.environ.get("GDL_CONV_PERSIST") is not None:    # tuning hook: 0 = never pick the persistent 256^2 tile (dense 1x1 layers)
        lib.gdl_debug_set_conv_persist.argtypes = [C.c_int]
        lib.gdl_debug_set_conv_persist(int(os.environ["GDL_CONV_PERSIST"]))
    if os.environ.get("GDL_CONV_W4P") is not None:        # tuning hook: 0 = never pick the persistent deferred-store 256^2 tile
        lib.gdl_debug_set_conv_w4p.argtypes = [C.c_int]
        lib.gdl_debug_set_conv_w4p(int(os.environ["GDL_CONV_W4P"]))
    if os.environ.get("GDL_FLASH_FWD") is not None:       # tuning hook: 2 = the round-2 attention forward (64-query waves)
        lib.gdl_debug_set_flash_fwd.argtypes = [C.c_int, C.c_float]
        lib.gdl_debug_set_flash_fwd(int(os.environ["GDL_FLASH_FWD"]), float(os.environ.get("GDL_FLASH_DEFER", "6")))
    if os.environ.get("GDL_TAPSUM_PERSIST") is not None:  # tuning hook: 0 = one workgroup per patch in the forward gather-sum (round 3)
        lib.gdl_debug_set_tapsum_persist.argtypes = [C.c_int]
        lib.gdl_debug_set_tapsum_persist(int(os.environ["GDL_TAPSUM_PERSIST"]))
    if os.environ.get("GDL_TAPSUM_ROLL") is not None:     # tuning hook: 0 = versions 1 / 2 of the forward gather-sum where the rolling-window one applies
        lib.gdl_debug_set_tapsum_roll.argtypes = [C.c_int]
        lib.gdl_debug_set_tapsum_roll(int(os.environ["GDL_TAPSUM_ROLL"]))
    if os.environ.get("GDL_GATHER_MFMA") is not None:     # tuning hook: 0 = the VALU gathers of the low-resolution backward
        lib.gdl_debug_set_gather_mfma.argtypes = [C.c_int]
        lib.gdl_debug_set_gather_mfma(int(os.environ["GDL_GATHER_MFMA"]))
    if os.environ.get("GDL_WGRAD_XCD_GROUP") is not None:   # tuning hook: XCDs (1, 2, 4; 8 = all) the tiles of a pixel range of the 256^2 weight-gradient kernel are dealt to
        lib.gdl_debug_set_wgrad_xcd_group.argtypes = [C.c_int]
        lib.gdl_debug_set_wgrad_xcd_group(int(os.environ["GDL_WGRAD_XCD_GROUP"]))
    if os.environ.get("GDL_WGRAD_OLD_SPLITS") is not None:   # tuning hook: 1 = the round-3 split-K rule of the 256^2 weight-gradient kernel
        lib.gdl_debug_set_wgrad_old_splits.argtypes = [C.c_int]
        lib.gdl_debug_set_wgrad_old_splits(int(os.environ["GDL_WGRAD_OLD_SPLITS"]))
    if os.environ.get("GDL_WGRAD_ROWS_XCD") is not None:   # tuning hook: 1 = all tiles of a pixel range of wgrad_rows_kernel on one XCD
        lib.gdl_debug_set_wgrad_rows_xcd.argtypes = [C.c_int]
        lib.gdl_debug_set_wgrad_rows_xcd(int(os.environ["GDL_WGRAD_ROWS_XCD"]))
    if os.environ.get("GDL_HEAD_MFMA") is not None:     # tuning hook: 0 = the round-4 classifier-head kernels (wave per pixel)
        lib.gdl_debug_set_head_mfma.argtypes = [C.c_int]
        lib.gdl_debug_set_head_mfma(int(os.environ["GDL_HEAD_MFMA"]))
    if os.environ.get("GDL_WGRAD_MODE") is not None:   # tuning hook: kernel-selection bits of gdl_debug_force_wgrad_small
        lib.gdl_debug_force_wgrad_small.argtypes = [C.c_int]
        lib.gdl_debug_force_wgrad_small(int(os.environ["GDL_WGRAD_MODE"]))
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        err = load().gdl_last_error().decode(errors="replace")
        if status == -1:
            raise ValueError(f"{what}: {err}")
        raise GdlHipError(f"{what}: status {status}: {err}")
