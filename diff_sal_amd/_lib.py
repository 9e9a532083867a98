"""ctypes binding of libdiffsal_hip.so (the C ABI declared in include/diffsal.h).

The library is the product path: if it is missing or does not load, importing the ops fails
loudly -- there is no PyTorch/CPU fallback.
"""
import ctypes as C
import os

from . import build as _build

_LIB = None

c_f = C.c_void_p  # device pointers travel as integers
c_i = C.c_int
c_fl = C.c_float
c_sz = C.c_size_t


class ConvDesc(C.Structure):
    """struct diffsal_conv_desc (include/diffsal.h)."""

    _fields_ = [(n, C.c_int) for n in (
        "N", "H", "W", "Cin", "Ho", "Wo", "Cout", "KH", "KW", "stride_h", "stride_w", "pad_t", "pad_l",
        "dil_h", "dil_w", "act", "rowvec_ld", "w_format", "precision", "dtype")]


class Wino4Ext(C.Structure):
    """struct diffsal_wino4_ext (include/diffsal.h): optional extras of diffsal_conv_wino4_ex."""

    _fields_ = [("in_ab", C.c_void_p), ("side_a", C.c_void_p), ("side_w", C.c_void_p), ("side_out", C.c_void_p),
                ("out_stats", C.c_void_p), ("up2_c", C.c_void_p), ("up2_scale", C.c_void_p), ("up2_shift", C.c_void_p),
                ("side_rows", C.c_longlong), ("in_swish", C.c_int), ("out_groups", C.c_int), ("up2_act", C.c_int), ("reserved", C.c_int)]


# must equal diffsal_version() of the loaded binary: bumped whenever a signature or struct in include/diffsal.h changes,
# so that a stale libdiffsal_hip.so is rejected instead of being called with the wrong argument lists
ABI_VERSION = 42


SIGNATURES = {
    "diffsal_version": (c_i, []),
    "diffsal_last_error": (C.c_char_p, []),
    "diffsal_last_gemm_kernel": (C.c_char_p, []),
    "diffsal_conv_igemm_group": (c_i, [c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_f]),
    "diffsal_kv_prep_proj": (c_i, [c_f] * 14 + [c_i, c_i, c_i, c_i, c_i, c_fl, c_f, c_f, c_fl, c_i, c_i, c_f]),
    "diffsal_block_front": (c_i, [c_f, c_f, c_f, c_f, c_f, c_fl, c_f, c_f, c_f, c_fl, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i,
                                  c_i, c_i, c_fl, c_i, c_f]),
    "diffsal_set_tuning": (c_i, [C.c_char_p, c_i]),
    "diffsal_get_tuning": (c_i, [C.c_char_p]),
    "diffsal_temb_mlp": (c_i, [c_f, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "diffsal_dense_small": (c_i, [c_f, c_i, c_i, c_i, c_f, c_f, c_i, c_f, c_f]),
    "diffsal_conv_in": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_maxpool2d": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_conv_in_s4": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_tapsum": (c_i, [c_f, c_f, c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_i, c_i, c_f]),
    "diffsal_tapsum_head": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_f]),
    "diffsal_tapsum_bwd_ws_bytes": (C.c_long, [c_i, c_i, c_i, c_i]),
    "diffsal_tapsum_bwd": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_groupnorm_ws_bytes": (c_sz, [c_i, c_i]),
    "diffsal_groupnorm_swish": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_fl, c_f, c_sz, c_i, c_f]),
    "diffsal_groupnorm": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_fl, c_i, c_f, c_sz, c_i, c_f]),
    "diffsal_softmax_rows": (c_i, [c_f, c_f, C.c_long, c_i, c_fl, c_f]),
    "diffsal_upsample_nearest2": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_avgpool2": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_sigmoid_gate": (c_i, [c_f, c_f, c_f, C.c_long, c_f]),
    "diffsal_conv_igemm_ws_bytes": (c_sz, [C.POINTER(ConvDesc)]),
    "diffsal_conv_igemm": (c_i, [C.POINTER(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_f]),
    "diffsal_linear_f32out": (c_i, [C.POINTER(ConvDesc), c_f, c_f, c_f, c_f, c_f]),
    "diffsal_linear_pair": (c_i, [C.POINTER(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_f]),
    "diffsal_conv_wgrad_ws_bytes": (c_sz, [C.POINTER(ConvDesc)]),
    "diffsal_conv_wgrad_splits": (c_i, [C.POINTER(ConvDesc)]),
    "diffsal_pack_weight_many": (c_i, [c_f, c_i, c_i, c_i, c_f]),
    "diffsal_conv_wgrad": (c_i, [C.POINTER(ConvDesc), c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_f]),
    "diffsal_colsum": (c_i, [c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    "diffsal_act_bwd": (c_i, [c_f, c_f, c_f, c_sz, c_i, c_f]),
    "diffsal_rowstats_chunks": (c_i, [c_i, c_i]),
    "diffsal_rowstats": (c_i, [c_f] * 8 + [c_i, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_affine_act": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_norm_bwd_apply": (c_i, [c_f] * 11 + [c_i, c_i, c_i, c_i, c_f]),
    "diffsal_layernorm_bwd_blocks": (c_i, [c_i, c_i]),
    "diffsal_wino4_weight": (c_i, [c_f, c_f, c_i, c_i, c_i, c_f]),
    "diffsal_layernorm_multi": (c_i, [c_f] * 4 + [C.POINTER(c_i), c_i, c_i, C.POINTER(c_fl), c_f]),
    "diffsal_layernorm_bwd_multi": (c_i, [c_f] * 5 + [C.POINTER(c_i), c_i, c_i, C.POINTER(c_fl), c_f]),
    "diffsal_layernorm_bwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_fl, c_f]),
    "diffsal_dropout": (c_i, [c_f, c_f, c_sz, c_fl, C.c_uint64, c_f]),
    "diffsal_dwconv": (c_i, [c_f, c_f, c_f] + [c_i] * 7 + [c_f]),
    "diffsal_dwconv_bwd_data": (c_i, [c_f, c_f, c_f] + [c_i] * 7 + [c_f]),
    "diffsal_dwconv_bwd_weight_chunks": (c_i, [c_i] * 6),
    "diffsal_dwconv_bwd_weight": (c_i, [c_f, c_f, c_f] + [c_i] * 7 + [c_f]),
    "diffsal_attention_bwd_blocks": (c_i, [c_i, c_i, c_i]),
    "diffsal_attention_bwd": (c_i, [c_f] * 7 + [c_i] * 6 + [c_fl, c_f]),
    "diffsal_wgrad_segmented_ws_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "diffsal_wgrad_segmented": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_sz, c_f]),
    "diffsal_resize_bilinear_bwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_sz, c_f]),
    "diffsal_unpack_frames": (c_i, [c_f, c_f] + [c_i] * 5 + [c_f]),
    "diffsal_head_bwd": (c_i, [c_f] * 6 + [c_i, c_i, c_i, c_f]),
    "diffsal_conv_in_bwd": (c_i, [c_f, c_f, c_f] + [c_i] * 5 + [c_f]),
    "diffsal_dense_small_bwd": (c_i, [c_f] * 6 + [c_i] * 4 + [c_f]),
    "diffsal_audio_fuse_bwd": (c_i, [c_f] * 5 + [c_i] * 7 + [c_f]),
    "diffsal_pack_frames": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_pack_frames_multi": (c_i, [c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_f, c_i, c_f]),
    "diffsal_resize_bilinear": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_resize_sum": (c_i, [C.POINTER(C.c_void_p), C.POINTER(c_i), C.POINTER(c_i), c_i, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_audio_fuse": (c_i, [c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_layernorm": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_fl, c_i, c_f]),
    "diffsal_dwconv3_ln": (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_fl, c_i, c_f]),
    "diffsal_dwpool_ln_kv": (c_i, [c_f] * 10 + [c_i, c_i, c_i, c_i, c_i, c_fl, c_i, c_f]),
    "diffsal_qkv_prep": (c_i, [c_f] * 15 + [c_i, c_i, c_i, c_i, c_i, c_fl, c_f, c_f, c_fl, c_i, c_i, c_f]),
    "diffsal_mlp_block": (c_i, [c_f, c_f, c_f, c_fl, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_fl, C.c_long, c_i, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_block16": (c_i, [c_f] * 6 + [c_fl] + [c_f] * 8 + [c_fl, C.c_long] + [c_i] * 6 + [c_f]),
    "diffsal_attention": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_fl, c_i, c_f]),
    "diffsal_head_sigmoid": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    "diffsal_cast": (c_i, [c_f, c_i, c_f, c_i, C.c_long, c_f]),
    "diffsal_axpbypcz": (c_i, [c_f, c_f, c_f, c_fl, c_fl, c_fl, c_f, c_sz, c_f]),
    "diffsal_attention_general": (c_i, [c_f] * 4 + [c_f] + [c_f] * 4 + [c_i] * 7 + [C.POINTER(C.c_long)] * 4 + [c_fl, c_i, c_f, c_sz, c_f]),
    "diffsal_attention_general_tail_floats": (c_sz, [c_i] * 5),
    "diffsal_attention_general_bwd_splits": (c_i, [c_i] * 4),
    "diffsal_attention_general_bwd_qtail_floats": (c_sz, [c_i] * 6),
    "diffsal_attention_general_bwd_ds_floats": (c_sz, [c_i] * 4),
    "diffsal_attention_general_bwd": (c_i, [c_f] * 12 + [c_sz] + [c_f, c_sz] + [c_f] * 4 + [c_i] * 7 + [C.POINTER(C.c_long)] * 4 + [c_fl, c_i, c_f]),
    "diffsal_im2col3d": (c_i, [c_f, c_f] + [c_i] * 15 + [c_f]),
    "diffsal_pool3d_ln": (c_i, [c_f] * 5 + [c_i] * 9 + [C.c_long, C.c_long, c_fl, c_f]),
    "diffsal_maxpool_tokens": (c_i, [c_f, c_f] + [c_i] * 11 + [c_f]),
    "diffsal_relpos_project": (c_i, [c_f] * 5 + [c_i] * 9 + [c_f]),
    "diffsal_tokens_to_channels_first": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_pool3d_bwd_data": (c_i, [c_f] * 3 + [c_i] * 9 + [C.c_long, C.c_long, c_f]),
    "diffsal_pool3d_bwd_weight_chunks": (c_i, []),
    "diffsal_pool3d_bwd_weight": (c_i, [c_f] * 3 + [c_i] * 9 + [C.c_long, C.c_long, c_f]),
    "diffsal_maxpool_tokens_idx": (c_i, [c_f] * 3 + [c_i] * 11 + [c_f]),
    "diffsal_maxpool_tokens_bwd": (c_i, [c_f] * 3 + [c_i] * 11 + [c_f]),
    "diffsal_qkv_pool": (c_i, [c_f] * 6 + [c_i] * 6 + [c_f] * 2 + [c_i, c_i, c_f]),
    "diffsal_qkv_pool_bwd_data": (c_i, [c_f] * 3 + [c_i] * 6 + [c_f] * 2 + [c_i, c_f]),
    "diffsal_conv_wino_supported": (c_i, [C.POINTER(ConvDesc)]),
    "diffsal_conv_wino_ws_bytes": (c_sz, [C.POINTER(ConvDesc)]),
    "diffsal_conv_wino": (c_i, [C.POINTER(ConvDesc)] + [c_f] * 9 + [c_sz, c_f]),
    "diffsal_conv_wino4_supported": (c_i, [C.POINTER(ConvDesc)]),
    "diffsal_conv_wino4_ws_bytes": (c_sz, [C.POINTER(ConvDesc)]),
    "diffsal_conv_wino4": (c_i, [C.POINTER(ConvDesc)] + [c_f] * 9 + [c_sz, c_f]),
    "diffsal_conv_wino4_stages": (c_i, [C.POINTER(ConvDesc)] + [c_f] * 9 + [c_sz, c_i, c_f]),
    "diffsal_conv_wino4_ex": (c_i, [C.POINTER(ConvDesc)] + [c_f] * 9 + [c_sz, C.POINTER(Wino4Ext), c_i, c_f]),
    "diffsal_conv_wino4_stats_bytes": (c_sz, [C.POINTER(ConvDesc), c_i]),
    "diffsal_conv_wino4_side_supported": (c_i, [C.POINTER(ConvDesc), C.c_long]),
    "diffsal_gn_affine": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_fl, c_f, c_sz, c_i, c_f]),
    "diffsal_gn_affine_wino4": (c_i, [C.POINTER(ConvDesc), c_f, c_f, c_f, c_i, c_fl, c_f, c_f]),
    "diffsal_workspace_bytes": (c_sz, [c_i, C.POINTER(ConvDesc), C.POINTER(C.c_long), c_i]),
    "diffsal_border_gather": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_up2_conv_commute": (c_i, [c_f] * 5 + [c_i] * 6 + [c_f]),
    "diffsal_up2_conv_commute_ring": (c_i, [c_f] * 5 + [c_i] * 6 + [c_f]),
    "diffsal_rel_tables": (c_i, [c_f] * 5 + [c_i, c_f]),
    "diffsal_rel_tables_bwd": (c_i, [c_f] * 6 + [c_i, c_f]),
    "diffsal_qkv_pool_bwd_weight_chunks": (c_i, [c_i] * 5 + [C.POINTER(c_i)]),
    "diffsal_qkv_pool_bwd_weight": (c_i, [c_f] * 3 + [c_i] * 6 + [c_f] * 2 + [c_i, c_f]),
    "diffsal_relpos_project_bwd_chunks": (c_i, []),
    "diffsal_relpos_project_bwd": (c_i, [c_f] * 6 + [c_i] + [c_f] + [c_i] * 9 + [c_f]),
    "diffsal_resize_update": (c_i, [c_f] * 6 + [c_i] * 5 + [c_fl] * 5 + [c_f]),
    "diffsal_saliency_metrics_ws_bytes": (c_sz, [c_i]),
    "diffsal_saliency_metrics_bwd": (c_i, [c_f, c_f, c_i, C.c_long, c_f, c_sz, c_f, c_f, c_f]),
    "diffsal_saliency_metrics": (c_i, [c_f, c_f, c_i, C.c_long, c_f, c_sz, c_f, c_f, c_f]),
    "diffsal_reduce_partials": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_norm_finalize_fwd": (c_i, [c_f] * 7 + [c_i, c_i, c_i, C.c_double, C.c_double] + [c_f] * 4 + [c_fl, c_fl, c_f]),
    "diffsal_norm_finalize_bwd": (c_i, [c_f] * 8 + [c_i, c_i, c_i, C.c_double, c_f]),
    "diffsal_pack_weight": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "diffsal_split_weight": (c_i, [c_f, c_f, C.c_long, c_f]),
    "diffsal_col2im_disjoint": (c_i, [c_f, c_f] + [c_i] * 12 + [c_f]),
    "diffsal_col2im_gather": (c_i, [c_f, c_f] + [c_i] * 12 + [c_f]),
    "diffsal_reduce_blocks": (c_i, []),
    "diffsal_multi_copy": (c_i, [C.POINTER(C.c_void_p), C.POINTER(C.c_long), C.POINTER(C.c_long), c_i, c_f, c_f]),
    "diffsal_scale_by": (c_i, [c_f, c_f, c_f, C.c_long, c_f]),
    "diffsal_mse_loss": (c_i, [c_f, c_f, c_f, c_f, c_f, C.c_long, c_fl, c_f]),
    "diffsal_grad_norm": (c_i, [c_f, C.c_long, c_fl, c_f, c_f, c_f]),
    "diffsal_adam_step": (c_i, [c_f] * 4 + [C.c_long] + [C.c_double] * 5 + [c_i, c_fl, c_f, c_fl, c_i, c_f]),
}

(WS_GROUPNORM, WS_CONV_IGEMM, WS_CONV_WINO, WS_CONV_WINO4, WS_CONV_WINO4_STATS, WS_CONV_WGRAD, WS_WGRAD_SEGMENTED, WS_TAPSUM_BWD,
 WS_SALIENCY_METRICS, WS_ATTENTION_TAIL, WS_ATTENTION_BWD_QTAIL, WS_ATTENTION_BWD_DS) = range(12)      # enum DIFFSAL_WS_* of include/diffsal.h


def workspace_bytes(op: int, desc=None, dims=()):
    """``diffsal_workspace_bytes``: scratch bytes of operator ``op`` (WS_*) for a descriptor and / or a tuple of integers."""
    arr = (C.c_long * len(dims))(*[int(v) for v in dims]) if dims else None
    return load().diffsal_workspace_bytes(op, C.byref(desc) if desc is not None else None, arr, len(dims))


ACT_NONE, ACT_RELU, ACT_GELU, ACT_SIGMOID = 0, 1, 2, 3
ACT_GELU_GRAD = 5      # training only: product * gelu'(residual) (include/diffsal.h)
PREC_FP32, PREC_BF16X3 = 0, 1
F32, BF16, F16 = 0, 1, 2


def library_path():
    return _build.LIB


def load():
    """dlopen the HIP library (building it first if sources are newer); raises if impossible."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB
    if not os.path.exists(path) or (_build.needs_build() and os.environ.get("DIFFSAL_NO_REBUILD") != "1"):
        try:
            _build.build_library()
        except Exception as e:  # noqa: BLE001
            # never bind the current signatures to a binary older than its sources: a changed argument list would be
            # silent memory corruption on the GPU.  DIFFSAL_NO_REBUILD=1 opts out (the ABI version is still checked).
            raise RuntimeError(
                f"libdiffsal_hip.so is {'stale' if os.path.exists(path) else 'missing'} and could not be built ({e}); "
                "run `python -m diff_sal_amd.build` (or set DIFFSAL_NO_REBUILD=1 to load the existing binary as is)"
            ) from e
    # PyTorch bundles its own libamdhip64 (same soname as /opt/rocm's).  Import it first so that ONE HIP
    # runtime lives in the process and streams / device pointers are shared with torch.
    import torch  # noqa: F401

    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = lib.diffsal_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"{path} reports ABI version {got}, the Python binding expects {ABI_VERSION}: "
                           "rebuild with `python -m diff_sal_amd.build --force`")
    _LIB = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().diffsal_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libdiffsal_hip {what} failed (code {rc}): {msg}")


TUNING_EPOCH = [0]      # bumped by set_tuning: host-side caches of the library's planner answers key on it


def set_tuning(name: str, value) -> None:
    """Set a test / tuning switch of the library (include/diffsal.h: diffsal_set_tuning); None or a negative value unsets it."""
    check(load().diffsal_set_tuning(name.encode(), -1 if value is None else int(value)), "set_tuning")
    TUNING_EPOCH[0] += 1


def get_tuning(name: str) -> int:
    return load().diffsal_get_tuning(name.encode())
