"""ctypes binding of libmcamd.so (include/mcamd.h).

The product path has no CPU or PyTorch fallback: if the HIP library cannot be
loaded this raises, and every wrapper raises `McamdError` on a non-zero return.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmcamd.so")


class McamdError(RuntimeError):
    pass


class ConvGeom(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("ksize", C.c_int32),
                ("cin", C.c_int32), ("cout", C.c_int32), ("x_ld", C.c_int32), ("x_choff", C.c_int32),
                ("stem", C.c_int32), ("pad", C.c_int32), ("x_wrap", C.c_int32), ("x_f8", C.c_int32), ("x_f8_wexp", C.c_int32)]


class ConvEpilogue(C.Structure):
    _fields_ = [("mode", C.c_int32), ("y_ld", C.c_int32), ("y_choff", C.c_int32),
                ("y", C.c_void_p), ("bias", C.c_void_p), ("stats", C.c_void_p),
                ("stats_rows", C.c_int32), ("stats_ld", C.c_int32),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("slope", C.c_float), ("overflow", C.c_void_p),
                ("dst_mode", C.c_int32), ("y2", C.c_void_p), ("y2_ld", C.c_int32), ("y2_choff", C.c_int32),
                ("concurrent", C.c_int32)]


class ActDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("y", C.c_void_p), ("y_ld", C.c_int32), ("y_choff", C.c_int32),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("slope", C.c_float), ("mode", C.c_int32),
                ("dst", C.c_void_p), ("dst_ld", C.c_int32), ("dst_choff", C.c_int32),
                ("dst2", C.c_void_p), ("dst2_ld", C.c_int32), ("dst2_choff", C.c_int32),
                ("y_dtype", C.c_int32), ("planes", C.c_int32), ("dst_plane", C.c_int32), ("dst2_plane", C.c_int32),
                ("dst_pad", C.c_int32), ("dst2_pad", C.c_int32), ("border", C.c_void_p), ("planes2", C.c_int32),
                ("pool_act", C.c_void_p), ("pool_act_ld", C.c_int32), ("pool_act_pad", C.c_int32)]


class ChanMap(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("cols", C.c_void_p)]


class PackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("mask", C.c_void_p), ("dst_fwd", C.c_void_p), ("dst_dgrad", C.c_void_p),
                ("rows", C.c_void_p), ("cols", C.c_void_p),
                ("first_tile", C.c_int64), ("cout", C.c_int32), ("cin", C.c_int32), ("ksize", C.c_int32),
                ("split", C.c_int32), ("f8_wexp", C.c_int32)]


class ActBwdDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("y", C.c_void_p), ("y_ld", C.c_int32), ("y_choff", C.c_int32),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p),
                ("slope", C.c_float), ("mode", C.c_int32),
                ("g", C.c_void_p), ("g_ld", C.c_int32), ("g_choff", C.c_int32),
                ("g2", C.c_void_p), ("g2_ld", C.c_int32), ("g2_choff", C.c_int32),
                ("dy", C.c_void_p), ("dy_ld", C.c_int32), ("dy_choff", C.c_int32),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("grad_scale", C.c_float),
                ("dy_keep", C.c_void_p), ("chan_perm", C.c_void_p), ("y_dtype", C.c_int32),
                ("overflow", C.c_void_p), ("dy_pad", C.c_int32), ("skip_dead_param_grads", C.c_int32),
                ("act", C.c_void_p), ("act_ld", C.c_int32), ("act_choff", C.c_int32), ("act_pad", C.c_int32)]


class FoldDesc(C.Structure):
    _fields_ = [("w", C.c_void_p), ("mask", C.c_void_p), ("rows", C.c_void_p), ("cols", C.c_void_p),
                ("beta", C.c_void_p), ("slope", C.c_float),
                ("n", C.c_int32), ("cin_t", C.c_int32), ("cin_k", C.c_int32), ("cin_aug", C.c_int32), ("ksize", C.c_int32)]


class StemBlockDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("x", C.c_void_p), ("wp", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("momentum", C.c_float), ("eps", C.c_float), ("training", C.c_int32),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("save_mean", C.c_void_p), ("save_invstd", C.c_void_p),
                ("slope", C.c_float),
                ("dst", C.c_void_p), ("dst_ld", C.c_int32), ("dst_choff", C.c_int32),
                ("g", C.c_void_p), ("g_ld", C.c_int32), ("g_choff", C.c_int32),
                ("mask", C.c_void_p), ("grad_scale", C.c_float),
                ("dw", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("cout", C.c_int32), ("planes", C.c_int32),
                ("x_lo", C.c_void_p), ("wp_lo", C.c_void_p)]


class RegionDesc(C.Structure):
    _fields_ = [("output", C.c_void_p), ("target", C.c_void_p),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("num_anchors", C.c_int32), ("num_classes", C.c_int32),
                ("max_boxes", C.c_int32), ("anchors", C.c_float * 16),
                ("coord_scale", C.c_float), ("noobject_scale", C.c_float), ("object_scale", C.c_float),
                ("class_scale", C.c_float), ("thresh", C.c_float)]


class FoldJob(C.Structure):
    _fields_ = [("d", FoldDesc), ("waug", C.c_void_p), ("first_block", C.c_int64)]


EPI_RAW_F16, EPI_NCHW_F32, EPI_PAD_F16, EPI_RAW_F32 = 0, 1, 2, 3
DST_PLAIN, DST_POOL, DST_REORG = 0, 1, 2

# name -> (restype, argtypes); the complete list of symbols include/mcamd.h declares.
_P, _I32, _I64, _F, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
SIGNATURES = {
    "mcamd_version": (C.c_int, []),
    "mcamd_arch": (C.c_char_p, []),
    "mcamd_last_error": (C.c_char_p, []),
    "mcamd_reload_config": (None, []),
    "mcamd_conv_stats_rows": (_I32, [C.POINTER(ConvGeom)]),
    "mcamd_conv_stats_rows_mode": (_I32, [C.POINTER(ConvGeom), _I32]),
    "mcamd_conv_fwd_f8_ok": (_I32, [C.POINTER(ConvGeom)]),
    "mcamd_conv_tile_info": (C.c_int, [C.POINTER(ConvGeom), _I32, C.POINTER(_I32)]),
    "mcamd_packed_elems_fwd": (_I64, [C.POINTER(ConvGeom)]),
    "mcamd_packed_elems_dgrad": (_I64, [C.POINTER(ConvGeom)]),
    "mcamd_pack_weights": (C.c_int, [C.POINTER(ConvGeom), _P, _P, C.POINTER(ChanMap), _P, _P, _P]),
    "mcamd_pack_weights_many": (C.c_int, [_P, _I32, _I64, _P]),
    "mcamd_conv_fwd": (C.c_int, [C.POINTER(ConvGeom), _P, _P, C.POINTER(ConvEpilogue), _P]),
    "mcamd_conv_dgrad": (C.c_int, [C.POINTER(ConvGeom), _P, _I32, _I32, _P, C.POINTER(ConvEpilogue), _P]),
    "mcamd_conv_wgrad_workspace_bytes": (_SZ, [C.POINTER(ConvGeom)]),
    "mcamd_conv_wgrad": (C.c_int, [C.POINTER(ConvGeom), _P, _P, _I32, _I32, _P, C.POINTER(ChanMap), _F, _P, _P, _P, _SZ, _P]),
    "mcamd_bn_coeffs": (C.c_int, [_P, _I32, _I32, _I32, _I64, _P, _P, _P, _P, _F, _F, _I32, _P, _P, _P, _P, _P, _P]),
    "mcamd_bn_coeffs_ex": (C.c_int, [_P, _I32, _I32, _I32, _I64, _P, _P, _P, _P, _F, _F, _I32, _P, _P, _P, _P, _P, _I32, _P]),
    "mcamd_fold_weights": (C.c_int, [C.POINTER(FoldDesc), _P, _P]),
    "mcamd_fold_weights_many": (C.c_int, [_P, _I32, _I64, _P]),
    "mcamd_unfold_wgrad": (C.c_int, [C.POINTER(FoldDesc), _P, _P, _P, _P, _I32, _P]),
    "mcamd_bn_act_fwd": (C.c_int, [C.POINTER(ActDesc), _P]),
    "mcamd_bn_act_bwd_workspace_bytes": (_SZ, [C.POINTER(ActBwdDesc)]),
    "mcamd_bn_act_bwd": (C.c_int, [C.POINTER(ActBwdDesc), _P, _SZ, _P]),
    "mcamd_stem_block_workspace_bytes": (_SZ, []),
    "mcamd_stem_block_fwd": (C.c_int, [C.POINTER(StemBlockDesc), _P, _SZ, _P]),
    "mcamd_stem_block_bwd": (C.c_int, [C.POINTER(StemBlockDesc), _P, _SZ, _P]),
    "mcamd_stem_block_stats_rows": (_I32, [C.POINTER(StemBlockDesc)]),
    "mcamd_stem_block_stats": (C.c_int, [C.POINTER(StemBlockDesc), _P, _I32, _I32, _P]),
    "mcamd_nchw_f32_to_nhwc4_split": (C.c_int, [_P, _I32, _I32, _I32, _P, _P, _P]),
    "mcamd_pack_stem_split": (C.c_int, [_P, _P, _I32, _P, _P, _P]),
    "mcamd_nchw_f32_to_padded_nhwc_f16": (C.c_int, [_P, _I32, _I32, _I32, _I32, _F, _P, _I32, _I32, _P, _P]),
    "mcamd_nchw_f32_to_padded_nhwc_f16_pad": (C.c_int, [_P, _I32, _I32, _I32, _I32, _F, _P, _I32, _I32, _I32, _P, _P]),
    "mcamd_nchw_f32_to_padded_nhwc_f16_split": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _I32, _I32, _I32, _P]),
    "mcamd_stem_conv_f32_stats_rows": (_I32, []),
    "mcamd_stem_conv_f32": (C.c_int, [_P, _I32, _I32, _I32, _P, _P, _I32, _P, _P, _I32, _P, _I32, _I32, _P]),
    "mcamd_region_loss_workspace_bytes": (_SZ, [_I32]),
    "mcamd_region_loss": (C.c_int, [C.POINTER(RegionDesc), _P, _P, _P, _P, _SZ, _P]),
    "mcamd_plan_begin": (C.c_int, [C.POINTER(_P), _I32]),
    "mcamd_plan_mark": (_I32, []),
    "mcamd_plan_end": (_P, []),
    "mcamd_plan_segments": (_I32, [_P]),
    "mcamd_plan_launches": (_I32, [_P]),
    "mcamd_plan_run": (C.c_int, [_P, _I32, _I32, C.POINTER(_P), _I32]),
    "mcamd_plan_destroy": (None, [_P]),
    "mcamd_stream_wait": (C.c_int, [_P, _P]),
    "mcamd_memset_zero": (C.c_int, [_P, _SZ, _P]),
    "mcamd_step_flags": (C.c_int, [_P, _I32, _P, _P, _P, _P, _P]),
    "mcamd_kth_magnitude_workspace_bytes": (_SZ, []),
    "mcamd_kth_magnitude": (C.c_int, [C.POINTER(_P), C.POINTER(_I64), _I32, _I64, _P, _P, _SZ, _P]),
    "mcamd_magnitude_mask": (C.c_int, [_P, _I64, _P, _P, _P]),
    "mcamd_filter_scores_workspace_bytes": (_SZ, [_I32]),
    "mcamd_filter_scores": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P, _SZ, _P]),
    "mcamd_filter_mean_square": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P, _SZ, _P]),
    "mcamd_filter_mask": (C.c_int, [_P, _I32, _I64, _P, _P]),
    "mcamd_count_zeros": (C.c_int, [_P, _I64, _P, _P]),
    "mcamd_masked_residual": (C.c_int, [_P, _P, _I64, _P, _P]),
}

_lib = None


def lib():
    """The loaded library; raises McamdError when it is absent (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise McamdError(
                "libmcamd.so is missing (%s). Build it with `python -m modelcompression_amd.build`; "
                "modelcompression_amd has no CPU/PyTorch fallback for its compute path." % LIB_PATH)
        # torch first: PyTorch-ROCm ships its own libamdhip64 and must be the HIP runtime of this process.  Loaded
        # the other way round, libmcamd.so binds /opt/rocm's copy and its launches go to a second runtime that
        # never saw torch's device / streams ("no ROCm-capable device is detected").
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)          # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def reload_config():
    """Have the library re-read its MCAMD_* switches (they are cached at first use; tests that change one call this)."""
    lib().mcamd_reload_config()


def check(rc, what=""):
    if rc != 0:
        msg = lib().mcamd_last_error().decode()
        raise McamdError("%s failed (%d): %s" % (what or "mcamd call", rc, msg))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
