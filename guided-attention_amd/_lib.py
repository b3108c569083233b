"""ctypes binding of libga_hip.so (declared in include/ga_hip.h).

There is no CPU fallback: if the library is missing or a kernel reports an error the call
raises.  Tensors are passed as raw device pointers (`tensor.data_ptr()`), work is enqueued on
torch's current HIP stream, nothing synchronises.
"""
import ctypes
import os
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("GA_HIP_LIB", _HERE / "libga_hip.so"))

GA_VERSION = 180   # the GA_VERSION of include/ga_hip.h these prototypes were written for (tests/test_abi.py compares the two)
GA_F16, GA_BF16, GA_F32 = 0, 1, 2
GA_LINEAR_STREAM = 8   # `stages` of ga_linear_fused: the persistent one-workgroup-per-CU form (include/ga_hip.h)
GA_TOK_COOR, GA_TOK_BOX = 0, 1
GA_TERMS = 8
DTYPE_CODE = {torch.float16: GA_F16, torch.bfloat16: GA_BF16, torch.float32: GA_F32}


class GaError(RuntimeError):
    pass


class ga_token_t(ctypes.Structure):
    _fields_ = [("token", ctypes.c_int32), ("kind", ctypes.c_int32), ("geom", ctypes.c_double * 4),
                ("weight", ctypes.c_float), ("_pad", ctypes.c_float)]


class ga_loss_params_t(ctypes.Structure):
    _fields_ = [("inside_scale", ctypes.c_float), ("outside_scale", ctypes.c_float), ("center_weight", ctypes.c_float),
                ("sigma", ctypes.c_float), ("shrink", ctypes.c_double), ("ksize", ctypes.c_int32),
                ("smooth", ctypes.c_int32), ("strict", ctypes.c_int32), ("_pad", ctypes.c_int32)]


class ga_linear_epilogue_t(ctypes.Structure):
    _fields_ = [("bias", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("ld_res", ctypes.c_int64),
                ("geglu", ctypes.c_int32), ("preact", ctypes.c_void_p), ("ld_pre", ctypes.c_int64),
                ("ln_partials", ctypes.c_void_p), ("ln_parts", ctypes.c_int32), ("ln_eps", ctypes.c_float),
                ("ln_colsum", ctypes.c_void_p), ("ln_shift", ctypes.c_void_p), ("ln_stats_out", ctypes.c_void_p),
                ("row_partials_out", ctypes.c_void_p), ("gn_partials", ctypes.c_void_p), ("gn_groups", ctypes.c_int32),
                ("gn_hw", ctypes.c_int32)]


_vp, _i, _f, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64

# name -> argtypes, exactly the prototypes of include/ga_hip.h
PROTOTYPES = {
    "ga_version": [],
    "ga_strerror": [_i],
    "ga_attn_capture_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "ga_attn_capture_bwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "ga_attn_scores_max": [_vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp],
    "ga_attn_capture_fwd_biased": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "ga_attn_capture_bwd_biased": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "ga_aggregate_maps": [ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _i, _i, _vp, _i, _vp],
    "ga_smooth_loss_fwd": [_vp, _i, _i, _i, _i, ctypes.POINTER(ga_token_t), _i, ctypes.POINTER(ga_loss_params_t), _vp,
                           _vp, _vp],
    "ga_smooth_loss_bwd": [_vp, _i, _i, _i, _i, ctypes.POINTER(ga_token_t), _i, ctypes.POINTER(ga_loss_params_t), _vp,
                           _vp, _vp, _f, _i, _vp],
    "ga_aggregate_loss_fwd": [ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _i, _i, _i, _i, ctypes.POINTER(ga_token_t), _i,
                              ctypes.POINTER(ga_loss_params_t), _vp, _vp, _vp, _vp, _i, _vp],
    "ga_gaussian_weights": [_i, _f, ctypes.POINTER(_f)],
    "ga_latent_axpy": [_vp, _vp, _f, _vp, _vp, _i64, _i, _vp],
    "ga_latent_axpby": [_vp, _vp, _f, _f, _vp, _i64, _i, _vp],
    "ga_cfg_ddim_step": [_vp, _vp, _f, _vp, _f, _f, _vp, _vp, _i64, _i, _vp],
    "ga_self_attn_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "ga_self_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "ga_group_norm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp],
    "ga_group_norm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "ga_geglu_fwd": [_vp, _vp, _i64, _i, _i, _vp],
    "ga_geglu_bwd": [_vp, _vp, _vp, _i64, _i, _i, _vp],
    "ga_bias_residual_add": [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "ga_cat_channels": [_vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "ga_cat_channels_gn_blocks": [_i, _i, _i, _i],
    "ga_linear_gn_blocks": [_i, _i, _i, _i, _i],
    "ga_group_norm_one_launch": [_i, _i, _i, _i],
    "ga_cat_group_norm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp],
    "ga_cat_channels_gn": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "ga_conv3x3_packed_elems": [_i, _i],
    "ga_conv3x3_thin_packed_elems": [_i, _i],
    "ga_conv3x3_thin_supported": [_i, _i, _i, _i],
    "ga_conv3x3_thin_pack": [_vp, _vp, _i, _i, _i64, _i64, _i64, _i64, _i, _i, _vp],
    "ga_conv3x3_thin_in": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ga_conv3x3_thin_out": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ga_conv3x3_pack_weights": [_vp, _vp, _i, _i, _i64, _i64, _i64, _i64, _i, _i, _vp],
    "ga_conv3x3_plan": [_i, _i, _i, _i, _i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i),
                        ctypes.POINTER(ctypes.c_longlong)],
    "ga_splitk_workspace_floats": [_i64, _i, _i, _i, _i],
    "ga_conv3x3_nhwc": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ga_conv3x3_gn_blocks": [_i, _i, _i, _i, _i, _i],
    "ga_conv3x3_nhwc_gn": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i],
    "ga_group_norm_apply": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp],
    "ga_group_norm_two_launch": [_i, _i, _i, _i],
    "ga_conv3x3_up2x_nhwc": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ga_gemm_nt": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp],
    "ga_linear_workspace": [_i64, _i, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(_i)],
    "ga_linear_fused": [_vp, _i64, _vp, _vp, _i64, ctypes.POINTER(ga_linear_epilogue_t), _vp, _vp, _i64, _i, _i, _i, _i, _i,
                        _i, _i, _vp],
    "ga_add_layer_norm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _i, _vp],
    "ga_add_layer_norm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
}

_lib = None


def load():
    """Load libga_hip.so once; raise GaError (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise GaError(f"{LIB_PATH} not found: build it with `make` (or __graft_entry__.build()); "
                          "the guided-attention path has no CPU fallback")
        lib = ctypes.CDLL(str(LIB_PATH))
        lib.ga_version.argtypes, lib.ga_version.restype = [], ctypes.c_int
        built = lib.ga_version()
        if built != GA_VERSION:
            # signatures changed between versions by pointers inserted in the MIDDLE of argument lists: a mismatched pair would
            # hand e.g. the bias where the ticket array is expected — refuse before the first call
            raise GaError(f"{LIB_PATH} is ABI version {built}, this binding is written for {GA_VERSION}: rebuild with `make`")
        for name, argtypes in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = (ctypes.c_char_p if name == "ga_strerror" else
                          ctypes.c_longlong if name in ("ga_splitk_workspace_floats", "ga_conv3x3_packed_elems", "ga_conv3x3_thin_packed_elems") else ctypes.c_int)
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise GaError(f"{what} failed: {load().ga_strerror(rc).decode()} ({rc})")


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(t):
    try:
        return DTYPE_CODE[t.dtype]
    except KeyError:
        raise GaError(f"unsupported dtype {t.dtype}") from None


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise GaError("guided-attention kernels run on the GPU only (got a CPU tensor); there is no CPU fallback")
