"""ctypes loader for libhashmod.so (the C ABI declared in include/hashmod.h).

There is NO fallback: if the library is missing or a call fails, the product raises.
PyTorch is used only for device memory (``tensor.data_ptr()``) and streams.
"""
import ctypes as C
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# HM_LIB_PATH: kernel-experiment builds of the same ABI (profiling only)
LIB_PATH = os.environ.get("HM_LIB_PATH") or os.path.join(_PKG, "libhashmod.so")
_lib = None

_p = C.c_void_p
_i64 = C.c_int64
_int = C.c_int

# name -> (restype, argtypes); must list every symbol include/hashmod.h declares
SIGNATURES = {
    "hm_version": (_int, []),
    "hm_last_error": (C.c_char_p, []),
    "hm_device_count": (_int, []),
    "hm_grid_desc_create": (_int, [_int, _int, _p, _p, _p, C.POINTER(_p)]),
    "hm_grid_desc_destroy": (None, [_p]),
    "hm_grid_embed_dim": (_int, [_p]),
    "hm_corner_ids": (_int, [_p, _int, _p, _i64, _p, _p, _p]),
    "hm_diag_gather_calib": (_int, [_p, _i64, _i64, _int, _int, _p, _p]),
    "hm_diag_mfma_f32_stream": (_int, [_int, _int, _p, _p]),
    "hm_encode_fwd": (_int, [_p, _p, _i64, _p, _p, _p, _i64, _int, _p]),
    "hm_encode_workspace_bytes": (_i64, [_p, _i64]),
    "hm_encode_fwd_ws": (_int, [_p, _p, _i64, _p, _p, _p, _i64, _int, _p, _i64, _p]),
    "hm_encode_bwd_table": (_int, [_p, _p, _i64, _p, _i64, _p, _int, _p]),
    "hm_encode_bwd_workspace_bytes": (_i64, [_p, _i64]),
    "hm_encode_bwd_table_ws": (_int, [_p, _p, _i64, _p, _i64, _p, _int, _p, _i64, _p]),
    "hm_encode_rows": (_int, [_p, _p, _i64, _int, _p, _p, _p]),
    "hm_encode_bwd_table_sorted": (_int, [_p, _p, _p, _i64, _int, _p, _i64, _p, _p, _p]),
    "hm_sort_workspace_bytes": (_i64, [_i64]),
    "hm_sort_pairs_i32": (_int, [_p, _i64, _int, _p, _p, _p, _i64, _p]),
    "hm_encode_bwd_input": (_int, [_p, _p, _i64, _p, _p, _i64, _p, _p, _p]),
    "hm_encode_jvp": (_int, [_p, _p, _i64, _p, _p, _p, _i64, _p]),
    "hm_encode_bwd_table_jvp": (_int, [_p, _p, _i64, _p, _p, _i64, _p, _p]),
    "hm_fourier_bwd_input": (_int, [_p, _i64, _p, _int, _p, _i64, _p, _p]),
    "hm_fourier_bwd_input_bwd": (_int, [_p, _i64, _p, _int, _p, _i64, _p, _p, _p, _i64, _int, _p]),
    "hm_weight_norm_multi": (_int, [_int, _int, _p, _p]),
    "hm_rownorm": (_int, [_int, _p, _p, _p, _p, _p, _i64, _int, C.c_float, _p]),
    "hm_sine": (_int, [_int, _p, _p, _p, _p, _p, _i64, C.c_float, _p]),
    "hm_posenc": (_int, [_int, _p, _int, _int, _p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _p]),
    "hm_sdf_fwd": (_int, [_p, _p, _p, _i64, _p, _p, _p, _i64, _int, _int, _int, _p, _int, _p]),
    "hm_nffb_fwd": (_int, [_p, _p, _p, _i64, _p, _p, _p, _i64, _int, _p, _p]),
    "hm_sdf_fwd_emb": (_int, [_p, _p, _i64, _int, _i64, _p, _i64, _int, _int, _p, _int, _p]),
    "hm_trace_workspace_bytes": (_i64, [_i64, _p]),
    "hm_trace_forward": (_int, [_p, _p, _p, _p, _int, _int, _p, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _p, _p, _p,
                                _i64, _p, _p]),
    "hm_trace_workspace_bytes_nffb": (_i64, [_i64, _p, _int]),
    "hm_trace_forward_nffb": (_int, [_p, _p, _p, _p, _p, _int, _int, _p, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _p,
                                     _p, _p, _i64, _p, _p]),
    "hm_pack_mlp_layer_bf16": (_int, [_p, _i64, _int, _int, _int, _p, _p]),
    "hm_sdf_fwd_bf16": (_int, [_p, _p, _p, _i64, _p, _p, _p, _i64, _int, _p, _i64, _p]),
    "hm_sdf_fwd_emb_bf16": (_int, [_p, _p, _i64, _int, _i64, _p, _i64, _p, _i64, _p]),
    "hm_pack_mlp_layer_split": (_int, [_p, _i64, _int, _int, _int, C.c_float, C.c_float, _int, _p, _p]),
    "hm_sdf_fwd_split": (_int, [_p, _p, _p, _i64, _p, _p, _p, _i64, _int, _p, _i64, _p]),
    "hm_sdf_fwd_emb_split": (_int, [_p, _p, _i64, _int, _i64, _p, _i64, _p, _i64, _p]),
    "hm_pack_mlp_layers": (_int, [_p, _int, _p]),
    "hm_pack_mlp_layer": (_int, [_p, _i64, _p, _int, _int, _int, _p, _p, _p, _p]),
    "hm_softplus": (_int, [_int, _p, _p, _p, _p, _p, _i64, C.c_float, C.c_float, _p]),
    "hm_sdf_head": (_int, [_int, _p, _i64, _i64, C.c_float, _p, _p, _p, _p, _p, _p]),
    "hm_colsum": (_int, [_p, _i64, _i64, _i64, _p, _p]),
    "hm_colsum_acc": (_int, [_p, _i64, _i64, _i64, _p, _p]),
    "hm_colsum_acc_multi": (_int, [_p, _int, _p]),
    "hm_copy2d_f32": (_int, [_p, _i64, _p, _i64, _i64, _i64, _p]),
    "hm_camera_rays": (_int, [_p, _p, _p, _i64, _i64, C.c_float, _p, _p, _p, _p, _p]),
    "hm_idr_loss": (_int, [_p, _p, _p, _p, _p, _i64, _p, _i64, C.c_float, C.c_float, C.c_float, _p, _p, _p, _p, _p]),
    "hm_adam_scratch_floats": (_i64, [_p, _int]),
    "hm_adam_step": (_int, [_p, _int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _p, _p]),
    "hm_encode_bwd_table_tracked": (_int, [_p, _p, _i64, _p, _i64, _p, _int, _p, _p, _p, _i64, _p]),
    "hm_rows_pack": (_int, [_p, _int, _p, _p, _i64, _p, _p, _p, _p]),
    "hm_rows_apply": (_int, [_p, _i64, _int, _p, _i64, _i64, _int, C.c_float, _p]),
    "hm_rows_clear": (_int, [_p, _i64, _int, _p, _i64, _i64, _int, _p, _p]),
    "hm_multi_copy_f32": (_int, [_p, _int, _p]),
    "hm_gemm_f32_group_tn": (_int, [_p, _int, _p]),
    "hm_gemm_f32_ep": (_int, [_int, _int, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _p]),
    "hm_gemm_f32": (_int, [_int, _int, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _int, _p]),
}


class MlpLayer(C.Structure):
    _fields_ = [("w_packed", C.c_void_p), ("bias", C.c_void_p), ("out_dim", C.c_int32), ("n_tiles", C.c_int32),
                ("seg_octets", C.c_int32 * 2), ("seg_src", C.c_int32 * 2), ("activation", C.c_int32),
                ("post_div_sqrt2", C.c_int32), ("w_packed_m16", C.c_void_p), ("seg_blocks16", C.c_int32 * 2),
                ("w_packed_bf16", C.c_void_p), ("w_packed_split", C.c_void_p)]


class GemmEpilogue(C.Structure):
    _fields_ = [("mode", C.c_int32), ("nz", C.c_int32), ("scale", C.c_float), ("beta", C.c_float),
                ("threshold", C.c_float), ("z", C.c_void_p), ("ldz", C.c_int64), ("g", C.c_void_p), ("ldg", C.c_int64),
                ("out1", C.c_void_p), ("ld1", C.c_int64), ("out2", C.c_void_p), ("ld2", C.c_int64),
                ("out3", C.c_void_p), ("ld3", C.c_int64)]


class AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("step", C.c_void_p), ("numel", C.c_int64)]


class GemmGroupItem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("M", C.c_int64), ("N", C.c_int64),
                ("K", C.c_int64), ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64)]


class ColsumItem(C.Structure):
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("M", C.c_int64), ("N", C.c_int64), ("ld", C.c_int64)]


class PackItem(C.Structure):
    _fields_ = [("W", C.c_void_p), ("bias", C.c_void_p), ("w_packed", C.c_void_p), ("w_packed_m16", C.c_void_p),
                ("bias_padded", C.c_void_p), ("ldw", C.c_int64), ("out_dim", C.c_int32), ("seg_width0", C.c_int32),
                ("seg_width1", C.c_int32), ("pad_", C.c_int32)]


class CopyItem(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("numel", C.c_int64)]


class TraceCfg(C.Structure):
    _fields_ = [("object_bounding_sphere", C.c_float), ("sdf_threshold", C.c_float), ("line_search_step", C.c_double),
                ("line_step_iters", C.c_int32), ("sphere_tracing_iters", C.c_int32), ("n_steps", C.c_int32),
                ("n_secant_steps", C.c_int32), ("training", C.c_int32), ("coarse_bf16", C.c_int32),
                ("sampler_head", C.c_int32)]


class WnLayer(C.Structure):
    _fields_ = [("v", C.c_void_p), ("g", C.c_void_p), ("w", C.c_void_p), ("norm", C.c_void_p), ("grad_w", C.c_void_p),
                ("grad_v", C.c_void_p), ("grad_g", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32)]


class NffbDesc(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("bound", C.c_float), ("w0", C.c_float), ("style_eps", C.c_float),
                ("trunk_w", C.c_void_p * 32), ("trunk_b", C.c_void_p * 32), ("out_w", C.c_void_p), ("out_b", C.c_void_p),
                ("style_w", C.c_void_p), ("style_b", C.c_void_p)]


class MlpDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("beta", C.c_float), ("layer", MlpLayer * 16), ("split_kind", C.c_int32)]


class HashmodError(RuntimeError):
    pass


# Parameters written through raw pointers (hm_adam_step, graph replays) do not bump torch's tensor._version, so
# every cache derived from parameter VALUES (ImplicitNetwork.packed_weights) also keys on this counter; every
# writer that bypasses ATen calls bump_param_epoch().
_param_epoch = [0]


def param_epoch():
    return _param_epoch[0]


def bump_param_epoch():
    _param_epoch[0] += 1


def lib():
    """Load libhashmod.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HashmodError(
                f"{LIB_PATH} is missing - build it with `python -m hashmodnffbanks_idr_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no CPU / PyTorch fallback for the hot path")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc < 0:
        msg = lib().hm_last_error()
        msg = msg.decode() if msg else "unknown error"
        if rc == -1:
            raise ValueError(f"hashmod: {msg}")
        raise HashmodError(f"hashmod: {msg} (code {rc}) {what}")
    return rc


def stream_ptr(t):
    """hipStream_t of torch's CURRENT stream on t's device (never the legacy default stream)."""
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def dptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HashmodError("hashmod: the hot path runs only on HIP (cuda) tensors; got a CPU tensor. "
                               "There is deliberately no CPU fallback (see oracle/ for the CPU checker).")
