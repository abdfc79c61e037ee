"""ctypes binding of libishara_hip.so (C ABI declared in include/ishara_hip.h).

There is no CPU fallback: if the shared library is missing this module raises, and every
entry point that needs a GPU fails loudly through IsharaError.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libishara_hip.so")

F32, BF16, F16 = 0, 1, 2
FAMILY_KERAS_HYBRID, FAMILY_TORCH_CONFORMER, FAMILY_TORCH_SQUEEZEFORMER = 0, 1, 2


class IsharaError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("num_conv_squeeze_blocks", C.c_int32), ("num_conv_conform_blocks", C.c_int32),
        ("num_kernel_sizes", C.c_int32), ("kernel_sizes", C.c_int32 * 8), ("num_conv_per_block", C.c_int32),
        ("dropout_rate", C.c_float), ("num_heads", C.c_int32), ("expansion_factor", C.c_int32),
        ("transformer_kernel_size", C.c_int32), ("frames", C.c_int32), ("features", C.c_int32),
        ("num_classes", C.c_int32), ("top_dim", C.c_int32), ("squeeze_expansion", C.c_int32),
        ("conformer_expansion", C.c_int32), ("head_dropout", C.c_float), ("conformer_attn_dropout", C.c_float),
        ("dtype", C.c_int32), ("max_batch", C.c_int32), ("max_label_len", C.c_int32), ("attn_impl", C.c_int32),
        ("family", C.c_int32), ("reduce_layer_index", C.c_int32), ("recover_layer_index", C.c_int32), ("half_step_residual", C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/ishara_hip.h declares
_P, _I32, _I64, _U32, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_float
SIGNATURES = {
    "ishara_last_error": (C.c_char_p, []),
    "ishara_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "ishara_destroy": (None, [_P]),
    "ishara_param_total": (_I64, [_P]),
    "ishara_param_trainable": (_I64, [_P]),
    "ishara_param_entries": (_I32, [_P]),
    "ishara_param_info": (C.c_int, [_P, _I32, C.POINTER(C.c_char_p), C.POINTER(_I32), C.POINTER(_I64 * 2), C.POINTER(_I64), C.POINTER(_I32)]),
    "ishara_workspace_bytes": (_I64, [_P]),
    "ishara_workspace_plan_check": (_I32, [_P]),
    "ishara_workspace_guard_check": (C.c_int, [_P]),
    "ishara_grad_buckets": (_I32, [_P]),
    "ishara_grad_bucket": (C.c_int, [_P, _I32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ishara_grad_buckets_enable": (C.c_int, [_P]),
    "ishara_grad_bucket_wait": (C.c_int, [_P, _I32, _P]),
    "ishara_bind": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64]),
    "ishara_sync_weights": (C.c_int, [_P, _P]),
    "ishara_forward": (C.c_int, [_P, _P, _I32, _P, _I32, _U32, _P]),
    "ishara_encoder_forward": (C.c_int, [_P, _P, _I32, _P, _I32, _U32, _P]),
    "ishara_encoder_backward": (C.c_int, [_P, _P, _I32, _P, _P]),
    "ishara_encoder_output_frames": (_I32, [_P]),
    "ishara_loss_backward": (C.c_int, [_P, _P, _P, _I32, _P, _P, _F, _P]),
    "ishara_optimizer_step": (C.c_int, [_P, _F, _F, _P]),
    "ishara_optimizer_iterations": (_I32, [_P]),
    "ishara_optimizer_set_iterations": (C.c_int, [_P, _I32]),
    "ishara_profile_enable": (C.c_int, [_P, _I32]),
    "ishara_profile_report": (C.c_int, [_P, C.c_char_p, _I32]),
    "ishara_greedy_decode": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "ishara_preprocess": (C.c_int, [_P, _P, _I32, _P, _P, _P, _I32, _P]),
    "ishara_ctc_workspace_bytes": (_I64, [_I32, _I32, _I32]),
    "ishara_ctc_loss": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _F, _P, _P]),
    "ishara_dropout_mask": (C.c_int, [_U32, _U32, _I32, _I32, _F, _P, _P]),
    "ishara_debug_set_as_flags": (C.c_int, [_I32]),
    "ishara_debug_set_nt_big": (C.c_int, [_I32]),
    "ishara_debug_force_regstage": (C.c_int, [_I32]),
    "ishara_op_scratch_bytes": (_I64, [_I32, _I32, _I32]),
    "ishara_op_dense_fwd": (C.c_int, [_I32, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P]),
    "ishara_op_dense_fwd_ex": (C.c_int, [_I32, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P]),
    "ishara_op_dense_bwd": (C.c_int, [_I32, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P]),
    "ishara_op_log_softmax_fwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _P]),
    "ishara_op_log_softmax_bwd": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _P]),
    "ishara_op_layernorm_fwd": (C.c_int, [_I32, _P, _P, _P, _F, _P, _P, _P, _I32, _I32, _P]),
    "ishara_op_layernorm_bwd": (C.c_int, [_I32, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _P]),
    "ishara_op_dwconv_fwd": (C.c_int, [_I32, _I32, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "ishara_op_dwconv_scratch_bytes": (_I64, [_I32, _I32]),
    "ishara_op_dwconv_fwd_scratch_bytes": (_I64, [_I32, _I32, _I32]),
    "ishara_op_dwconv_fwd_ex": (C.c_int, [_I32, _I32, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "ishara_op_dwconv_bwd": (C.c_int, [_I32, _I32, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "ishara_op_attn_scratch_bytes": (_I64, [_I32, _I32, _I32, _I32]),
    "ishara_op_attn_fwd": (C.c_int, [_I32, _P, _P, _I32, _I32, _I32, _I32, _F, _U32, _U32, _F, _I32, _P, _P]),
    "ishara_op_attn_bwd": (C.c_int, [_I32, _P, _P, _P, _I32, _I32, _I32, _I32, _F, _U32, _U32, _F, _I32, _P, _P]),
}

_lib = None


def load():
    """Load libishara_hip.so (built by ishara_amd/build.py or `make -C ishara_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IsharaError(f"{LIB_PATH} is missing: build it with `python -m ishara_amd.build` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().ishara_last_error()
        raise IsharaError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
