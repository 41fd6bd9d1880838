"""ctypes binding of libsinddm_hip.so (the C ABI declared in include/sinddm_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, an
exception is raised.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsinddm_hip.so")

# every symbol include/sinddm_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = (
    "sinddm_abi_version", "sinddm_param_count", "sinddm_param_tensors", "sinddm_param_offset",
    "sinddm_packed_count", "sinddm_workspace_bytes", "sinddm_pack_weights", "sinddm_net_forward",
    "sinddm_q_sample", "sinddm_reverse_step", "sinddm_reverse_step_edit", "sinddm_upsample_bilinear", "sinddm_prof_begin", "sinddm_prof_end", "sinddm_prof_end2", "sinddm_prof_end3", "sinddm_debug_wgrad_map", "sinddm_debug_conv_path",
    "sinddm_debug_block_train", "sinddm_debug_infer_path", "sinddm_debug_train_path",
    "sinddm_train_workspace_bytes", "sinddm_packed_bwd_count", "sinddm_pack_weights_bwd",
    "sinddm_net_forward_train", "sinddm_net_backward", "sinddm_l1_loss_fwd_bwd", "sinddm_adam_ema_step",
    "sinddm_cond_embed", "sinddm_cond_stride", "sinddm_sample_chain", "sinddm_sample_chain2", "sinddm_normal_fill",
)


ABI_VERSION = 3               # SINDDM_ABI_VERSION of include/sinddm_hip.h this binding was written against
DIM_FP32_CONVS = 0x10000     # SINDDM_DIM_FP32_CONVS of include/sinddm_hip.h: option bit of every `dim` argument


class StepCoefs(C.Structure):
    """Mirror of `sinddm_step_coefs` (include/sinddm_hip.h)."""
    _fields_ = [("mode", C.c_int), ("clip", C.c_int),
                ("sqrt_recip_ac_t", C.c_float), ("sqrt_recipm1_ac_t", C.c_float),
                ("coef1_t", C.c_float), ("coef2_t", C.c_float),
                ("gamma_t", C.c_float), ("gamma_tm1", C.c_float),
                ("sqrt_ac_tm1", C.c_float), ("sqrt_ac_t", C.c_float), ("sqrt_1m_ac_t", C.c_float),
                ("sqrt_1m_ac_tm1_mvar", C.c_float), ("sigma", C.c_float)]


class SinddmError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load (once) and type the library.  Raises if it was not built -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SinddmError(
            f"{LIB_PATH} is missing: build it with `python -m sinddm_amd.build` (hipcc, gfx950). "
            "sinddm_amd has no CPU / PyTorch fallback for the hot path.")
    from . import build as _build
    try:
        _build.verify()                      # a binary built from other sources than the ones beside it is refused
    except RuntimeError as e:
        raise SinddmError(str(e)) from None
    lib = C.CDLL(LIB_PATH)
    p, i, i64, f, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
    sig = {
        "sinddm_abi_version": (i, []),
        "sinddm_param_count": (i64, [i]),
        "sinddm_param_tensors": (i, [i]),
        "sinddm_param_offset": (i64, [i, i]),
        "sinddm_packed_count": (i64, [i]),
        "sinddm_workspace_bytes": (sz, [i, i, i, i]),
        "sinddm_pack_weights": (i, [p, p, i, p]),
        "sinddm_net_forward": (i, [p, p, p, p, i, f, p, i, i, i, i, p, sz, p]),
        "sinddm_cond_embed": (i, [p, p, i, f, i, i, p, p, p, p]),
        "sinddm_cond_stride": (i, [i]),
        "sinddm_q_sample": (i, [p, p, p, p, p, p, p, p, i, i, i64, p]),
        "sinddm_reverse_step": (i, [p, p, p, p, p, C.POINTER(StepCoefs), i64, p]),
        "sinddm_reverse_step_edit": (i, [p, p, p, p, p, C.POINTER(StepCoefs), p, p, i, i, i, p]),
        "sinddm_sample_chain": (i, [p, p, p, p, p, p, C.POINTER(StepCoefs), C.POINTER(C.c_int), i, f, C.c_uint64, C.c_uint64,
                                    i, i, i, i, p, sz, p, C.POINTER(C.c_int)]),
        "sinddm_sample_chain2": (i, [p, p, p, p, p, p, C.POINTER(StepCoefs), C.POINTER(C.c_int), i, f, C.c_uint64, C.c_uint64,
                                     i, i, i, i, p, sz, p, p, C.POINTER(C.c_int)]),
        "sinddm_normal_fill": (i, [p, i64, C.c_uint64, C.c_uint64, p]),
        "sinddm_upsample_bilinear": (i, [p, p, i, i, i, i, i, p]),
        "sinddm_prof_begin": (i, []),
        "sinddm_prof_end": (i, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
        "sinddm_prof_end2": (i, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                 C.POINTER(C.c_double)]),
        "sinddm_prof_end3": (i, [i, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                 C.POINTER(C.c_double), i]),
        "sinddm_debug_wgrad_map": (i, [i, i, i64, i, C.POINTER(C.c_uint32), i, C.POINTER(C.c_int32), i]),
        "sinddm_debug_conv_path": (i, [i, i, i, i]),
        "sinddm_debug_infer_path": (i, [i, i, i, i]),
        "sinddm_debug_train_path": (i, [i, i, i, i]),
        "sinddm_debug_block_train": (i, [p, p, p, i, i, p, p, p, p, p, p, p, i, i, i, p, sz, p]),
        "sinddm_train_workspace_bytes": (sz, [i, i, i, i]),
        "sinddm_packed_bwd_count": (i64, [i]),
        "sinddm_pack_weights_bwd": (i, [p, p, i, p]),
        "sinddm_net_forward_train": (i, [p, p, p, p, i, f, p, i, i, i, i, p, sz, p]),
        "sinddm_net_backward": (i, [p, p, p, p, p, p, p, i, i, i, i, p, sz, p]),
        "sinddm_l1_loss_fwd_bwd": (i, [p, p, p, p, i64, f, p]),
        "sinddm_adam_ema_step": (i, [p, p, p, p, p, f, f, f, f, f, f, f, i, i64, p]),
    }
    for name, (res, args) in sig.items():
        if not hasattr(lib, name):
            continue  # reported by check_symbols(); calling it raises AttributeError loudly
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def missing_symbols():
    lib = load()
    return [s for s in ABI_SYMBOLS if not hasattr(lib, s)]


def check(rc: int, what: str) -> None:
    if rc != 0:
        kind = {-1: "bad argument", -2: "unsupported shape", -3: "workspace too small"}.get(rc, f"hipError {rc}")
        raise SinddmError(f"{what} failed: {kind}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SinddmError("sinddm_amd kernels need tensors on a ROCm device (no CPU fallback)")
    if not t.is_contiguous():
        raise SinddmError("sinddm_amd kernels need contiguous tensors")
    return t.data_ptr()


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream
