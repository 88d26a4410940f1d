"""ctypes binding of libafm_hip.so (the C ABI declared in include/afm_hip.h).

There is no fallback: if the library is missing and cannot be built, or a call returns an
error code, this raises.  PyTorch is used only as the owner of device memory and streams:
every argument crossing the boundary is a raw device pointer, an integer or a float.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libafm_hip.so")

AFM_F32, AFM_BF16, AFM_BF16X2, AFM_F16 = 0, 1, 2, 3
ABI_VERSION = 6
ACT_NONE, ACT_RELU, ACT_GELU, ACT_GELU_BWD, ACT_GELU_SAVE_GRAD, ACT_MUL_SAVED = 0, 1, 2, 3, 4, 5
ACT_GLU, ACT_GLU_SAVE, ACT_GLU_BWD = 6, 7, 8
ALGO_AUTO, ALGO_GENERIC, ALGO_MFMA = 0, 1, 2
ERR_ARG, ERR_UNSUPPORTED, ERR_LAUNCH = -1, -2, -3


class AfmError(RuntimeError):
    pass


class Dropout(C.Structure):
    _fields_ = [("p", C.c_float), ("site", C.c_uint32), ("seed", C.c_uint64)]


class PatchDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("L", C.c_int32), ("patch_size", C.c_int32), ("step", C.c_int32),
                ("interpolation", C.c_int32), ("derivative", C.c_int32), ("masking", C.c_int32),
                ("seq_first", C.c_int32), ("mean", C.c_double), ("std", C.c_double)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("transA", C.c_int32), ("transB", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
        ("a_dtype", C.c_int32), ("b_dtype", C.c_int32), ("c_dtype", C.c_int32),
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("bias", C.c_void_p), ("residual", C.c_void_p), ("pre_act", C.c_void_p), ("a_colsum", C.c_void_p),
        ("act", C.c_int32), ("accumulate", C.c_int32), ("algo", C.c_int32), ("reserved", C.c_int32),
        ("drop", Dropout),
        ("glu_rows", C.c_int32), ("reserved2", C.c_int32),
        ("k_live", C.c_void_p),
    ]


class LnShape(C.Structure):
    _fields_ = [
        ("rows", C.c_int64), ("d", C.c_int32), ("y_dtype", C.c_int32),
        ("seg_len", C.c_int64), ("out_seg_stride", C.c_int64), ("out_off", C.c_int64),
        ("eps", C.c_float), ("add_dtype", C.c_int32), ("add_drop", Dropout),
        ("row_live", C.c_void_p), ("flags", C.c_int32), ("reserved", C.c_int32), ("row_map", C.c_void_p),
    ]


class CastItem(C.Structure):
    """afm_cast_item (afm_cast_weights_batch): one matrix of the batched weight-shadow cast."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("dst_t", C.c_void_p),
                ("rows", C.c_int32), ("cols", C.c_int32), ("glu_rows", C.c_int32), ("tile0", C.c_int32)]


class BeamDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("k", C.c_int32), ("V", C.c_int32), ("ldl", C.c_int32),
        ("cur_len", C.c_int32), ("max_length", C.c_int32),
        ("eos", C.c_int32), ("pad", C.c_int32), ("stop_rule", C.c_int32), ("Lmax", C.c_int32),
        ("logits", C.c_void_p), ("seq_in", C.c_void_p), ("seq_out", C.c_void_p), ("beam_scores", C.c_void_p),
        ("beam_idx", C.c_void_p), ("hyp_seq", C.c_void_p), ("hyp_score", C.c_void_p), ("hyp_len", C.c_void_p),
        ("hyp_count", C.c_void_p), ("done", C.c_void_p), ("n_open", C.c_void_p),
    ]


class AttnShape(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("H", C.c_int32), ("Tq", C.c_int32), ("Tk", C.c_int32), ("dh", C.c_int32),
        ("dtype", C.c_int32),
        ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldv", C.c_int32), ("ldo", C.c_int32),
        ("causal", C.c_int32), ("algo", C.c_int32), ("scale", C.c_float), ("reserved", C.c_int32),
        ("key_pad", C.c_void_p),
        ("drop", Dropout),
        ("sqb", C.c_int64), ("skb", C.c_int64), ("svb", C.c_int64), ("sob", C.c_int64),
        ("drop_bits", C.c_void_p),
        ("q_off", C.c_void_p), ("k_off", C.c_void_p),
    ]


_P = C.c_void_p
_I32, _I64, _F = C.c_int32, C.c_int64, C.c_float
_SIGS = {
    "afm_abi_version": (C.c_int, []),
    "afm_struct_size": (C.c_int, [C.c_int]),
    "afm_relu_bwd": (C.c_int, [_P, _P, _P, _I64, _P]),
    "afm_convert": (C.c_int, [_P, _I32, _I32, _P, _I32, _I32, _I64, _I32, _P]),
    "afm_cast_x2": (C.c_int, [_P, _P, _P, _I32, _I32, _P]),
    "afm_cast_weights": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "afm_cast_weights_batch": (C.c_int, [_P, _I32, _I32, _I32, _P]),
    "afm_error_string": (C.c_char_p, [C.c_int]),
    "afm_last_algo": (C.c_char_p, []),
    "afm_last_hint": (C.c_int, []),
    "afm_gemm": (C.c_int, [C.POINTER(GemmDesc), _P]),
    "afm_gemm_group": (C.c_int, [_P, C.c_int32, _P]),
    "afm_gather_rows": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _P]),
    "afm_scatter_add_rows": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _I64, _P]),
    "afm_layernorm_fwd": (C.c_int, [C.POINTER(LnShape), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "afm_layernorm_bwd_ws_floats": (C.c_int64, [C.POINTER(LnShape)]),
    "afm_layernorm_bwd": (C.c_int, [C.POINTER(LnShape), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(Dropout), _P]),
    "afm_attn_fwd": (C.c_int, [C.POINTER(AttnShape), _P, _P, _P, _P, _P, _P]),
    "afm_attn_drop_bits_fill": (C.c_int, [C.POINTER(AttnShape), _P]),
    "afm_attn_bwd": (C.c_int, [C.POINTER(AttnShape), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                               _I32, _I32, _I32, _P]),
    "afm_glu_fwd": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, C.POINTER(Dropout), _P]),
    "afm_glu_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32,
                              C.POINTER(Dropout), _P]),
    "afm_dropout_cast": (C.c_int, [_P, _P, _I64, _I32, _I32, _I32, _I32, C.POINTER(Dropout), _P]),
    "afm_colsum": (C.c_int, [_P, _P, _I64, _I32, _I32, _I32, _I32, _P]),
    "afm_add_inplace": (C.c_int, [_P, _P, _I64, _P]),
    "afm_batch_sum": (C.c_int, [_P, _P, _I32, _I64, _I32, _I32, _P]),
    "afm_cast_bf16": (C.c_int, [_P, _P, _P, _I32, _I32, _P]),
    "afm_masked_mean_fwd": (C.c_int, [_P, _I32, _P, _I32, _I32, _I32, _P, _P]),
    "afm_masked_mean_bwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _I32, _P]),
    "afm_align_loss": (C.c_int, [_P, _P, _I32, _I32, _I32, _F, _P, _P, _P, _P]),
    "afm_mix_spectra": (C.c_int, [_P, _I64, _I32, _P, _I32, _I32, _P, _I32, _I32, _P, _P]),
    "afm_patch_count": (C.c_int32, [C.POINTER(PatchDesc)]),
    "afm_patch_preprocess": (C.c_int, [C.POINTER(PatchDesc), _P, _P, _P, _P, _P]),
    "afm_ce_fwd": (C.c_int, [_P, _P, _I64, _I32, _I32, _P, _P, _P, _P]),
    "afm_ce_bwd": (C.c_int, [_P, _P, _P, _P, _F, _P, _P, _I32, _I32, _I64, _I32, _I32, _P]),
    "afm_beam_step": (C.c_int, [C.POINTER(BeamDesc), _P]),
    "afm_beam_finalize": (C.c_int, [C.POINTER(BeamDesc), _P, _P, _P, _P]),
    "afm_cache_reorder": (C.c_int, [_P, _P, _P, _I32, _I64, _I64, _P]),
    "afm_sumsq": (C.c_int, [_P, _I64, _P, _P, _P]),
    "afm_adam_step": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _P, _I32, _I32, _P, _P]),
    "afm_scaler_update": (C.c_int, [_P, _P, _F, _F, _I32, _P]),
    "afm_place_rows": (C.c_int, [_P, _P, _P, _I64, _I32, _I64, _I64, _I64, _I32, _P]),
    "afm_compact_plan": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "afm_permute_rows": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "afm_comm_unique_id": (C.c_int, [_P]),
    "afm_comm_create": (C.c_int, [C.POINTER(_P), _P, _I32, _I32]),
    "afm_allreduce_bucket": (C.c_int, [_P, _P, _I64, _P]),
    "afm_comm_count": (C.c_int, [_P, C.POINTER(_I32), C.POINTER(_I32)]),
    "afm_comm_destroy": (C.c_int, [_P]),
}

_lock = threading.Lock()
_lib = None


def exported_symbols():
    return sorted(_SIGS)


def load(build_if_missing: bool = True):
    """dlopen libafm_hip.so (building it with hipcc first if absent) and type every symbol."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        # (re)build whenever hipcc is here: csrc/build.py stamps every object with a digest of its sources, so
        # this is a no-op for a fresh library and a stale git-ignored .so can never be loaded silently
        override = os.environ.get("AFM_LIB_OVERRIDE")   # timing experiments: an ablation build of the same sources (tools/)
        if override:
            build_if_missing = False
        if build_if_missing:
            from .csrc import build as _b
            if _b.have_hipcc():
                _b.build()
        lib_path = override or LIB_PATH
        if not os.path.exists(lib_path):
            raise AfmError(f"{lib_path} is missing; run `python -m multimodalanalytical_amd.csrc.build`")
        # torch ships its own libamdhip64: it must be the HIP runtime already resident when this library's
        # dependency on that soname is resolved, or the process ends up with two runtimes (the system one
        # bound here, torch's owning every stream and pointer) and every launch fails
        import torch  # noqa: F401
        try:
            lib = C.CDLL(lib_path)
        except OSError as e:  # no CPU fallback by design
            raise AfmError(f"cannot load {lib_path}: {e}") from e
        for name, (res, args) in _SIGS.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise AfmError(f"{LIB_PATH} does not export {name}") from e
            fn.restype, fn.argtypes = res, args
        if lib.afm_abi_version() != ABI_VERSION:
            raise AfmError(f"libafm_hip.so ABI version {lib.afm_abi_version()} != binding {ABI_VERSION}: rebuild it")
        for which, st in enumerate((Dropout, GemmDesc, LnShape, AttnShape, PatchDesc, BeamDesc, CastItem)):
            if lib.afm_struct_size(which) != C.sizeof(st):
                raise AfmError(f"libafm_hip.so was built with a different {st.__name__} layout: rebuild it")
        _lib = lib
        return lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().afm_error_string(code).decode()
        raise AfmError(f"{what or 'libafm_hip'}: {msg} ({code})")
