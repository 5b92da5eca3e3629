"""ctypes binding of libm324.so -- the C ABI declared in include/m324.h.

The product path has no fallback: if the shared library is missing or a call fails, this module
raises.  Nothing here touches torch; motion324_amd.ops adapts tensors to raw pointers.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("M324_LIB") or os.path.join(HERE, "libm324.so")      # M324_LIB: lab builds (tools/lablibs)

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU = 0, 1
ABI_VERSION = 22
ERR_UNSUPPORTED = -3          # m324_status M324_ERR_UNSUPPORTED


class M324Error(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_long),
        ("W", C.c_void_p), ("ldw", C.c_long),
        ("C", C.c_void_p), ("ldc", C.c_long),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("in_dtype", C.c_int), ("out_dtype", C.c_int),
        ("bias", C.c_void_p),
        ("act", C.c_int),
        ("gamma", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", C.c_long), ("res_rows", C.c_int),
        ("row_gin", C.c_int), ("row_gout", C.c_int), ("row_off", C.c_int),
        ("batch", C.c_int),
        ("strideA", C.c_long), ("strideW", C.c_long), ("strideC", C.c_long),
        ("aux", C.c_void_p), ("ldaux", C.c_long), ("aux_mode", C.c_int),
        ("qkv_q", C.c_void_p), ("qkv_k", C.c_void_p), ("qkv_v", C.c_void_p),
        ("qkv_qw", C.c_void_p), ("qkv_kw", C.c_void_p),
        ("qkv_eps", C.c_float), ("qkv_qscale", C.c_float), ("qkv_L", C.c_int), ("qkv_H", C.c_int),
        ("ln_rowstat", C.c_void_p), ("ln_colsum", C.c_void_p),
        ("ln_stats_out", C.c_void_p),
        ("ln_copy_out", C.c_void_p), ("ln_ldcopy", C.c_long), ("ln_ncb", C.c_int), ("ln_eps", C.c_float),
    ]


class ColsumItem(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("ld", C.c_long), ("rows", C.c_int), ("cols", C.c_int),
                ("accumulate", C.c_int), ("chain", C.c_int)]


class MirrorItem(C.Structure):
    _fields_ = [("src_off", C.c_long), ("dst_off", C.c_long), ("dstT_off", C.c_long), ("first_tile", C.c_long), ("rows", C.c_int),
                ("cols", C.c_int), ("ldT", C.c_int), ("pad_", C.c_int)]


_P, _L, _I, _F = C.c_void_p, C.c_long, C.c_int, C.c_float

# name -> argtypes: exactly the declarations of include/m324.h
SIGNATURES = {
    "m324_abi_version": [],
    "m324_last_error": [C.c_char_p, _I],
    "m324_device_info": [C.c_char_p, _I],
    "m324_set_tunable": [C.c_char_p, _I],
    "m324_gemm": [C.POINTER(GemmArgs), _P],
    "m324_gemm_plan": [C.POINTER(GemmArgs), C.c_char_p, _I],
    "m324_gemm_pair": [C.POINTER(GemmArgs), C.POINTER(GemmArgs), _P],
    "m324_attention_plan": [_I, _I, _I, _I, _I, _I, C.c_char_p, _I],
    "m324_gemm_tn": [_P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _L, _P],
    "m324_n3_finish": [_P, _I, _I, _P, _P, _P],
    "m324_rowstats_finish": [_P, _I, _I, _F, _P, _P],
    "m324_rowstats": [_P, _L, _I, _I, _F, _P, _P, _L, _P],
    "m324_layernorm": [_P, _L, _P, _P, _F, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "m324_layernorm_in": [_P, _I, _L, _P, _P, _F, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "m324_layernorm_pair": [_P, _L, _P, _P, _F, _P, _L, _I, _I, _I, _I, _P, _L, _P, _P, _F, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "m324_qkv_split": [_P, _L, _P, _L, _P, _L, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "m324_attention": [_P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _F, _I, _P, _I, _P],
    "m324_attention_merge": [_P, _P, _P, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _P],
    "m324_patchify": [_P, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    "m324_patchify_u8": [_P, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    "m324_point_encode": [_P, _I, _P, _L, _I, _P],
    "m324_point_concat": [_P, _P, _I, _P, _I, _I, _I, _P],
    "m324_dino_cls_rows": [_P, _P, _P, _I, _I, _I, _P],
    "m324_assemble_tokens": [_P, _P, _P, _F, _P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _F, C.c_ulonglong, _P],
    "m324_linear_n3": [_P, _L, _P, _P, _P, _I, _I, _I, _P],
    "m324_mse": [_P, _P, _L, _F, _P, _P, _P],
    "m324_smooth_trajectories": [_P, _P, _P, _I, _I, _I, _F, _F, _P],
    "m324_smooth_savgol": [_P, _P, _I, _I, _I, _P, _I, _P],
    "m324_smooth_oneeuro": [_P, _P, _I, _I, _I, _F, _F, _F, _P],
    "m324_nearest_point": [_P, _I, _P, _I, _P, _P],
    "m324_transpose": [_P, _L, _P, _L, _I, _I, _I, _I, _P],
    "m324_colsum": [_P, _L, _P, _I, _I, _I, _I, _P, _I, _P],
    "m324_colsum_multi": [C.POINTER(ColsumItem), _I, _P],
    "m324_gelu": [_P, _P, _L, _I, _P],
    "m324_gelu_bwd": [_P, _P, _P, _L, _I, _P],
    "m324_cast": [_P, _L, _I, _P, _L, _I, _I, _I, _P],
    "m324_attention_delta": [_P, _P, _L, _P, _I, _I, _I, _I, _P],
    "m324_attention_bwd": [_P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "m324_attention_bwd_mfma": [_P, _P, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "m324_qkv_split_bwd": [_P, _P, _P, _P, _L, _P, _L, _P, _P, _F, _P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _P],
    "m324_linear_n3_bwd": [_P, _L, _P, _P, _P, _L, _P, _I, _I, _I, _I, _P, _L, _P],
    "m324_mse_bwd": [_P, _P, _P, _F, _P, _L, _P],
    "m324_adamw": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P, _P],
    "m324_adamw_flat": [_P, _P, _P, _P, _L, _L, _F, _F, _F, _F, _F, _I, _P, _P],
    "m324_weight_mirror": [_P, _P, _P, _P, _I, _L, _P],
    "m324_grad_sumsq": [_P, _L, _I, _P, _P, _I, _P],
    "m324_comm_unique_id": [C.c_char_p, _I],
    "m324_comm_init": [C.POINTER(C.c_void_p), C.c_char_p, _I, _I],
    "m324_comm_allreduce": [_P, _P, _L, _I, _I, _P],
    "m324_comm_allgather": [_P, _P, _P, _L, _I, _P],
    "m324_comm_destroy": [_P],
    "m324_layernorm_bwd": [_P, _L, _P, _F, _P, _L, _I, _P, _L, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "m324_layernorm_bwd_cast": [_P, _L, _P, _F, _P, _L, _I, _P, _L, _I, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P],
}

_lib = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """Loads libm324.so (once) and installs argtypes.  Raises M324Error when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise M324Error(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m motion324_amd.build` "
                "(there is no CPU or PyTorch fallback on this path).")
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is not exported
            fn.argtypes = argtypes
            fn.restype = C.c_int
        if lib.m324_abi_version() != ABI_VERSION:
            raise M324Error(f"libm324 ABI {lib.m324_abi_version()} != binding {ABI_VERSION}; rebuild")
        _lib = lib
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().m324_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise M324Error(f"{what} failed ({rc}): {last_error()}")


TUNABLE_DEFAULT = -2 ** 31      # m324_set_tunable(name, INT_MIN) restores the load-time default


def set_tunable(name: str, value: int = TUNABLE_DEFAULT) -> None:
    """Lab / test hook: overrides one of the kernel-chooser switches that libm324 otherwise reads from the
    environment once, at load (M324_GEMM, M324_ATTN_NW, ...).  `value` omitted = back to the default."""
    check(load().m324_set_tunable(name.encode(), int(value)), "m324_set_tunable")


class tunable:
    """with tunable("M324_GEMM", 10): ...   (restores the default afterwards)"""

    def __init__(self, name: str, value):
        self.name = name
        self.value = int(str(value).lstrip("vV")) if value is not None else TUNABLE_DEFAULT

    def __enter__(self):
        set_tunable(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_tunable(self.name)


def device_info():
    buf = C.create_string_buffer(256)
    n = load().m324_device_info(buf, 256)
    check(0 if n >= 0 else n, "m324_device_info")
    return buf.value.decode(), n
