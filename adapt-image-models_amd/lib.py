"""ctypes binding of libaim_hip.so (C ABI: include/aim_kernels.h)."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class LibraryNotBuilt(RuntimeError):
    pass


def library_path() -> str:
    # AIM_HIP_LIB selects an experimental build of the same library (tools/ only); there is still no fallback
    return os.environ.get("AIM_HIP_LIB") or os.path.join(_HERE, "libaim_hip.so")


class GemmArgs(Structure):
    """Mirror of ``aim_gemm_args``."""
    _fields_ = [
        ("A", c_void_p), ("W", c_void_p), ("lda", c_int32), ("ldw", c_int32),
        ("strideA", c_int64), ("strideW", c_int64),
        ("M", c_int32), ("N", c_int32), ("K", c_int32),
        ("bias", c_void_p), ("resid", c_void_p), ("ldr", c_int32),
        ("af", c_void_p), ("at", c_void_p), ("vec", c_void_p), ("bt", c_void_p),
        ("ldv", c_int32), ("ntok", c_int32),
        ("aux", c_void_p), ("ldaux", c_int32),
        ("out", c_void_p), ("ldo", c_int32), ("out2", c_void_p), ("ldo2", c_int32),
        ("scale", c_float), ("act", c_int32), ("rs_bias_only", c_int32), ("n_split", c_int32), ("act2", c_int32),
        ("xrow", c_void_p), ("ldx", c_int32),
        ("reserve_cus", c_int32), ("probe", c_void_p), ("probe_cap", c_int32),
        ("wscale", c_void_p), ("aux_grad", c_int32), ("aux_frag", c_int32), ("row0", c_int32),
    ]


P = c_void_p
I = c_int
L = c_int64
F = c_float

# name -> argtypes; every entry point include/aim_kernels.h declares
SIGNATURES = {
    "aim_version": [],
    "aim_last_error": [],
    "aim_gemm_bf16": [POINTER(GemmArgs), I, I, P],
    "aim_gemm_fp8": [POINTER(GemmArgs), I, P],
    "aim_gemm_expsum_tiles": [I, I],
    "aim_wgrad_bf16": [P, I, P, I, P, I, P, I, I, I, P, L, P],
    "aim_wgrad_workspace_bytes": [I, I, I],
    "aim_wgrad_bias_bf16": [P, I, P, I, P, I, P, P, I, I, I, I, P, L, P],
    "aim_layernorm_fwd": [P, L, P, P, P, P, L, P, P, I, I, F, P],
    "aim_layernorm_fwd_fp8": [P, L, P, P, P, L, I, I, F, P],
    "aim_layernorm_fwd_x16": [P, L, P, P, P, P, P, L, I, I, F, P],
    "aim_layernorm_bwd": [P, I, L, P, L, P, P, P, P, I, P, P, L, P, P, I, I, P],
    "aim_layernorm_bwd_fsum": [P, L, P, L, P, P, P, P, P, L, P, P, I, I, I, I, P],
    "aim_attn_fwd": [P, P, P, I, I, I, P],
    "aim_attn_fwd_fp8": [P, P, P, I, I, I, P],
    "aim_attn_bwd": [P, P, P, P, P, P, I, I, I, P],
    "aim_cls_attn_fwd": [P, P, P, I, I, I, I, P],
    "aim_cls_attn_bwd": [P, P, P, P, I, I, I, I, I, P],
    "aim_tattn_fwd": [P, P, P, I, I, I, I, P],
    "aim_tattn_bwd": [P, P, P, P, I, I, I, I, P],
    "aim_add_bf16": [P, L, P, L, P, L, I, I, P],
    "aim_acc_bf16": [P, P, L, I, I, P],
    "aim_lambda_partials": [P, P, P, I, P],
    "aim_qk_cross": [P, P, I, P, I, I, I, F, P],
    "aim_lambda": [P, P, I, P, P, I, P, P, I, I, I, F, P],
    "aim_qk_border": [P, P, I, P, P, I, I, I, I, I, F, P],
    "aim_patchify": [P, I, P, P, P, I, I, I, I, I, I, P],
    "aim_embed_ln": [P, P, P, P, P, P, P, P, P, I, I, I, I, F, P],
    "aim_embed_bwd": [P, I, P, P, P, P, P, P, P, P, I, I, I, I, P, L, P],
    "aim_embed_bwd_workspace_bytes": [I, I, I, I],
    "aim_frame_sum": [P, I, P, P, I, I, I, P],
    "aim_colsum_bf16": [P, I, P, P, I, P, I, I, P, L, P],
    "aim_cast_bf16": [P, P, I, I, I, I, P],
    "aim_scale_rows": [P, P, P, P, I, I, P],
    "aim_add_rows_bf16": [P, L, P, I, I, P],
    "aim_adamw_flat": [P, P, P, P, L, F, F, F, F, F, I, F, P],
    "aim_head_fwd": [P, P, P, P, P, P, I, I, I, I, P],
    "aim_head_bwd": [P, P, P, P, P, P, P, I, I, I, I, P],
    "aim_ce_topk": [P, P, P, P, P, I, I, I, P],
    "aim_cast_multi": [P, I, P],
    "aim_gemm_f32": [POINTER(GemmArgs), I, I, P],
    "aim_attn_fwd_f32": [P, P, I, I, I, P],
    "aim_cls_attn_fwd_f32": [P, L, P, I, I, I, P],
    "aim_lambda_f32": [P, I, P, P, I, P, P, I, I, I, F, P],
    "aim_patchify_f32": [P, I, P, P, P, I, I, I, I, I, I, P],
    "aim_embed_ln_f32": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, P],
    "aim_attn_bwd_f32_workspace_bytes": [I, I, I],
    "aim_attn_bwd_f32": [P, P, P, I, I, I, P, L, P],
    "aim_cls_attn_bwd_f32": [P, L, P, P, I, I, I, P],
    "aim_tattn_fwd_f32": [P, P, I, I, I, I, P],
    "aim_tattn_bwd_f32": [P, P, P, I, I, I, I, P],
    "aim_wgrad_f32_workspace_bytes": [I, I, I],
    "aim_wgrad_f32": [P, I, P, I, P, I, I, I, P, P, I, P, L, P],
}

ABI_VERSION = 6


def load_library():
    """Load libaim_hip.so once; raise ``LibraryNotBuilt`` (never fall back) when it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise LibraryNotBuilt(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C adapt-image-models_amd/csrc`). There is no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = c_char_p if name == "aim_last_error" else (c_int64 if name.endswith("_bytes") else c_int)
    if lib.aim_version() != ABI_VERSION:
        raise LibraryNotBuilt(f"{path}: ABI version {lib.aim_version()} != {ABI_VERSION}; rebuild")
    _LIB = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load_library().aim_last_error()
        raise RuntimeError(f"libaim_hip {what} failed (rc={rc}): {msg.decode() if msg else '?'}")
