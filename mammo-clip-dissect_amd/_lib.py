"""ctypes binding of libmcd_hip.so (include/mcd_hip.h).  No fallback: a missing library is an error."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCD_LIB_PATH") or os.path.join(_HERE, "csrc", "libmcd_hip.so")   # MCD_LIB_PATH: dev builds of the same ABI

_i64 = ctypes.c_int64
_int = ctypes.c_int
_f = ctypes.c_float
_p = ctypes.c_void_p
_sz = ctypes.c_size_t

# name -> (restype, argtypes): exactly the symbols include/mcd_hip.h declares
SIGNATURES = {
    "mcd_last_error": (ctypes.c_char_p, []),
    "mcd_abi_version": (_int, []),
    "mcd_normalize_rows": (_int, [_p, _i64, _i64, _i64, _p, _i64, _p]),
    "mcd_center_cube_normalize_rows": (_int, [_p, _i64, _i64, _i64, _f, _p, _i64, _p]),
    "mcd_embed_gemm_workspace": (_sz, [_i64, _i64, _i64, _int]),
    "mcd_embed_gemm": (_int, [_p, _i64, _p, _i64, _i64, _i64, _i64, _int, _p, _i64, _p, _sz, _p]),
    "mcd_embed_gemm_exp_workspace": (_sz, [_i64, _i64, _i64]),
    "mcd_embed_gemm_exp": (_int, [_p, _i64, _p, _i64, _i64, _i64, _i64, _f, _int, _p, _i64, _p, _p, _sz, _p]),
    "mcd_embed_gemm_exp_time_kernel": (_int, [_int]),
    "mcd_embed_gemm_exp_kernel_ms": (_f, []),
    "mcd_wpmi_score_bf16_workspace": (_sz, [_i64, _int]),
    "mcd_wpmi_score_bf16": (_int, [_p, _i64, _i64, _i64, _p, _p, _i64, _i64, _int, _p, _f, _int, _p, _i64, _p, _sz, _p]),
    "mcd_row_softmax": (_int, [_p, _i64, _i64, _i64, _f, _p, _i64, _p]),
    "mcd_col_topk_workspace": (_sz, [_i64, _i64, _i64, _i64, _int]),
    "mcd_col_topk": (_int, [_p, _i64, _i64, _i64, _i64, _int, _p, _p, _i64, _p, _sz, _p]),
    "mcd_transpose": (_int, [_p, _i64, _i64, _i64, _p, _i64, _p]),
    "mcd_wpmi_score": (_int, [_p, _i64, _i64, _i64, _p, _i64, _i64, _int, _p, _f, _int, _int, _p, _i64, _p]),
    "mcd_logsumexp_sub_workspace": (_sz, [_i64, _i64, _int]),
    "mcd_logsumexp_sub": (_int, [_p, _i64, _i64, ctypes.POINTER(_i64), _int, _f, _int, _p, _i64, _p, _sz, _p]),
    "mcd_row_topk": (_int, [_p, _i64, _i64, _i64, _int, _p, _p, _p]),
    "mcd_hook_pool": (_int, [_p, _i64, _i64, _i64, _int, _p, _i64, _i64, _i64, _i64, _p]),
    "mcd_rank_reorder": (_int, [_p, _i64, _i64, _i64, _p, _p, _i64, _i64, _int, _p, _int, _f, _f, _p, _p, _i64, _p]),
    "mcd_vit_attention": (_int, [_p, _i64, _i64, _i64, _p, _p]),
    "mcd_layer_norm": (_int, [_p, _i64, _i64, _p, _p, _f, _p, _p]),
    "mcd_patchify": (_int, [_p, _i64, _i64, _i64, _i64, _i64, _p, _p]),
}

MCD_E_RANGE = -2

_lib = None


class McdError(RuntimeError):
    """An MCD_E_* status from libmcd_hip.so, with mcd_last_error() as the message."""

    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def load():
    """Load libmcd_hip.so once.  Raises ImportError (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "mammo-clip-dissect_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C mammo-clip-dissect_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = load().mcd_last_error().decode("utf-8", "replace")
        raise McdError(rc, msg)


# ---- libmcd_blaslt.so: the optional hipBLASLt companion (include/mcd_blaslt.h) ------------------------------
BLASLT_PATH = os.path.join(os.path.dirname(LIB_PATH), "libmcd_blaslt.so")
BLASLT_SIGNATURES = {
    "mcd_blaslt_last_error": (ctypes.c_char_p, []),
    "mcd_linear_residual_workspace": (_sz, []),
    "mcd_linear_residual": (_int, [_p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _p, _sz, _p]),
    "mcd_linear_residual_plan_info": (_int, [_i64, _i64, _i64, ctypes.POINTER(_f), ctypes.POINTER(_int)]),
    "mcd_linear_residual_get_picks": (_int, [ctypes.POINTER(_i64), _int]),
    "mcd_linear_residual_set_pick": (_int, [_i64, _i64, _i64, _int, _int]),
}
_blaslt = None


def load_blaslt():
    """libmcd_blaslt.so, or None when it has not been built / hipBLASLt cannot be loaded.  It only serves an encoder-side
    fusion (core.linear_residual); callers keep PyTorch's own linear + add without it."""
    global _blaslt
    if _blaslt is None:
        try:
            L = ctypes.CDLL(BLASLT_PATH)
            for name, (res, args) in BLASLT_SIGNATURES.items():
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
            _blaslt = L
        except (OSError, AttributeError):
            _blaslt = False
    return _blaslt or None
