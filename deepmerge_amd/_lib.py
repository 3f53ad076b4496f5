"""ctypes binding of libdeepmerge_hip.so (the C-ABI declared in include/deepmerge_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  `lib()` raises
`DeepMergeLibraryError` when the shared object is missing or lacks a declared symbol, and every
wrapper raises `RuntimeError`/`ValueError` with `dm_last_error()` when a call returns non-zero.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "libdeepmerge_hip.so")
HEADER_PATH = os.path.join(ROOT, "include", "deepmerge_hip.h")

DM_F32, DM_BF16 = 0, 1
DM_NT, DM_NN, DM_TN = 0, 1, 2
DM_BF16_PAIR = 2      # DmGemmArgs.c_dtype: C as a hi / lo plane pair
DM_EPI_NONE, DM_EPI_GELU, DM_EPI_DGELU, DM_EPI_GELU_GRAD, DM_EPI_MUL = 0, 1, 2, 3, 4

_STATUS = {-1: "bad shape", -2: "bad dtype", -3: "bad alignment", -4: "workspace", -5: "HIP error", -6: "unsupported"}


class DeepMergeLibraryError(RuntimeError):
    pass


class DmGemmArgs(C.Structure):
    _fields_ = [
        ("layout", C.c_int32), ("ab_dtype", C.c_int32), ("c_dtype", C.c_int32), ("aux_dtype", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("epilogue", C.c_int32), ("accumulate", C.c_int32), ("split_k", C.c_int32),
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("B", C.c_void_p), ("ldb", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("aux", C.c_void_p), ("ldaux", C.c_int64),
        ("rows_per_group", C.c_int32), ("group_stride", C.c_int64),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("colsum_a", C.c_void_p), ("colsum_accumulate", C.c_int32),
        ("k_fold", C.c_int32), ("a_fold", C.c_int64 * 3), ("b_fold", C.c_int64 * 3),
        ("c_plane", C.c_int64),
    ]


class DmReduceItem(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("out0", C.c_void_p), ("out1", C.c_void_p),
                ("nrows", C.c_int32), ("width", C.c_int32), ("split", C.c_int32), ("accumulate", C.c_int32)]


class DmProfRow(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("total_flops", C.c_double), ("total_bytes", C.c_double)]


_P, _I, _L, _F, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double

# name -> (restype, argtypes); must list every function include/deepmerge_hip.h declares.
SIGNATURES = {
    "dm_abi_version": (_I, []),
    "dm_last_error": (C.c_char_p, []),
    "dm_arch": (C.c_char_p, []),
    "dm_gemm": (_I, [C.POINTER(DmGemmArgs), _P]),
    "dm_gemm_workspace_bytes": (_L, [_I, _I, _I, _I]),
    "dm_gemm_grouped": (_I, [C.POINTER(DmGemmArgs), _I, _P, _L, _P]),
    "dm_gemm_grouped_workspace_bytes": (_L, [C.POINTER(DmGemmArgs), _I]),
    "dm_attention_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    "dm_attention_relpos_inkernel": (_I, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "dm_attention_fwd_relpos": (_I, [_P, _P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    "dm_attention_split_ok": (_I, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "dm_attention_split_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _F, _P]),
    "dm_attention_split_fwd_pair": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "dm_attention_split_bwd_chunks": (_I, [_I, _I, _I]),
    "dm_attention_split_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "dm_attention_split_bwd_pair": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "dm_attention_bwd_relpos": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    "dm_attention_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    "dm_attention_bwd_batch_chunks": (_I, [_I, _I, _I, _I]),
    "dm_relpos_bias_gather": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "dm_relpos_bias_reduce": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dm_layernorm_fwd": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I, _F, _P]),
    "dm_layernorm_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P]),
    "dm_layernorm_bwd_partials": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, C.POINTER(C.c_int32), _P]),
    "dm_layernorm_bwd_partials_pair": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, C.POINTER(C.c_int32), _P]),
    "dm_partial_reduce_batch": (_I, [C.POINTER(DmReduceItem), _I, _P]),
    "dm_layernorm_bwd_partial_floats": (_L, [_I]),
    "dm_token_pool_fwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dm_token_pool_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dm_group_mean_fwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "dm_group_mean_bwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "dm_colsum": (_I, [_P, _I, _L, _P, _I, _I, _I, _P, _P]),
    "dm_colsum_partial_floats": (_L, [_I]),
    "dm_cast": (_I, [_P, _P, _I, _L, _P]),
    "dm_split_bf16": (_I, [_P, _L, _L, _L, _P, _I, _I, _P]),
    "dm_split_colsum_partial_floats": (_L, [_L, _L]),
    "dm_split_bf16_colsum": (_I, [_P, _L, _L, _L, _P, _I, _I, _P, _P, _P]),
    "dm_split_bf16_planes": (_I, [_P, _L, _L, _L, _P, _P, _P, _P]),
    "dm_patchify": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "dm_contrastive_loss": (_I, [_P, _P, _P, _F, _F, _P, _P, _P, _I, _I, _P]),
    "dm_cross_entropy": (_I, [_P, _P, _P, _F, _P, _P, _I, _I, _P]),
    "dm_batchnorm_workspace_bytes": (_L, [_I, _I]),
    "dm_batchnorm_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _F, _F, _I, _I, _P, _P]),
    "dm_batchnorm_bwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "dm_adam_step": (_I, [_P, _P, _P, _P, _P, _L, _I, _D, _D, _D, _D, _D, _P]),
    "dm_adam_hyper": (_I, [_I, _D, _D, _D, _P]),
    "dm_adam_step_dev": (_I, [_P, _P, _P, _P, _P, _L, _P, _D, _D, _D, _D, _P]),
    "dm_adam_step_dev_pair": (_I, [_P, _P, _P, _P, _P, _P, _L, _P, _D, _D, _D, _D, _P]),
    "dm_segment_mean": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "dm_edge_similarity": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "dm_patch_pyramid": (_I, [_P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P]),
    "dm_patch_pyramid_cols": (_I, [_P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "dm_pair_batch_gather": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P]),
    "dm_label_stats": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "dm_label_features": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "dm_rag_edges": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P, _P]),
    "dm_merge_round": (_I, [_P, _P, _I, _I, _P, _P, _I, _P]),
    "dm_gru_cell_fwd": (_I, [_P, _L, _P, _P, _P, _P, _I, _I, _P]),
    "dm_gru_cell_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "dm_prof_enable": (_I, [_I]),
    "dm_prof_collect": (_I, [C.POINTER(DmProfRow), _I]),
}

_lock = threading.Lock()
_lib = None


def declared_symbols(header_path: str = HEADER_PATH):
    """Function names declared in the public header (used by the ABI tests)."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dm_[a-z0-9_]+)\s*\(", text)))


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise loudly if it is not usable."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise DeepMergeLibraryError(
                f"{LIB_PATH} not found: build the HIP kernels first "
                f"(`python -c 'import __graft_entry__ as g; g.build()'` or `make -C deepmerge_amd/csrc`). "
                f"deepmerge_amd has no CPU fallback.")
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:
            raise DeepMergeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise DeepMergeLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        if handle.dm_abi_version() != 6:
            raise DeepMergeLibraryError(f"ABI version mismatch: library {handle.dm_abi_version()} != 6")
        _lib = handle
        return _lib


def check(rc: int, what: str):
    if rc == 0:
        return
    msg = lib().dm_last_error().decode("utf-8", "replace")
    text = f"{what} failed ({_STATUS.get(rc, rc)}): {msg}"
    if rc in (-1, -2, -3, -6):
        raise ValueError(text)
    raise RuntimeError(text)
