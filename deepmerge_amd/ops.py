"""Torch-facing wrappers over the C-ABI (include/deepmerge_hip.h).

Two layers:
  * `raw_*` / plain functions: validate tensors, pass device pointers + the current HIP stream to the
    library.  PyTorch is only the allocator / stream provider here.
  * `*Fn` autograd Functions: the forward/backward pairs the reference gets from torch autograd for
    nn.Linear, nn.LayerNorm, the attention core, pooling and the loss (reference lines cited per
    class), built from the raw calls.

Numerics mode: "bf16" (throughput: bf16 operands/activations, fp32 accumulate, fp32 residual stream,
LayerNorm statistics, softmax, loss and master weights) or "fp32" (parity: fp32 everywhere on the
f32-input MFMA).  There is no CPU path: every function needs CUDA(HIP) tensors and the built library.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import DM_BF16, DM_EPI_DGELU, DM_EPI_GELU, DM_EPI_GELU_GRAD, DM_EPI_MUL, DM_EPI_NONE, DM_F32, DM_NN, DM_NT, DM_TN, DmGemmArgs, check

_NUMERICS = "bf16"


NUMERICS_MODES = ("bf16", "fp32", "bf16x3")
_FP32_PRODUCTS = "mfma_f32"     # how gemm() multiplies fp32 operands: "mfma_f32" (exact fmaf chains) or "bf16x3" (split-bf16)
_SPLIT_MIN_WORK = 1 << 24       # products below this many multiply-adds stay on the fp32 MFMA kernel
_PAIR_LN_MAX_COLS = 1024        # widest row dm_layernorm_fwd(DM_BF16_PAIR) / dm_layernorm_bwd_partials_pair take (csrc/dm_rows.hip, MAXCH)


def set_numerics(mode: str) -> None:
    """Default numerics mode for modules constructed afterwards ("bf16", "fp32" or "bf16x3")."""
    global _NUMERICS
    _NUMERICS = check_numerics(mode)


def get_numerics() -> str:
    return _NUMERICS


def check_numerics(mode: str) -> str:
    """Validate a module's numerics mode (no side effect).  "bf16x3" = the fp32 mode (fp32 activations, gradients, attention, row
    kernels) with its large products on the bf16 matrix pipe as split-bf16 triples (dm_split_bf16: ~2^-17 relative per product
    instead of bf16's 2^-9).  The product kind belongs to the MODULE: `bind_numerics` scopes every forward call of a "bf16x3"
    module, and each autograd node replays the kind its forward ran under in its backward (`ctx.products`)."""
    if mode not in NUMERICS_MODES:
        raise ValueError(f"numerics must be one of {NUMERICS_MODES}, got {mode!r}")
    return mode


def bind_numerics(module, mode: str) -> str:
    """`self.numerics = ops.bind_numerics(self, mode)`: validates, and for "bf16x3" makes every call of this module run inside
    `fp32_products("bf16x3")` (pre / post forward hooks; the post hook also runs when forward raises).  Modules of the other
    modes leave the ambient kind alone, so `with ops.fp32_products("bf16x3"):` around an fp32 module still works as sugar."""
    check_numerics(mode)
    if mode == "bf16x3" and module is not None:
        scopes = []

        def _enter(mod, args):
            sc = fp32_products("bf16x3")
            sc.__enter__()
            scopes.append(sc)

        def _exit(mod, args, out):
            if scopes:
                scopes.pop().__exit__(None, None, None)

        module.register_forward_pre_hook(_enter)
        module.register_forward_hook(_exit, always_call=True)
    return mode


def set_fp32_products(kind: str) -> None:
    """Ambient product kind for fp32 operands OUTSIDE any module scope ("mfma_f32" by default).  Prefer the module's `numerics`
    argument or the `fp32_products` context manager; nothing in this package calls this setter implicitly."""
    global _FP32_PRODUCTS
    if kind not in ("mfma_f32", "bf16x3"):
        raise ValueError(f"fp32 products are 'mfma_f32' or 'bf16x3', got {kind!r}")
    _FP32_PRODUCTS = kind


def get_fp32_products() -> str:
    return _FP32_PRODUCTS


class fp32_products:
    """with ops.fp32_products("bf16x3"): ...   scopes how gemm() multiplies fp32 operands.  Autograd nodes created inside the block
    remember the kind (`_save_products`) and their backward runs under it wherever it is called from."""

    def __init__(self, kind: str):
        if kind not in ("mfma_f32", "bf16x3"):
            raise ValueError(f"fp32 products are 'mfma_f32' or 'bf16x3', got {kind!r}")
        self.kind = kind

    def __enter__(self):
        global _FP32_PRODUCTS
        self.prev = _FP32_PRODUCTS
        _FP32_PRODUCTS = self.kind
        return self

    def __exit__(self, *exc):
        global _FP32_PRODUCTS
        _FP32_PRODUCTS = self.prev
        return False


def module_products(module):
    """The product scope of `module` as a context manager: for callers that enter a model through a method other than `__call__`
    (PairTrainer's `forward_pair_batched`), where the forward hooks of `bind_numerics` do not fire."""
    return fp32_products("bf16x3" if getattr(module, "numerics", None) == "bf16x3" else _FP32_PRODUCTS)


def _save_products(ctx):
    ctx.products = _FP32_PRODUCTS


def _replay_products(backward):
    """Decorator for an autograd Function's backward: run it under the product kind its forward recorded."""
    import functools

    @functools.wraps(backward)
    def wrapped(ctx, *grads):
        with fp32_products(getattr(ctx, "products", "mfma_f32")):
            return backward(ctx, *grads)
    return wrapped


def act_dtype(mode: str) -> torch.dtype:
    return torch.bfloat16 if mode == "bf16" else torch.float32


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return DM_F32
    if t.dtype == torch.bfloat16:
        return DM_BF16
    raise ValueError(f"unsupported dtype {t.dtype}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("deepmerge_amd ops need tensors on the GPU (there is no CPU fallback)")


# ------------------------------------------------------------------------------------------------
# workspace (grow-only scratch per device; stream-ordered reuse on the current stream)
# ------------------------------------------------------------------------------------------------
_ws = {}
_ws_retired = []     # replaced workspaces stay allocated: a captured hipGraph (PairTrainer.enable_graph) has their raw addresses
                     # baked into its kernel arguments, so handing the memory back to the caching allocator would let a later
                     # tensor alias the split-K slabs / LayerNorm partials of every replay.  Growth is geometric, so the
                     # retired buffers of a slot sum to less than its live one.


def workspace(nbytes: int, device, slot: str = "main") -> torch.Tensor:
    key = (torch.device(device).index, slot)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        want = max(int(nbytes), 1 << 20, 2 * buf.numel() if buf is not None else 0)
        if buf is not None:
            _ws_retired.append(buf)
        buf = torch.empty(want, dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


# ------------------------------------------------------------------------------------------------
# raw calls
# ------------------------------------------------------------------------------------------------
class Planes:
    """hi / lo plane pair of an fp32 matrix (dm_split_bf16_planes): `t` is bf16 [2, rows, cols], the operand format of the folded
    "bf16x3" products (DmGemmArgs.k_fold).  One split serves every product that reads the matrix, on either side and in any layout."""
    __slots__ = ("t", "rows", "cols")

    def __init__(self, t: torch.Tensor):
        self.t, self.rows, self.cols = t, t.shape[1], t.shape[2]


_PLANES = os.environ.get("DM_X3_PLANES", "1") != "0"      # A/B aid: 0 = the three-piece images of dm_split_bf16 everywhere


def planes_ok(rows: int, cols: int) -> bool:
    """Can a [rows, cols] fp32 matrix be a folded operand on either side?  (contraction along either extent: a multiple of the K stage;
    16-byte row pieces; plane offset inside the kernels' 32-bit range)"""
    return _PLANES and rows % 64 == 0 and cols % 64 == 0 and rows * cols < (1 << 30)


def split_planes(x: torch.Tensor, colsum_out: Optional[torch.Tensor] = None, colsum_accumulate: bool = False, defer: bool = False) -> Planes:
    """x fp32 [rows, cols] (row stride % 4 == 0) -> Planes; with colsum_out [cols] (+)= the column sums of x on the side (the bias
    gradient that goes with a weight gradient: x = dy is read once for both).  defer=True (inside a backward pass, colsum_out a
    gradient sink nobody reads before the pass ends): the reduction of the partial rows joins the end-of-backward batch
    (`flush_reductions`) instead of being a launch of its own."""
    _need_cuda(x, colsum_out)
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("split_planes takes a 2-D fp32 matrix with contiguous rows")
    rows, cols = x.shape
    out = torch.empty((2, rows, cols), dtype=torch.bfloat16, device=x.device)
    if colsum_out is None:
        check(_lib.lib().dm_split_bf16_planes(x.data_ptr(), x.stride(0), rows, cols, out.data_ptr(), None, None, _stream()), "dm_split_bf16_planes")
        return Planes(out)
    n_part = _lib.lib().dm_split_colsum_partial_floats(rows, cols)
    deferred = defer and _DEFER_REDUCTIONS and _in_backward()
    part = (torch.empty(n_part, dtype=torch.float32, device=x.device) if deferred      # (alive until the batch launch: not a shared slot)
            else workspace(4 * n_part, x.device, "planes.partial").view(torch.float32))
    rows_out = C.c_int32(0)
    check(_lib.lib().dm_split_bf16_planes(x.data_ptr(), x.stride(0), rows, cols, out.data_ptr(), part.data_ptr(), C.byref(rows_out), _stream()),
          "dm_split_bf16_planes")
    if deferred:
        _queue_reduce(part, colsum_out, colsum_out, rows_out.value, cols, cols, colsum_accumulate)
        return Planes(out)
    item = (_lib.DmReduceItem * 1)()
    item[0].partial, item[0].out0, item[0].out1 = part.data_ptr(), colsum_out.data_ptr(), colsum_out.data_ptr()
    item[0].nrows, item[0].width, item[0].split, item[0].accumulate = rows_out.value, cols, cols, int(bool(colsum_accumulate))
    check(_lib.lib().dm_partial_reduce_batch(item, 1, _stream()), "dm_partial_reduce_batch")
    return Planes(out)


def gemm(layout: int, A: torch.Tensor, B: torch.Tensor, C_out: torch.Tensor, M: int, N: int, K: int, *,
         lda: Optional[int] = None, ldb: Optional[int] = None, ldc: Optional[int] = None,
         bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, ldr: Optional[int] = None,
         epilogue: int = DM_EPI_NONE, aux: Optional[torch.Tensor] = None, ldaux: Optional[int] = None,
         accumulate: bool = False, split_k: int = 0, rows_per_group: int = 0, group_stride: int = 0,
         colsum_out: Optional[torch.Tensor] = None, colsum_accumulate: bool = False, ws_slot: str = "gemm") -> torch.Tensor:
    """dm_gemm.  A/B/C are 2-D (or flat) row-major tensors; leading dims default to their last-dim size.
    DM_TN only: colsum_out [M] fp32 (+)= column sums of A (the bias gradient that goes with dW = dy^T x)."""
    c_pair = C_out if isinstance(C_out, Planes) else None      # the result as a hi / lo plane pair (DM_BF16_PAIR): no split pass for its consumers
    if c_pair is not None:
        if (c_pair.rows, c_pair.cols) != (M, N) or ldc not in (None, N):
            raise ValueError("a plane-pair result is a dense [2, M, N] tensor")
        C_out = c_pair.t
    fold = None
    if isinstance(A, Planes) or isinstance(B, Planes) or (
            _FP32_PRODUCTS == "bf16x3" and A.dtype == torch.float32 and M * N * K >= _SPLIT_MIN_WORK and planes_ok(M, K) and planes_ok(N, K)
            and (lda is None or lda % 4 == 0) and (ldb is None or ldb % 4 == 0) and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0):
        # folded split-bf16 product: both operands as hi / lo plane pairs (split here unless the caller already holds the pair),
        # one bf16 product over three K segments (hi.hi + hi.lo + lo.hi) that re-read the planes in place
        a_rows, a_cols = (M, K) if layout != DM_TN else (K, M)
        b_rows, b_cols = (N, K) if layout == DM_NT else (K, N)
        if not isinstance(A, Planes):
            A = split_planes(torch.as_strided(A, (a_rows, a_cols), (lda if lda is not None else a_cols, 1)), colsum_out, colsum_accumulate)
            colsum_out = None
        if not isinstance(B, Planes):
            B = split_planes(torch.as_strided(B, (b_rows, b_cols), (ldb if ldb is not None else b_cols, 1)))
        if (A.rows, A.cols) != (a_rows, a_cols) or (B.rows, B.cols) != (b_rows, b_cols):
            raise ValueError(f"plane pair shapes {(A.rows, A.cols)} / {(B.rows, B.cols)} do not match the product {(a_rows, a_cols)} / {(b_rows, b_cols)}")
        # (column sums of a pre-split left operand: dm_gemm sums the hi and the lo plane -- fused into the weight-gradient kernels)
        fold = (K, A.t.stride(0), B.t.stride(0))      # (plane stride: rows * cols for a split, the flat buffer's length for a weight's mirror pair)
        A, B, lda, ldb, K = A.t, B.t, a_cols, b_cols, 3 * K
    _need_cuda(A, B, C_out, bias, residual, aux, colsum_out)
    if A.dtype != B.dtype:
        raise ValueError(f"A/B dtype mismatch: {A.dtype} vs {B.dtype}")
    if lda is None:
        lda = A.stride(-2) if A.dim() >= 2 else K
    if ldb is None:
        ldb = B.stride(-2) if B.dim() >= 2 else K
    if (_FP32_PRODUCTS == "bf16x3" and A.dtype == torch.float32 and M * N * K >= _SPLIT_MIN_WORK
            and M % 8 == 0 and N % 8 == 0 and K % 8 == 0 and lda % 4 == 0 and ldb % 4 == 0):     # 16-byte rows of the bf16 images
        # split-bf16: one bf16 product over a 3x longer contraction, [Ah | Ah | Al] . [Bh | Bl | Bh] (include/deepmerge_hip.h)
        a_rows, a_cols = (M, K) if layout != DM_TN else (K, M)          # the operands as row-major matrices
        b_rows, b_cols = (N, K) if layout == DM_NT else (K, N)
        a_stack, b_stack = int(layout == DM_TN), int(layout != DM_NT)
        A3 = workspace(6 * a_rows * a_cols, A.device, ws_slot + ".split_a").view(torch.bfloat16)    # per slot: the side stream
        B3 = workspace(6 * b_rows * b_cols, A.device, ws_slot + ".split_b").view(torch.bfloat16)    # has its own images
        # the fused split + column-sum pass has the vector kernel's preconditions (dm_rows.hip split_rows_ok: 16-byte aligned
        # source, ld % 4 == 0 -- required above -- and whole 8-column groups); an offset view of dy takes the two-pass route below
        fused_ok = colsum_out is not None and a_cols % 8 == 0 and A.data_ptr() % 16 == 0 and (a_rows * a_cols) % 8 == 0
        n_part = _lib.lib().dm_split_colsum_partial_floats(a_rows, a_cols) if fused_ok else 0
        if n_part > 0:                       # the bias gradient rides on the operand's split pass: A is read once for both
            part = workspace(4 * n_part, A.device, ws_slot + ".partial").view(torch.float32)
            rows_out = C.c_int32(0)
            check(_lib.lib().dm_split_bf16_colsum(A.data_ptr(), lda, a_rows, a_cols, A3.data_ptr(), a_stack, 0b100, part.data_ptr(),
                                                  C.byref(rows_out), _stream()), "dm_split_bf16_colsum")
            item = (_lib.DmReduceItem * 1)()
            item[0].partial, item[0].out0, item[0].out1 = part.data_ptr(), colsum_out.data_ptr(), colsum_out.data_ptr()
            item[0].nrows, item[0].width, item[0].split, item[0].accumulate = rows_out.value, a_cols, a_cols, int(bool(colsum_accumulate))
            check(_lib.lib().dm_partial_reduce_batch(item, 1, _stream()), "dm_partial_reduce_batch")
        else:
            check(_lib.lib().dm_split_bf16(A.data_ptr(), lda, a_rows, a_cols, A3.data_ptr(), a_stack, 0b100, _stream()), "dm_split_bf16")
            if colsum_out is not None:       # column sums of the fp32 operand itself (the stacked image would count hi twice)
                colsum(torch.as_strided(A, (a_rows, a_cols), (lda, 1)), colsum_out, accumulate=colsum_accumulate, ws_slot=ws_slot + ".partial")
        check(_lib.lib().dm_split_bf16(B.data_ptr(), ldb, b_rows, b_cols, B3.data_ptr(), b_stack, 0b010, _stream()), "dm_split_bf16")
        return gemm(layout, A3, B3, C_out, M, N, 3 * K, lda=(a_cols if a_stack else 3 * a_cols), ldb=(b_cols if b_stack else 3 * b_cols),
                    ldc=ldc, bias=bias, residual=residual, ldr=ldr, epilogue=epilogue, aux=aux, ldaux=ldaux, accumulate=accumulate,
                    split_k=split_k, rows_per_group=rows_per_group, group_stride=group_stride, ws_slot=ws_slot)
    a = _gemm_args(layout, A, B, C_out, M, N, K, lda, ldb, ldc, bias, residual, ldr, epilogue, aux, ldaux, accumulate, split_k,
                   rows_per_group, group_stride, colsum_out, colsum_accumulate, ws_slot, c_pair, fold)
    check(_lib.lib().dm_gemm(C.byref(a), _stream()), "dm_gemm")
    return c_pair if c_pair is not None else C_out


def _gemm_args(layout, A, B, C_out, M, N, K, lda, ldb, ldc, bias, residual, ldr, epilogue, aux, ldaux, accumulate, split_k,
               rows_per_group, group_stride, colsum_out, colsum_accumulate, ws_slot, c_pair=None, fold=None, min_slabs: int = 0) -> DmGemmArgs:
    """The DmGemmArgs of one product (operands already in their final form), its workspace taken from slot `ws_slot`."""
    a = DmGemmArgs()
    a.layout, a.ab_dtype, a.c_dtype = layout, _dt(A), _dt(C_out)
    if c_pair is not None:
        if _dt(A) != DM_BF16:
            raise ValueError("a plane-pair result goes with bf16 / folded operands")
        a.c_dtype, a.c_plane = _lib.DM_BF16_PAIR, M * N
    a.aux_dtype = _dt(aux) if aux is not None else DM_F32
    a.M, a.N, a.K = M, N, K
    a.epilogue, a.accumulate, a.split_k = epilogue, int(accumulate), split_k
    a.A, a.lda = A.data_ptr(), lda
    a.B, a.ldb = B.data_ptr(), ldb
    a.C, a.ldc = C_out.data_ptr(), (ldc if ldc is not None else N)
    a.bias = _ptr(bias)
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() < N):
        raise ValueError("bias must be fp32 with >= N elements")
    a.residual, a.ldr = _ptr(residual), (ldr if ldr is not None else N)
    if residual is not None and residual.dtype != torch.float32:
        raise ValueError("residual must be fp32")
    a.aux, a.ldaux = _ptr(aux), (ldaux if ldaux is not None else N)
    a.rows_per_group, a.group_stride = rows_per_group, group_stride
    if colsum_out is not None:
        if colsum_out.dtype != torch.float32 or colsum_out.numel() < M or not colsum_out.is_contiguous():
            raise ValueError("colsum_out must be a contiguous fp32 tensor with >= M elements")
        a.colsum_a, a.colsum_accumulate = colsum_out.data_ptr(), int(colsum_accumulate)
    if fold is not None:
        a.k_fold = fold[0]
        a.a_fold[0], a.a_fold[1], a.a_fold[2] = 0, 0, fold[1]           # left operand: hi, hi, lo
        a.b_fold[0], a.b_fold[1], a.b_fold[2] = 0, fold[2], 0           # right operand: hi, lo, hi
    ws_bytes = _lib.lib().dm_gemm_workspace_bytes(layout, M, N, K) if (split_k != 1 or colsum_out is not None) else 0
    if split_k > 1:                       # caller-chosen slice count: its slab may be larger than the automatic one
        ws_bytes = max(ws_bytes, _lib.lib().dm_gemm_workspace_bytes(layout, M, N, K) + split_k * M * N * 4)
    if min_slabs > 0:                     # a grouped launch may slice the product more finely than dm_gemm alone would
        ws_bytes = max(ws_bytes, _lib.lib().dm_gemm_workspace_bytes(layout, M, N, K) + min_slabs * M * N * 4)
    if ws_bytes > 0:
        ws = workspace(ws_bytes, A.device, ws_slot)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    return a


def gemm_grouped(calls) -> None:
    """dm_gemm_grouped: `calls` is a list of (args, kwargs) as for gemm() -- INDEPENDENT products (the weight gradients of a block),
    issued as one library call that may run them in one launch.  Products the grouped entry does not describe (plane pairs, the
    split-bf16 forms of fp32 operands) make the whole list run as separate gemm() calls."""
    if not calls:
        return
    # the forms the entry describes: plain bf16 operands, or BOTH operands as hi / lo plane pairs (the folded "bf16x3" product)
    plain = len(calls) > 1
    for args, kw in calls:
        A, B = args[1], args[2]
        pair = isinstance(A, Planes) and isinstance(B, Planes)
        plain = plain and not isinstance(args[3], Planes) and (pair or (not isinstance(A, Planes) and not isinstance(B, Planes) and A.dtype == torch.bfloat16))
    if not plain:
        for args, kw in calls:
            gemm(*args, **kw)
        return
    arr = (DmGemmArgs * len(calls))()
    for i, (args, kw) in enumerate(calls):
        layout, A, B, C_out, M, N, K = args
        lda, ldb, fold = kw.get("lda"), kw.get("ldb"), None
        if isinstance(A, Planes):            # as in gemm(): three K segments that re-read the planes in place
            a_rows, a_cols = (M, K) if layout != DM_TN else (K, M)
            b_rows, b_cols = (N, K) if layout == DM_NT else (K, N)
            if (A.rows, A.cols) != (a_rows, a_cols) or (B.rows, B.cols) != (b_rows, b_cols):
                raise ValueError(f"plane pair shapes {(A.rows, A.cols)} / {(B.rows, B.cols)} do not match the product {(a_rows, a_cols)} / {(b_rows, b_cols)}")
            fold = (K, A.t.stride(0), B.t.stride(0))
            A, B, lda, ldb, K = A.t, B.t, a_cols, b_cols, 3 * K
        _need_cuda(A, B, C_out, kw.get("colsum_out"))
        if A.dtype != B.dtype:
            raise ValueError(f"A/B dtype mismatch: {A.dtype} vs {B.dtype}")
        lda = lda if lda is not None else (A.stride(-2) if A.dim() >= 2 else K)
        ldb = ldb if ldb is not None else (B.stride(-2) if B.dim() >= 2 else K)
        arr[i] = _gemm_args(layout, A, B, C_out, M, N, K, lda, ldb, kw.get("ldc"), kw.get("bias"), kw.get("residual"), kw.get("ldr"),
                            kw.get("epilogue", DM_EPI_NONE), kw.get("aux"), kw.get("ldaux"), kw.get("accumulate", False), kw.get("split_k", 0),
                            kw.get("rows_per_group", 0), kw.get("group_stride", 0), kw.get("colsum_out"), kw.get("colsum_accumulate", False),
                            f"{kw.get('ws_slot', 'gemm')}.g{i}", None, fold, min_slabs=16 if layout == DM_TN else 0)
    ws_bytes = _lib.lib().dm_gemm_grouped_workspace_bytes(arr, len(calls))
    ws = workspace(ws_bytes, calls[0][0][1].t.device if isinstance(calls[0][0][1], Planes) else calls[0][0][1].device, "gemm_grouped") if ws_bytes > 0 else None
    check(_lib.lib().dm_gemm_grouped(arr, len(calls), None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), _stream()), "dm_gemm_grouped")


def cast(src: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """fp32 -> activation dtype copy through dm_cast (identity object if already that dtype)."""
    if src.dtype == dtype:
        return src
    _need_cuda(src)
    if src.dtype != torch.float32:
        raise ValueError("dm_cast source must be fp32")
    src = src.contiguous()
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    check(_lib.lib().dm_cast(src.data_ptr(), dst.data_ptr(), _dt(dst), src.numel(), _stream()), "dm_cast")
    return dst


def colsum(X: torch.Tensor, out: torch.Tensor, accumulate: bool = False, ws_slot: str = "partial") -> torch.Tensor:
    _need_cuda(X, out)
    M, N = X.shape
    part = workspace(_lib.lib().dm_colsum_partial_floats(N) * 4, X.device, ws_slot)
    check(_lib.lib().dm_colsum(X.data_ptr(), _dt(X), X.stride(0), out.data_ptr(), M, N, int(accumulate), part.data_ptr(), _stream()), "dm_colsum")
    return out


def layernorm_fwd(x: torch.Tensor, gamma, beta, eps: float, out_dtype: torch.dtype, pair: bool = False):
    """(y, mean, rstd); pair=True: y is the Planes (hi / lo plane pair) of the normalised rows -- written by the kernel itself."""
    _need_cuda(x, gamma, beta)
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    if pair:
        yp = torch.empty((2, rows, cols), dtype=torch.bfloat16, device=x.device)
        check(_lib.lib().dm_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), yp.data_ptr(), _lib.DM_BF16_PAIR, mean.data_ptr(),
                                          rstd.data_ptr(), rows, cols, eps, _stream()), "dm_layernorm_fwd")
        return Planes(yp), mean, rstd
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    check(_lib.lib().dm_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _dt(y), mean.data_ptr(),
                                      rstd.data_ptr(), rows, cols, eps, _stream()), "dm_layernorm_fwd")
    return y, mean, rstd


# ---- deferred partial-row reductions ------------------------------------------------------------------------------------------------
# The LayerNorm backward leaves one [dgamma | dbeta] partial row per workgroup; reducing them is a ~5 us launch of 96 workgroups per
# LayerNorm application (16 per step of the headline model).  When the results go straight into gradient sinks (nothing reads them
# before the backward pass ends), the reductions are queued and done by ONE dm_partial_reduce_batch launch from an end-of-backward
# callback of the autograd engine -- inside a captured step that launch is captured like any other.  Same arithmetic, same bits.
_pending_reduce = []        # (DmReduceItem field tuple, tensors kept alive until the launch)
_pending_out = set()
_flush_queued = False
_flush_task = -1            # autograd graph task the queue belongs to (a backward pass that raised leaves its queue behind: dropped on sight)


def _in_backward() -> bool:
    return torch._C._current_graph_task_id() != -1


def flush_reductions() -> None:
    """Launch the queued reductions (no-op when nothing is pending).  Called by the autograd engine at the end of a backward pass;
    call it yourself before reading a deferred result outside one."""
    global _flush_queued
    _flush_queued = False
    if not _pending_reduce:
        return
    items = (_lib.DmReduceItem * len(_pending_reduce))()
    for i, (f, _keep) in enumerate(_pending_reduce):
        items[i].partial, items[i].out0, items[i].out1, items[i].nrows, items[i].width, items[i].split, items[i].accumulate = f
    n = len(_pending_reduce)
    keep = list(_pending_reduce)
    _pending_reduce.clear()
    _pending_out.clear()
    check(_lib.lib().dm_partial_reduce_batch(items, n, _stream()), "dm_partial_reduce_batch")
    del keep


def _queue_reduce(part, out0, out1, nrows, width, split, accumulate) -> None:
    global _flush_queued, _flush_task
    task = torch._C._current_graph_task_id()
    if task != _flush_task:                                   # leftovers of a backward pass that never reached its callback
        _pending_reduce.clear()
        _pending_out.clear()
        _flush_queued = False
        _flush_task = task
    keys = (out0.data_ptr(), out1.data_ptr())
    if keys[0] in _pending_out or keys[1] in _pending_out:      # a second contribution to the same parameter (a shared norm): keep the order
        flush_reductions()
    _pending_reduce.append(((part.data_ptr(), keys[0], keys[1], int(nrows), int(width), int(split), int(bool(accumulate))), (part, out0, out1)))
    _pending_out.update(keys)
    if not _flush_queued:
        torch.autograd.Variable._execution_engine.queue_callback(flush_reductions)
        _flush_queued = True


def layernorm_bwd(dy, x, gamma, mean, rstd, dres=None, dgamma=None, dbeta=None, accumulate=False, want_lp=False, defer=False, want_pair=False):
    """Returns (dx, dgamma, dbeta) or, with want_lp, (dx, dx_bf16, dgamma, dbeta); with want_pair, (dx, Planes(dx), dgamma, dbeta).
    defer=True (only with caller-provided dgamma / dbeta, inside a backward pass): their reduction is queued for the end of the pass
    (`flush_reductions`); dx is complete on return."""
    _need_cuda(dy, x)
    if not (dy.is_contiguous() and x.is_contiguous()) or dy.numel() != x.numel():
        raise ValueError("layernorm_bwd: dy and x must be contiguous and of the same size (an expanded gradient has no rows to read)")
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    dx_lp = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_lp else None
    if dgamma is None:
        dgamma = torch.empty(cols, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(cols, dtype=torch.float32, device=x.device)
        accumulate = False
        defer = False
    if want_pair:
        pair = torch.empty((2, rows, cols), dtype=torch.bfloat16, device=x.device)
        deferred = defer and _DEFER_REDUCTIONS and _in_backward()
        part = (torch.empty(_lib.lib().dm_layernorm_bwd_partial_floats(cols), dtype=torch.float32, device=x.device) if deferred
                else workspace(_lib.lib().dm_layernorm_bwd_partial_floats(cols) * 4, x.device, "partial").view(torch.float32))
        n_part = C.c_int32(0)
        check(_lib.lib().dm_layernorm_bwd_partials_pair(dy.data_ptr(), _dt(dy), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                        _ptr(dres), dx.data_ptr(), pair.data_ptr(), part.data_ptr(), rows, cols, C.byref(n_part),
                                                        _stream()), "dm_layernorm_bwd_partials_pair")
        if deferred:
            _queue_reduce(part, dgamma, dbeta, n_part.value, 2 * cols, cols, accumulate)
        else:
            item = (_lib.DmReduceItem * 1)()
            item[0].partial, item[0].out0, item[0].out1 = part.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr()
            item[0].nrows, item[0].width, item[0].split, item[0].accumulate = n_part.value, 2 * cols, cols, int(bool(accumulate))
            check(_lib.lib().dm_partial_reduce_batch(item, 1, _stream()), "dm_partial_reduce_batch")
        return dx, Planes(pair), dgamma, dbeta
    if defer and _DEFER_REDUCTIONS and _in_backward():
        part = torch.empty(_lib.lib().dm_layernorm_bwd_partial_floats(cols), dtype=torch.float32, device=x.device)
        n_part = C.c_int32(0)
        check(_lib.lib().dm_layernorm_bwd_partials(dy.data_ptr(), _dt(dy), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                   _ptr(dres), dx.data_ptr(), _ptr(dx_lp), part.data_ptr(), rows, cols, C.byref(n_part),
                                                   _stream()), "dm_layernorm_bwd_partials")
        _queue_reduce(part, dgamma, dbeta, n_part.value, 2 * cols, cols, accumulate)
        return (dx, dx_lp, dgamma, dbeta) if want_lp else (dx, dgamma, dbeta)
    part = workspace(_lib.lib().dm_layernorm_bwd_partial_floats(cols) * 4, x.device, "partial")
    check(_lib.lib().dm_layernorm_bwd(dy.data_ptr(), _dt(dy), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                      _ptr(dres), dx.data_ptr(), _ptr(dx_lp), dgamma.data_ptr(), dbeta.data_ptr(), int(accumulate),
                                      part.data_ptr(), rows, cols, _stream()), "dm_layernorm_bwd")
    return (dx, dx_lp, dgamma, dbeta) if want_lp else (dx, dgamma, dbeta)


_DEFER_REDUCTIONS = os.environ.get("DM_DEFER_REDUCTIONS", "1") != "0"      # A/B switch


def relpos_bias_gather(table: torch.Tensor, index32: torch.Tensor, N: int, transposed: bool = False):
    """Dense bias [H,N,N]; with transposed=True also its per-head transpose -> (bias, bias_t)."""
    _need_cuda(table, index32)
    n_bins, H = table.shape
    bias = torch.empty((H, N, N), dtype=torch.float32, device=table.device)
    bias_t = torch.empty_like(bias) if transposed else None
    check(_lib.lib().dm_relpos_bias_gather(table.data_ptr(), index32.data_ptr(), bias.data_ptr(), _ptr(bias_t), N, H, n_bins, _stream()),
          "dm_relpos_bias_gather")
    return (bias, bias_t) if transposed else bias


def attention_fwd(qkv: torch.Tensor, bias: Optional[torch.Tensor], B: int, N: int, H: int, D: int, scale: float):
    _need_cuda(qkv, bias)
    out = torch.empty((B, N, H * D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    check(_lib.lib().dm_attention_fwd(qkv.data_ptr(), _ptr(bias), out.data_ptr(), lse.data_ptr(), B, N, H, D, scale, _dt(qkv), _stream()),
          "dm_attention_fwd")
    return out, lse


def _split_attention(table, index32, qkv, B, N, H, D):
    """(use, cube): whether the forward runs on the split-bf16 attention kernels -- fp32 tensors inside a "bf16x3" scope, a shape they
    take, and a bias that is either absent or the table of a token cube the module vouches for.  `qkv`: the tensor or its dtype."""
    if (qkv if isinstance(qkv, torch.dtype) else qkv.dtype) != torch.float32 or _FP32_PRODUCTS != "bf16x3" or not _SPLIT_ATTENTION:
        return False, None
    cube = None if table is None else getattr(index32, "_dm_cube", None)
    if table is not None and cube is None:
        return False, None
    return attention_split_ok(B, N, H, D, cube), cube


_SPLIT_ATTENTION = os.environ.get("DM_ATTN_X3", "1") != "0"


def _inkernel_cube(table, index32, B, N, H, D, dtype):
    """The token cube if the forward kernel can form the bias from `table` itself (the module vouches for the index: `_dm_cube` on the
    int32 index tensor is set only after comparing it with the closed form), else None."""
    cube = None if table is None else getattr(index32, "_dm_cube", None)
    if cube is None or _INKERNEL_TABLE is False or not relpos_inkernel(B, N, H, D, cube, dtype):
        return None
    return cube


_INKERNEL_TABLE = os.environ.get("DM_ATTN_TABLE_IN_KERNEL", "1") != "0"      # A/B switch
# the backward kernels read the table themselves too; the dense rows are gathered only when those kernels are switched off (A/B runs)
_DENSE_BWD_ROWS = os.environ.get("DM_ATTN_Q32_TABKV", "1") == "0" or os.environ.get("DM_ATTN_Q32_BWD", "1") in ("0", "3")


def relpos_inkernel(B: int, N: int, H: int, D: int, cube, dtype: torch.dtype) -> bool:
    """True where dm_attention_fwd_relpos takes the table itself (no dense bias rows) for a (scales, rows, cols) token cube."""
    if cube is None or len(cube) != 3:
        return False
    return bool(_lib.lib().dm_attention_relpos_inkernel(B, N, H, D, int(cube[0]), int(cube[1]), int(cube[2]),
                                                        DM_BF16 if dtype == torch.bfloat16 else DM_F32))


def attention_fwd_relpos(qkv: torch.Tensor, table: torch.Tensor, cube, B: int, N: int, H: int, D: int, scale: float):
    """attention_fwd with bias = table[relative_position_index(cube)], formed inside the kernel (table [n_bins, H] fp32)."""
    _need_cuda(qkv, table)
    if table.dtype != torch.float32 or not table.is_contiguous() or tuple(table.shape) != ((2 * cube[0] - 1) * (2 * cube[1] - 1) * (2 * cube[2] - 1), H):
        raise ValueError(f"attention_fwd_relpos: table must be contiguous fp32 [(2s-1)(2h-1)(2w-1), H], got {tuple(table.shape)} {table.dtype}")
    out = torch.empty((B, N, H * D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    check(_lib.lib().dm_attention_fwd_relpos(qkv.data_ptr(), table.data_ptr(), int(cube[0]), int(cube[1]), int(cube[2]), out.data_ptr(),
                                             lse.data_ptr(), B, N, H, D, scale, _dt(qkv), _stream()), "dm_attention_fwd_relpos")
    return out, lse


def attention_split_ok(B: int, N: int, H: int, D: int, cube=None) -> bool:
    """True where the split-bf16 attention kernels (the "bf16x3" mode's) take the shape; cube = token cube of the bias table or None."""
    c = (0, 0, 0) if cube is None else tuple(int(v) for v in cube)
    return bool(_lib.lib().dm_attention_split_ok(B, N, H, D, int(cube is not None), c[0], c[1], c[2]))


def attention_fwd_split(qkv: torch.Tensor, table: Optional[torch.Tensor], cube, B: int, N: int, H: int, D: int, scale: float, out_pair: bool = False):
    """fp32 attention with split-bf16 products.  Returns (out, lse, qkv_hi, qkv_lo); the two bf16 images go to the backward pass.
    qkv: the fp32 tensor, or its Planes (the qkv product wrote the pair itself: its planes ARE the two images)."""
    if table is not None and (table.dtype != torch.float32 or not table.is_contiguous()):
        raise ValueError("attention_fwd_split: table must be contiguous fp32")
    if isinstance(qkv, Planes):
        _need_cuda(qkv.t, table)
        hi, lo, src, dev = qkv.t[0], qkv.t[1], None, qkv.t.device
    else:
        _need_cuda(qkv, table)
        if qkv.dtype != torch.float32 or not qkv.is_contiguous():
            raise ValueError("attention_fwd_split: qkv must be a contiguous fp32 tensor")
        hi = torch.empty(qkv.shape, dtype=torch.bfloat16, device=qkv.device)
        lo = torch.empty(qkv.shape, dtype=torch.bfloat16, device=qkv.device)
        src, dev = qkv.data_ptr(), qkv.device
    out = torch.empty((B, N, H * D), dtype=torch.float32, device=dev)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
    c = (0, 0, 0) if cube is None else tuple(int(v) for v in cube)
    if out_pair:        # (out, lse, hi, lo, Planes(out as [B*N, H*D])): the kernel writes the pair next to the fp32 result
        op = torch.empty((2, B * N, H * D), dtype=torch.bfloat16, device=dev)
        check(_lib.lib().dm_attention_split_fwd_pair(src, hi.data_ptr(), lo.data_ptr(), _ptr(table), c[0], c[1], c[2], out.data_ptr(), op.data_ptr(),
                                                     lse.data_ptr(), B, N, H, D, scale, _stream()), "dm_attention_split_fwd_pair")
        return out, lse, hi, lo, Planes(op)
    check(_lib.lib().dm_attention_split_fwd(src, hi.data_ptr(), lo.data_ptr(), _ptr(table), c[0], c[1], c[2], out.data_ptr(),
                                            lse.data_ptr(), B, N, H, D, scale, _stream()), "dm_attention_split_fwd")
    return out, lse, hi, lo


def attention_bwd_split(hi, lo, table, cube, out, dout, lse, B, N, H, D, scale, index32=None, n_bins=0, pair=False):
    """Backward of attention_fwd_split.  Returns (dqkv fp32, dbias_slab or None, info) like attention_bwd; pair=True: dqkv as the
    Planes of the [B*N, 3*H*D] matrix (written by the kernels themselves)."""
    _need_cuda(hi, lo, table, out, dout, lse, index32)
    if out.dtype != torch.float32 or dout.dtype != torch.float32 or not (out.is_contiguous() and dout.is_contiguous()):
        raise ValueError("attention_bwd_split: out / dout must be contiguous fp32")
    dqkv = (torch.empty((2, B * N, 3 * H * D), dtype=torch.bfloat16, device=hi.device) if pair
            else torch.empty(hi.shape, dtype=torch.float32, device=hi.device))
    delta = torch.empty((B, H, N), dtype=torch.float32, device=hi.device)
    dhi = torch.empty(dout.shape, dtype=torch.bfloat16, device=hi.device)
    dlo = torch.empty(dout.shape, dtype=torch.bfloat16, device=hi.device)
    slab, info = None, None
    if index32 is not None and table is not None:
        chunks = _lib.lib().dm_attention_split_bwd_chunks(B, N, H)
        slab = torch.empty((chunks, H, N, N), dtype=torch.float32, device=hi.device)
        info = (chunks, N, relpos_index_csr(index32, n_bins))
    c = (0, 0, 0) if cube is None else tuple(int(v) for v in cube)
    fn = _lib.lib().dm_attention_split_bwd_pair if pair else _lib.lib().dm_attention_split_bwd
    check(fn(hi.data_ptr(), lo.data_ptr(), _ptr(table), c[0], c[1], c[2], out.data_ptr(), dout.data_ptr(),
             dhi.data_ptr(), dlo.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), delta.data_ptr(), _ptr(slab),
             B, N, H, D, scale, _stream()), "dm_attention_split_bwd")
    return (Planes(dqkv) if pair else dqkv), slab, info


def relpos_index_csr(index32: torch.Tensor, n_bins: int):
    """CSR inverse of a relative_position_index: (positions int32 [<= N*N], offsets int32 [n_bins+1]) with
    positions[offsets[b]:offsets[b+1]] = the flat entries i*N+j whose index is b, ascending.  Entries outside
    [0, n_bins) are dropped.  Cached ON the index tensor (the int32 copy of the module's buffer lives as long as the module; a
    process-wide cache with eviction could drop an entry between a trainer's warm-up steps and its graph capture, and the
    rebuild below synchronises -- illegal while a stream is capturing)."""
    key = (index32._version, n_bins)
    hit = getattr(index32, "_dm_csr", None)
    if hit is None or hit[0] != key:
        flat = index32.reshape(-1).to(torch.int64)
        ok = (flat >= 0) & (flat < n_bins)
        where = torch.nonzero(ok).reshape(-1)
        order = torch.argsort(flat[where], stable=True)
        positions = where[order].to(torch.int32).contiguous()
        counts = torch.bincount(flat[where], minlength=n_bins)
        offsets = torch.zeros(n_bins + 1, dtype=torch.int64, device=index32.device)
        offsets[1:] = torch.cumsum(counts, 0)
        hit = index32._dm_csr = (key, positions, offsets.to(torch.int32).contiguous())
    return hit[1], hit[2]


def attention_bwd(qkv, bias, out, dout, lse, B, N, H, D, scale, index32=None, n_bins=0, bias_t=None, table=None, cube=None):
    """Returns (dqkv, dbias_slab or None, info); `info` goes to relpos_bias_scatter with the slab.  table + cube (the forward ran
    attention_fwd_relpos): the passes that can read the table themselves do, the dense `bias` rows serve the rest."""
    _need_cuda(qkv, bias, out, dout, lse, index32, bias_t, table)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    slab, info = None, None
    if index32 is not None:
        if D != 64 or N > 256:
            raise ValueError(f"the relative-position bias gradient needs head dim 64 and N <= 256 (got D={D}, N={N})")
        chunks = _lib.lib().dm_attention_bwd_batch_chunks(B, N, H, _dt(qkv))
        slab = torch.empty((chunks, H, N, N), dtype=torch.float32, device=qkv.device)
        info = (chunks, N, relpos_index_csr(index32, n_bins))
    if table is not None and cube is not None:
        check(_lib.lib().dm_attention_bwd_relpos(qkv.data_ptr(), table.data_ptr(), int(cube[0]), int(cube[1]), int(cube[2]), _ptr(bias),
                                                 _ptr(bias_t), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                                 delta.data_ptr(), _ptr(slab), B, N, H, D, scale, _dt(qkv), _stream()),
              "dm_attention_bwd_relpos")
        return dqkv, slab, info
    check(_lib.lib().dm_attention_bwd(qkv.data_ptr(), _ptr(bias), _ptr(bias_t), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                      delta.data_ptr(), _ptr(slab), B, N, H, D, scale, _dt(qkv), _stream()),
          "dm_attention_bwd")
    return dqkv, slab, info


def relpos_bias_scatter(slab, dtable, B, H, info, n_bins, accumulate=False):
    """Fold the dense bias-gradient slab of attention_bwd into d(relative_position_bias_table) [n_bins, H]."""
    _need_cuda(slab, dtable)
    chunks, N, (positions, offsets) = info
    check(_lib.lib().dm_relpos_bias_reduce(slab.data_ptr(), positions.data_ptr(), offsets.data_ptr(), dtable.data_ptr(), chunks, H, N,
                                           n_bins, int(accumulate), _stream()), "dm_relpos_bias_reduce")
    return dtable


def patchify(x: torch.Tensor, patch: int, dtype: torch.dtype) -> torch.Tensor:
    _need_cuda(x)
    B, Cc, H, W = x.shape
    if H != W:
        raise ValueError("square images only")
    x = x.contiguous()
    g = H // patch
    cols = torch.empty((B * g * g, Cc * patch * patch), dtype=dtype, device=x.device)
    check(_lib.lib().dm_patchify(x.data_ptr(), cols.data_ptr(), _dt(cols), B, Cc, H, patch, _stream()), "dm_patchify")
    return cols


RESIZE_RULES = {"opencv": 0, "exact_area": 1}      # DM_RESIZE_OPENCV / DM_RESIZE_EXACT_AREA (include/deepmerge_hip.h)


def _resize_rule(resize: str) -> int:
    if resize not in RESIZE_RULES:
        raise ValueError(f"resize must be one of {tuple(RESIZE_RULES)}, got {resize!r}")
    return RESIZE_RULES[resize]


def patch_pyramid(tile: torch.Tensor, xy: torch.Tensor, windows: torch.Tensor, target: int, max_window: Optional[int] = None,
                  resize: str = "opencv") -> torch.Tensor:
    """One scale of the patch pyramid: tile uint8 [bands,H,W], xy int32 [P,2], windows int32 [P] -> float32 [P,bands,t,t].
    resize: "opencv" = cv2.resize(..., INTER_AREA) as the reference calls it (MyUtils1.py:202-216), restated branch by branch;
    "exact_area" = the exact rational area average (oracle/patches.py states both)."""
    _need_cuda(tile, xy, windows)
    if tile.dtype != torch.uint8:
        raise ValueError("tile must be uint8")
    bands, H, W = tile.shape
    P = xy.shape[0]
    if max_window is None:
        max_window = int(windows.max().item())
    out = torch.empty((P, bands, target, target), dtype=torch.float32, device=tile.device)
    check(_lib.lib().dm_patch_pyramid(tile.contiguous().data_ptr(), bands, H, W, xy.to(torch.int32).contiguous().data_ptr(),
                                      windows.to(torch.int32).contiguous().data_ptr(), max_window, P, target, _resize_rule(resize),
                                      out.data_ptr(), _stream()), "dm_patch_pyramid")
    return out


class PatchCols:
    """Patch-embed operand rows of one scale for P samples, produced straight from a uint8 tile by dm_patch_pyramid_cols:
    cols [P * grid * grid, bands * ps * ps] (bf16 / fp32).  PatchEmbed.forward accepts it in place of the [P, C, s, s] image
    tensor, so model code is unchanged while the fp32 patch tensor and the im2col pass never exist."""

    def __init__(self, cols: torch.Tensor, batch: int, side: int, patch: int, bands: int):
        self.cols, self.batch, self.side, self.patch, self.bands = cols, batch, side, patch, bands

    @property
    def shape(self):            # what the image tensor's shape would have been (PatchEmbed's size assertion reads it)
        return (self.batch, self.bands, self.side, self.side)

    def __getitem__(self, sl):  # batch slicing (FeatureIO walks points in chunks)
        g2 = (self.side // self.patch) ** 2
        if not isinstance(sl, slice) or sl.step not in (None, 1):
            raise TypeError("PatchCols supports contiguous batch slices only")
        a, b, _ = sl.indices(self.batch)
        return PatchCols(self.cols[a * g2:b * g2], b - a, self.side, self.patch, self.bands)

    def to(self, *_a, **_k):
        return self

    @staticmethod
    def cat(a: "PatchCols", b: "PatchCols") -> "PatchCols":
        """[a; b] along the batch (what torch.cat((x1, x2), 0) is for image tensors).  Two halves of one buffer -- the trainer's
        static step inputs, the rows a feed wrote for both sides -- are joined without a copy."""
        if (a.side, a.patch, a.bands, a.cols.dtype) != (b.side, b.patch, b.bands, b.cols.dtype):
            raise ValueError("PatchCols.cat: the two sides were built for different scales / dtypes")
        ca, cb = a.cols, b.cols
        if (ca.is_contiguous() and cb.is_contiguous() and ca.untyped_storage().data_ptr() == cb.untyped_storage().data_ptr()
                and cb.data_ptr() == ca.data_ptr() + ca.numel() * ca.element_size()):
            both = ca.as_strided((ca.shape[0] + cb.shape[0], ca.shape[1]), (ca.shape[1], 1), ca.storage_offset())
        else:
            both = torch.cat((ca, cb), 0)
        return PatchCols(both, a.batch + b.batch, a.side, a.patch, a.bands)


def cat_batch(a, b):
    """[a; b] along the batch for image tensors or PatchCols (the two sides of a pair batch)."""
    if isinstance(a, PatchCols) or isinstance(b, PatchCols):
        return PatchCols.cat(a, b)
    return torch.cat((a, b), 0)


def pair_batch_gather(tiles: torch.Tensor, tile_id: Optional[torch.Tensor], xy: torch.Tensor, inner: torch.Tensor, obj: torch.Tensor,
                      scale_index: int, target: int, max_window: int, out: torch.Tensor, grid: int = 0, resize: str = "opencv",
                      region_features: Optional[torch.Tensor] = None, designed: Optional[torch.Tensor] = None,
                      error_flag: Optional[torch.Tensor] = None) -> None:
    """One scale of a training batch, gathered into `out` from resident tiles and a device sample table (dm_pair_batch_gather):
    no host-side window arithmetic, no synchronisation.  tiles uint8 [T, bands, H, W]; tile_id / inner / obj int32 [P]; xy int32
    [P, 2]; out: float32 [P, bands, target, target] (grid = 0) or the patch-embed rows [P * grid^2, bands * (target / grid)^2]
    (bf16 / fp32) -- e.g. the trainer's static step inputs.  region_features [P, 15] + designed [P, 1, 19]: also writes the
    designed-feature rows.  error_flag int32 [1]: set when a sample's tile id / window is out of range (read it with the loss)."""
    _need_cuda(tiles, tile_id, xy, inner, obj, out, region_features, designed, error_flag)
    if tiles.dtype != torch.uint8 or tiles.dim() != 4 or not tiles.is_contiguous():
        raise ValueError("tiles must be a contiguous uint8 [T, bands, H, W] tensor")
    for name, t in (("tile_id", tile_id), ("xy", xy), ("inner", inner), ("obj", obj), ("error_flag", error_flag)):
        if t is not None and (t.dtype != torch.int32 or not t.is_contiguous()):
            raise ValueError(f"{name} must be a contiguous int32 tensor (a per-step cast would be a launch of its own)")
    T, bands, H, W = tiles.shape
    P = xy.shape[0]
    if not out.is_contiguous() or out.numel() != P * bands * target * target:
        raise ValueError(f"out must be contiguous with {P * bands * target * target} elements, got {tuple(out.shape)}")
    if (region_features is None) != (designed is None):
        raise ValueError("designed rows need the region features (and vice versa)")
    if designed is not None and (designed.dtype != torch.float32 or designed.numel() != P * 19 or not designed.is_contiguous()
                                 or region_features.dtype != torch.float32 or region_features.numel() != P * 15 or not region_features.is_contiguous()):
        raise ValueError("region_features must be contiguous float32 [P, 15] and designed contiguous float32 [P, 1, 19]")
    check(_lib.lib().dm_pair_batch_gather(tiles.data_ptr(), T, bands, H, W, _ptr(tile_id), xy.data_ptr(), inner.data_ptr(), obj.data_ptr(),
                                          scale_index, max_window, P, target, grid, _resize_rule(resize), out.data_ptr(), _dt(out),
                                          _ptr(region_features), _ptr(designed), _ptr(error_flag), _stream()), "dm_pair_batch_gather")


def patch_pyramid_cols(tile: torch.Tensor, xy: torch.Tensor, windows: torch.Tensor, target: int, grid: int = 8,
                       dtype: torch.dtype = torch.bfloat16, max_window: Optional[int] = None, resize: str = "opencv") -> PatchCols:
    """One scale of the patch pyramid as patch-embed GEMM rows (dm_patch_pyramid_cols); `resize` as in `patch_pyramid`."""
    _need_cuda(tile, xy, windows)
    if tile.dtype != torch.uint8:
        raise ValueError("tile must be uint8")
    bands, H, W = tile.shape
    P = xy.shape[0]
    if target % grid:
        raise ValueError(f"target {target} is not a multiple of the token grid {grid}")
    ps = target // grid
    if max_window is None:
        max_window = int(windows.max().item())
    cols = torch.empty((P * grid * grid, bands * ps * ps), dtype=dtype, device=tile.device)
    check(_lib.lib().dm_patch_pyramid_cols(tile.contiguous().data_ptr(), bands, H, W, xy.to(torch.int32).contiguous().data_ptr(),
                                           windows.to(torch.int32).contiguous().data_ptr(), max_window, P, target, grid, _resize_rule(resize),
                                           cols.data_ptr(), _dt(cols), _stream()), "dm_patch_pyramid_cols")
    return PatchCols(cols, P, target, ps, bands)


def adam_step(param, grad, m, v, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, param_lp=None):
    _need_cuda(param, grad, m, v)
    check(_lib.lib().dm_adam_step(param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(param_lp), param.numel(), step,
                                  lr, beta1, beta2, eps, grad_scale, _stream()), "dm_adam_step")


def adam_hyper(step: int, lr: float, beta1: float = 0.9, beta2: float = 0.999) -> torch.Tensor:
    """Pinned-host fp32[2] = (lr / (1 - beta1^step), sqrt(1 - beta2^step)) for adam_step_dev."""
    h = torch.empty(2, dtype=torch.float32).pin_memory() if torch.cuda.is_available() else torch.empty(2, dtype=torch.float32)
    check(_lib.lib().dm_adam_hyper(step, lr, beta1, beta2, h.data_ptr()), "dm_adam_hyper")
    return h


def adam_step_dev(param, grad, m, v, hyper_dev, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, param_lp=None, param_lo=None):
    """Adam with the step-dependent scalars read from device memory (capturable in a hipGraph).  param_lo (with param_lp): the updated
    weights also as the hi / lo plane pair of the "bf16x3" products (dm_adam_step_dev_pair)."""
    _need_cuda(param, grad, m, v, hyper_dev)
    if param_lo is not None:
        check(_lib.lib().dm_adam_step_dev_pair(param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), param_lp.data_ptr(), param_lo.data_ptr(),
                                               param.numel(), hyper_dev.data_ptr(), beta1, beta2, eps, grad_scale, _stream()), "dm_adam_step_dev_pair")
        return
    check(_lib.lib().dm_adam_step_dev(param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(param_lp), param.numel(),
                                      hyper_dev.data_ptr(), beta1, beta2, eps, grad_scale, _stream()), "dm_adam_step_dev")


def segment_mean(F: torch.Tensor, ptr: torch.Tensor, idx: torch.Tensor, validate: bool = True) -> torch.Tensor:
    """validate: one host check per call that the CSR is well formed (an out-of-range id would be an out-of-bounds device read,
    i.e. a GPU fault rather than a Python error); pass False when the caller has already checked."""
    _need_cuda(F, ptr, idx)
    S, D = ptr.numel() - 1, F.shape[1]
    if validate and S > 0:
        bad = (ptr[1:] < ptr[:-1]).any() | (ptr[0] != 0) | (ptr[-1] > idx.numel())
        if idx.numel():
            bad = bad | (idx.min() < 0) | (idx.max() >= F.shape[0])
        if bool(bad):
            raise ValueError("segment_mean: ptr must be non-decreasing from 0 to <= len(idx) and idx must index rows of F")
    pooled = torch.empty((S, D), dtype=torch.float32, device=F.device)
    check(_lib.lib().dm_segment_mean(F.data_ptr(), ptr.data_ptr(), idx.data_ptr(), pooled.data_ptr(), S, D, _stream()), "dm_segment_mean")
    return pooled


def edge_similarity(pooled: torch.Tensor, edges: torch.Tensor, margin: float = 1.0, validate: bool = True):
    _need_cuda(pooled, edges)
    E, D = edges.shape[0], pooled.shape[1]
    if validate and E > 0 and int(edges.max()) >= pooled.shape[0]:
        raise ValueError(f"edge_similarity: edge endpoint {int(edges.max())} is not a row of pooled (S = {pooled.shape[0]}); negative ids mean 'no polygon'")
    simi = torch.empty(E, dtype=torch.float32, device=pooled.device)
    merge = torch.empty(E, dtype=torch.uint8, device=pooled.device)
    check(_lib.lib().dm_edge_similarity(pooled.data_ptr(), edges.data_ptr(), simi.data_ptr(), merge.data_ptr(), E, D, margin, _stream()),
          "dm_edge_similarity")
    return simi, merge


# ------------------------------------------------------------------------------------------------
# autograd Functions
# ------------------------------------------------------------------------------------------------
def _as_operand(t: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Contiguous tensor in the MFMA operand dtype (fp32 -> bf16 goes through dm_cast)."""
    t = t.contiguous()
    if t.dtype == dtype:
        return t
    if t.dtype == torch.float32:
        return cast(t, dtype)
    raise ValueError(f"cannot use a {t.dtype} gradient with {dtype} operands")


def _linear_backward(x, w, dy, need_dx, need_dw, need_db, wshape, wparam=None, bparam=None):
    """dx = dy W (NN), dW = dy^T x (TN, split-K), db = column sums -- autograd of y = x W^T + b.
    wparam / bparam: the nn.Parameters behind w / b; when they have a usable gradient sink the results are accumulated
    there and None is returned for them."""
    M, K = x.shape
    N = w.shape[0]
    dx = dw = db = None
    if need_dx:
        dx = torch.empty((M, K), dtype=x.dtype, device=x.device)
        gemm(DM_NN, dy, w, dx, M, K, N, lda=N, ldb=K, ldc=K)
    wsink = _multi_use_sink(wparam, (N, K)) if need_dw else None
    bsink = _multi_use_sink(bparam, (N,)) if need_db else None
    if need_db:
        db = bsink if bsink is not None else torch.empty(N, dtype=torch.float32, device=x.device)
    if need_dw:
        dw = wsink if wsink is not None else torch.empty((N, K), dtype=torch.float32, device=x.device)
        gemm(DM_TN, dy, x, dw, N, K, M, lda=N, ldb=K, ldc=K, accumulate=_acc(wparam, wsink is not None),
             colsum_out=db, colsum_accumulate=_acc(bparam, bsink is not None))     # db rides on the wgrad when it can
        dw = None if wsink is not None else dw.reshape(wshape)
    elif need_db:
        colsum(dy, db, accumulate=_acc(bparam, bsink is not None))
    if bsink is not None:
        db = None
    return dx, dw, db


class LinearFn(torch.autograd.Function):
    """y = x W^T + b [+ residual]  (nn.Linear / k=1 Conv1d / patch Conv2d after patchify;
    nets/ShfitScaleFormer.py:35, :76-79, :119, :134, :948).

    x [M,K] in the operand dtype; weight is the fp32 master [N,K,...]; residual fp32 [M,N] or None.
    """

    @staticmethod
    def forward(ctx, x, weight, bias, residual, out_dtype):
        _save_products(ctx)
        x = x.contiguous()
        M, K = x.shape
        N = weight.shape[0]
        w = lp_weight(weight, x.dtype, (N, K))           # the trainer's bf16 mirror when there is one, else a cast
        y = torch.empty((M, N), dtype=out_dtype, device=x.device)
        gemm(DM_NT, x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=bias,
             residual=None if residual is None else residual.contiguous())
        ctx.save_for_backward(x, w)
        ctx.has_bias, ctx.wshape = bias is not None, weight.shape
        ctx.params = (weight, bias)
        return y

    @staticmethod
    @_replay_products
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dres = dy if ctx.needs_input_grad[3] else None
        dyo = _as_operand(dy, x.dtype)
        dx, dw, db = _linear_backward(x, w, dyo, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                      ctx.has_bias and ctx.needs_input_grad[2], ctx.wshape, *ctx.params)
        return dx, dw, db, dres, None


class PatchEmbedCatFn(torch.autograd.Function):
    """`torch.cat([PatchEmbed_i(x_i) for i], dim=1)` (v3.patch_embed, nets/ShfitScaleFormer.py:869-882) without the cat: every
    scale's patch-embed GEMM writes its 64 tokens per sample straight into the token cube [B, S * T, C] through the GEMM's
    grouped-row addressing (rows_per_group = T, group_stride = S * T * C).  Arguments: n_scales, tokens per scale, then the S
    operand-row matrices cols_i [B * T, K_i], the S weights [C, K_i...] and the S biases."""

    @staticmethod
    def forward(ctx, S, T, *args):
        _save_products(ctx)
        cols, weights, biases = args[:S], args[S:2 * S], args[2 * S:3 * S]
        B = cols[0].shape[0] // T
        Cc = weights[0].shape[0]
        cube = torch.empty((B, S * T, Cc), dtype=torch.float32, device=cols[0].device)
        flat = cube.view(-1)
        ws = []
        for i in range(S):
            c = cols[i].contiguous()
            K = c.shape[1]
            w = lp_weight(weights[i], c.dtype, (Cc, K))
            ws.append(w)
            gemm(DM_NT, c, w, flat[i * T * Cc:], B * T, Cc, K, lda=K, ldb=K, ldc=Cc, bias=biases[i], rows_per_group=T, group_stride=S * T * Cc)
        ctx.save_for_backward(*cols, *ws)
        ctx.cfg = (S, T, B, Cc)
        ctx.params = (weights, biases)
        return cube

    @staticmethod
    @_replay_products
    def backward(ctx, dcube):
        S, T, B, Cc = ctx.cfg
        saved = ctx.saved_tensors
        cols, ws = saved[:S], saved[S:]
        weights, biases = ctx.params
        grads_w, grads_b = [], []
        # every scale's gradient rows in the operands' dtype, [S, B * T, C]: ONE strided cast-copy for all scales when they share a dtype
        same = all(c.dtype == cols[0].dtype for c in cols)
        dys = torch.empty((S, B, T, Cc), dtype=cols[0].dtype, device=dcube.device).copy_(dcube.contiguous().view(B, S, T, Cc).transpose(0, 1)) if same else None
        for i in range(S):
            c = cols[i]
            K = c.shape[1]
            dy = dys[i].view(B * T, Cc) if same else dcube[:, i * T:(i + 1) * T, :].to(c.dtype).reshape(B * T, Cc)
            need_w, need_b = ctx.needs_input_grad[2 + S + i], ctx.needs_input_grad[2 + 2 * S + i]
            _, dw, db = _linear_backward(c, ws[i], dy, False, need_w, need_b and biases[i] is not None, weights[i].shape, weights[i], biases[i])
            grads_w.append(dw)
            grads_b.append(db)
        return (None, None) + (None,) * S + tuple(grads_w) + tuple(grads_b)


class MlpFn(torch.autograd.Function):
    """y = fc2(GELU_erf(fc1(x))) [+ residual]  (Mlp, nets/ShfitScaleFormer.py:52-58; also the
    proj0 -> GELU -> proj1 head of FeatureEmbed, :76-78).  GELU is fused into fc1's epilogue and its
    derivative into the epilogue of fc2's dgrad."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, out_dtype):
        _save_products(ctx)
        x = x.contiguous()
        M, K = x.shape
        Hd, N = w1.shape[0], w2.shape[0]
        w1c = cast(w1.reshape(Hd, K), x.dtype)
        w2c = cast(w2.reshape(N, Hd), x.dtype)
        pre = torch.empty((M, Hd), dtype=x.dtype, device=x.device)
        h = torch.empty((M, Hd), dtype=x.dtype, device=x.device)
        gemm(DM_NT, x, w1c, h, M, Hd, K, lda=K, ldb=K, ldc=Hd, bias=b1, epilogue=DM_EPI_GELU_GRAD, aux=pre, ldaux=Hd)
        y = torch.empty((M, N), dtype=out_dtype, device=x.device)
        gemm(DM_NT, h, w2c, y, M, N, Hd, lda=Hd, ldb=Hd, ldc=N, bias=b2,
             residual=None if residual is None else residual.contiguous())
        ctx.save_for_backward(x, w1c, w2c, pre, h)
        ctx.shapes = (w1.shape, w2.shape)
        ctx.params = (w1, b1, w2, b2)
        return y

    @staticmethod
    @_replay_products
    def backward(ctx, dy):
        x, w1c, w2c, pre, h = ctx.saved_tensors
        M, K = x.shape
        Hd, N = w1c.shape[0], w2c.shape[0]
        need = ctx.needs_input_grad
        dres = dy if need[5] else None
        dyo = _as_operand(dy, x.dtype)
        # fc2: dW2 = dy^T h, db2 = colsum(dy); dpre = (dy W2) * gelu'(pre)  (DGELU epilogue)
        w1, b1, w2, b2 = ctx.params         # (gradients go to the trainer's sinks when there are any, like LinearFn's)
        _, dw2, db2 = _linear_backward(h, w2c, dyo, False, need[3], need[4], ctx.shapes[1], w2, b2)
        dpre = torch.empty((M, Hd), dtype=x.dtype, device=x.device)
        gemm(DM_NN, dyo, w2c, dpre, M, Hd, N, lda=N, ldb=Hd, ldc=Hd, epilogue=DM_EPI_MUL, aux=pre, ldaux=Hd)
        dx, dw1, db1 = _linear_backward(x, w1c, dpre, need[0], need[1], need[2], ctx.shapes[0], w1, b1)
        return dx, dw1, db1, dw2, db2, dres, None


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim, fp32 in, `out_dtype` out (nets/ShfitScaleFormer.py:173, :177, :856)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, out_dtype):
        x = x.contiguous()
        y, mean, rstd = layernorm_fwd(x, gamma, beta, eps, out_dtype)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        C = gamma.numel()
        gs, bs = _multi_use_sink(ctx.params[0], (C,)), _multi_use_sink(ctx.params[1], (C,))
        if gs is not None and bs is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            acc = _acc(ctx.params[0], True)
            _acc(ctx.params[1], True)
            r = layernorm_bwd(dy.contiguous(), x, gamma, mean, rstd, dgamma=gs, dbeta=bs, accumulate=acc)
            return r[0], None, None, None, None
        dx, dg, db = layernorm_bwd(dy.contiguous(), x, gamma, mean, rstd)
        return dx, dg, db, None, None


class AttentionFn(torch.autograd.Function):
    """softmax(scale * q k^T + table[index]) v over [B, N, 3, H, D] packed qkv
    (nets/ShfitScaleFormer.py:119-133; table/index None -> vit_model.py:119-133)."""

    @staticmethod
    def forward(ctx, qkv, table, index32, B, N, H, D, scale):
        qkv = qkv.contiguous()
        bias = bias_t = None
        cube = _inkernel_cube(table, index32, B, N, H, D, qkv.dtype)
        split, scube = _split_attention(table, index32, qkv, B, N, H, D)
        if split:                                            # "bf16x3": split-bf16 products on the matrix pipe, the table read in the kernel
            out, lse, hi, lo = attention_fwd_split(qkv, None if table is None else table.contiguous(), scube, B, N, H, D, scale)
            ctx.save_for_backward(hi, lo, out, lse, index32, table)
            ctx.dims = (B, N, H, D, scale, None if table is None else table.shape[0])
            ctx.split_cube = (scube,)
            return out
        elif cube is not None:                               # the kernel reads the table itself: no dense rows in the forward pass
            out, lse = attention_fwd_relpos(qkv, table.contiguous(), cube, B, N, H, D, scale)
        else:
            if table is not None:
                bias, bias_t = relpos_bias_gather(table.contiguous(), index32, N, transposed=True)
            out, lse = attention_fwd(qkv, bias, B, N, H, D, scale)
        ctx.save_for_backward(qkv, out, lse, bias, bias_t, index32, table if cube is not None else None)
        ctx.dims = (B, N, H, D, scale, None if table is None else table.shape[0])
        return out

    @staticmethod
    def backward(ctx, dout):
        B, N, H, D, scale, n_bins = ctx.dims
        if getattr(ctx, "split_cube", None) is not None:
            hi, lo, out, lse, index32, table = ctx.saved_tensors
            want_table = table is not None and ctx.needs_input_grad[1]
            dqkv, slab, rows = attention_bwd_split(hi, lo, None if table is None else table.contiguous(), ctx.split_cube[0], out,
                                                   dout.contiguous().float(), lse, B, N, H, D, scale, index32 if want_table else None, n_bins or 0)
            dtable = None
            if want_table:
                dtable = torch.empty((n_bins, H), dtype=torch.float32, device=hi.device)
                relpos_bias_scatter(slab, dtable, B, H, rows, n_bins)
            return dqkv, dtable, None, None, None, None, None, None
        qkv, out, lse, bias, bias_t, index32, table = ctx.saved_tensors
        cube = None
        if table is not None:
            table, cube = table.contiguous(), index32._dm_cube
            if _DENSE_BWD_ROWS:
                bias, bias_t = relpos_bias_gather(table, index32, N, transposed=True)
        want_table = (bias is not None or table is not None) and ctx.needs_input_grad[1]
        dqkv, slab, rows = attention_bwd(qkv, bias, out, _as_operand(dout, qkv.dtype), lse, B, N, H, D, scale,
                                         index32 if want_table else None, n_bins or 0, bias_t=bias_t, table=table, cube=cube)
        dtable = None
        if want_table:
            dtable = torch.empty((n_bins, H), dtype=torch.float32, device=qkv.device)
            relpos_bias_scatter(slab, dtable, B, H, rows, n_bins)
        return dqkv, dtable, None, None, None, None, None, None


class GRUCellFn(torch.autograd.Function):
    """One GRU step (torch.nn.GRU semantics, gate order r, z, n; reference Nets.py:60-66): gi [B, 3H] (may be a strided time slice of the
    all-steps input projection), gh [B, 3H] = h W_hh^T + b_hh, h [B, H] -> h_new [B, H].  Kernels: csrc/dm_gru.hip."""

    @staticmethod
    def forward(ctx, gi, gh, h):
        _need_cuda(gi, gh, h)
        B, H = h.shape
        if gi.shape != (B, 3 * H) or gh.shape != (B, 3 * H) or gi.stride(1) != 1:
            raise ValueError(f"GRUCellFn: gi {tuple(gi.shape)} / gh {tuple(gh.shape)} do not fit h {tuple(h.shape)}")
        gh, h = gh.contiguous(), h.contiguous()
        h_new = torch.empty_like(h)
        saved = torch.empty((B, 4 * H), dtype=torch.float32, device=h.device)
        check(_lib.lib().dm_gru_cell_fwd(gi.data_ptr(), gi.stride(0), gh.data_ptr(), h.data_ptr(), h_new.data_ptr(), saved.data_ptr(), B, H,
                                         _stream()), "dm_gru_cell_fwd")
        ctx.save_for_backward(saved, h)
        return h_new

    @staticmethod
    def backward(ctx, dh_new):
        saved, h = ctx.saved_tensors
        B, H = h.shape
        dh_new = dh_new.contiguous()
        dgi = torch.empty((B, 3 * H), dtype=torch.float32, device=h.device)
        dgh, dh = torch.empty_like(dgi), torch.empty_like(h)
        check(_lib.lib().dm_gru_cell_bwd(dh_new.data_ptr(), saved.data_ptr(), h.data_ptr(), dgi.data_ptr(), dgh.data_ptr(), dh.data_ptr(), B, H,
                                         _stream()), "dm_gru_cell_bwd")
        return dgi, dgh, dh


class BatchNormReluFn(torch.autograd.Function):
    """BatchNorm2d -> ReLU -> Dropout2d of the auxiliary heads (reference nets/ShfitScaleFormer.py:340-346) on the channels-last
    matrix x [samples * rows_per_sample, C] the convolution GEMM produces.  `mask`: None or [samples, C] multipliers (0 or
    1 / (1 - p)); running statistics are updated in place when training, exactly as torch.nn.BatchNorm2d does."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, mask, rows_per_sample, eps, momentum, training, relu):
        _need_cuda(x, gamma, beta, running_mean, running_var, mask)
        x = x.float().contiguous()
        M, Cc = x.shape
        if training and M <= 1:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
        y = torch.empty_like(x)
        mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
        rstd = torch.empty(Cc, dtype=torch.float32, device=x.device)
        if mask is not None:
            mask = mask.float().contiguous()
            if mask.shape != (M // rows_per_sample, Cc):
                raise ValueError(f"mask must be [samples, C] = {(M // rows_per_sample, Cc)}, got {tuple(mask.shape)}")
        ws = workspace(_lib.lib().dm_batchnorm_workspace_bytes(M, Cc), x.device, "bn")
        check(_lib.lib().dm_batchnorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(running_mean), _ptr(running_var), _ptr(mask),
                                          rows_per_sample, y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, Cc, eps, momentum, int(training),
                                          int(relu), ws.data_ptr(), _stream()), "dm_batchnorm_fwd")
        ctx.save_for_backward(x, y, gamma, mask, mean, rstd)
        ctx.cfg = (rows_per_sample, bool(training), bool(relu))
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mask, mean, rstd = ctx.saved_tensors
        rows_per_sample, training, relu = ctx.cfg
        M, Cc = x.shape
        dx = torch.empty_like(x)
        gs, bs = _multi_use_sink(ctx.params[0], (Cc,)), _multi_use_sink(ctx.params[1], (Cc,))
        direct = gs is not None and bs is not None
        dg = gs if direct else torch.empty(Cc, dtype=torch.float32, device=x.device)
        db = bs if direct else torch.empty(Cc, dtype=torch.float32, device=x.device)
        ws = workspace(_lib.lib().dm_batchnorm_workspace_bytes(M, Cc), x.device, "bn")
        acc = _acc(ctx.params[0], direct)
        _acc(ctx.params[1], direct)
        check(_lib.lib().dm_batchnorm_bwd(dy.float().contiguous().data_ptr(), x.data_ptr(), y.data_ptr(), gamma.data_ptr(), _ptr(mask), rows_per_sample,
                                          mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), int(acc), M, Cc,
                                          int(training), int(relu), ws.data_ptr(), _stream()), "dm_batchnorm_bwd")
        return dx, (None if direct else dg), (None if direct else db), None, None, None, None, None, None, None, None


class TokenPoolFn(torch.autograd.Function):
    """Per-scale AvgPool2d(2,2) over the token grid (nets/ShfitScaleFormer.py:892-901, :905-914)."""

    @staticmethod
    def forward(ctx, x, S, side):
        x = x.contiguous()
        B, N, Cc = x.shape
        y = torch.empty((B, S * (side // 2) ** 2, Cc), dtype=torch.float32, device=x.device)
        check(_lib.lib().dm_token_pool_fwd(x.data_ptr(), y.data_ptr(), B, S, side, Cc, _stream()), "dm_token_pool_fwd")
        ctx.dims = (B, S, side, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, S, side, Cc = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty((B, S * side * side, Cc), dtype=torch.float32, device=dy.device)
        check(_lib.lib().dm_token_pool_bwd(dy.data_ptr(), dx.data_ptr(), B, S, side, Cc, _stream()), "dm_token_pool_bwd")
        return dx, None, None


class GroupMeanFn(torch.autograd.Function):
    """Mean over groups of g consecutive tokens (AdaptiveAvgPool1d(1) per scale, :930-938)."""

    @staticmethod
    def forward(ctx, x, g):
        x = x.contiguous()
        rows, Cc = x.numel() // (x.shape[-1] * g), x.shape[-1]
        y = torch.empty((rows, Cc), dtype=torch.float32, device=x.device)
        check(_lib.lib().dm_group_mean_fwd(x.data_ptr(), y.data_ptr(), rows, g, Cc, _stream()), "dm_group_mean_fwd")
        ctx.dims = (rows, g, Cc, x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        rows, g, Cc, xshape = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty(xshape, dtype=torch.float32, device=dy.device)
        check(_lib.lib().dm_group_mean_bwd(dy.data_ptr(), dx.data_ptr(), rows, g, Cc, _stream()), "dm_group_mean_bwd")
        return dx, None


class ContrastiveLossFn(torch.autograd.Function):
    """Losses.py:34-38 forward and its gradient in one kernel."""

    @staticmethod
    def forward(ctx, a, b, flag, margin):
        a, b = a.contiguous(), b.contiguous()
        B, D = a.shape
        flag = flag.to(torch.float32).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        da, db = torch.empty_like(a), torch.empty_like(b)
        check(_lib.lib().dm_contrastive_loss(a.data_ptr(), b.data_ptr(), flag.data_ptr(), margin, 1.0, loss.data_ptr(),
                                             da.data_ptr(), db.data_ptr(), B, D, _stream()), "dm_contrastive_loss")
        ctx.save_for_backward(da, db)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        da, db = ctx.saved_tensors
        if is_unit_grad(g):
            return da, db, None, None
        return da * g, db * g, None, None


_unit_grads = {}


def unit_grad(device) -> torch.Tensor:
    """The scalar 1.0 on `device`, one tensor per device for the life of the process: `loss.backward(ops.unit_grad(loss.device))` tells the
    loss functions below that the incoming gradient is exactly 1, so they return their stored gradients as they are (no ones_like, no
    scalar multiply launches in the step; same bits: x * 1.0 == x)."""
    key = torch.device(device)
    t = _unit_grads.get(key)
    if t is None:
        t = _unit_grads[key] = torch.ones((), dtype=torch.float32, device=key)
    return t


def is_unit_grad(g: torch.Tensor) -> bool:
    t = _unit_grads.get(g.device)
    return t is not None and g.data_ptr() == t.data_ptr() and g.dim() == 0


def split_halves(f: torch.Tensor):
    """(f[:B], f[B:]) of a contiguous [2B, D] matrix, tagged so that contrastive_loss can hand the whole matrix to one kernel call and
    return ONE gradient (the Siamese encoders run both sides as one batch)."""
    B = f.shape[0] // 2
    a, b = f[:B], f[B:]
    a._dm_pair, b._dm_pair = f, f
    return a, b


def stacked_halves(a: torch.Tensor, b: torch.Tensor):
    """The matrix split_halves cut a and b from, or None."""
    f = getattr(a, "_dm_pair", None)
    if f is None or f is not getattr(b, "_dm_pair", None) or f.dim() != 2 or not f.is_contiguous() or a.shape != b.shape:
        return None
    B, D = a.shape
    if tuple(f.shape) != (2 * B, D) or a.data_ptr() != f.data_ptr() or b.data_ptr() != f.data_ptr() + B * D * f.element_size():
        return None
    return f


class ContrastivePairLossFn(torch.autograd.Function):
    """ContrastiveLossFn on the two halves of ONE [2B, D] matrix (the Siamese encoder runs both sides as one batch and returns
    f[:B], f[B:]): the gradient is written as one [2B, D] matrix, so autograd has no two slices to pad with zeros and add."""

    @staticmethod
    def forward(ctx, both, flag, margin):
        B, D = both.shape[0] // 2, both.shape[1]
        flag = flag.to(torch.float32).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=both.device)
        d = torch.empty_like(both)
        check(_lib.lib().dm_contrastive_loss(both.data_ptr(), both[B:].data_ptr(), flag.data_ptr(), margin, 1.0, loss.data_ptr(),
                                             d.data_ptr(), d[B:].data_ptr(), B, D, _stream()), "dm_contrastive_loss")
        ctx.save_for_backward(d)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        d, = ctx.saved_tensors
        return (d if is_unit_grad(g) else d * g), None, None


def contrastive_loss(a: torch.Tensor, b: torch.Tensor, flag: torch.Tensor, margin: float) -> torch.Tensor:
    both = stacked_halves(a, b) if (a.dtype == torch.float32 and b.dtype == torch.float32) else None
    if both is not None:
        return ContrastivePairLossFn.apply(both, flag, margin)
    return ContrastiveLossFn.apply(a, b, flag, margin)


class CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss (mean) forward and gradient in one kernel (Losses.py:52-53, :83-84).  `target`: int64 class
    indices [B] or float class probabilities [B,K] (the reference's argument is named *_one_hot)."""

    @staticmethod
    def forward(ctx, logits, target):
        _need_cuda(logits, target)
        x = logits.float().contiguous()
        B, K = x.shape
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        dl = torch.empty_like(x)
        if target.dtype.is_floating_point:
            if target.shape != x.shape:
                raise ValueError(f"probability targets must be [B,K] = {tuple(x.shape)}, got {tuple(target.shape)}")
            t = target.float().contiguous()
            ti, tp = 0, t.data_ptr()
        else:
            if target.shape != (B,):
                raise ValueError(f"index targets must be [B] = ({B},), got {tuple(target.shape)}")
            t = target.to(torch.int64).contiguous()
            if B and (int(t.min()) < 0 or int(t.max()) >= K):
                raise IndexError(f"Target {int(t.max())} is out of bounds.")
            ti, tp = t.data_ptr(), 0
        check(_lib.lib().dm_cross_entropy(x.data_ptr(), ti, tp, 1.0, loss.data_ptr(), dl.data_ptr(), B, K, _stream()), "dm_cross_entropy")
        ctx.save_for_backward(dl)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None


# ------------------------------------------------------------------------------------------------
# fused transformer block
# ------------------------------------------------------------------------------------------------


def lp_weight(weight: torch.Tensor, dtype: torch.dtype, shape2d) -> torch.Tensor:
    """Operand-dtype copy of a master weight.  The trainer keeps a flat bf16 mirror updated by the fused
    Adam kernel and attaches views as `weight._dm_lp`; otherwise cast on the fly."""
    if dtype == torch.float32:
        return weight.reshape(shape2d)
    lp = getattr(weight, "_dm_lp", None)
    if lp is not None and lp.dtype == dtype:
        return lp.reshape(shape2d)
    return cast(weight.reshape(shape2d), dtype)


def pair_weight(weight: torch.Tensor, w2d: torch.Tensor) -> Planes:
    """The hi / lo plane pair of a master weight for the folded "bf16x3" products: the mirror pair the trainer's Adam kernel keeps
    (`weight._dm_pair_src` = (flat [2, total] bf16 buffer, offset): a [2, rows, cols] VIEW whose plane stride is the buffer's length),
    else a split of the weight."""
    src = getattr(weight, "_dm_pair_src", None)
    if src is not None and _PLANES:
        flat, off = src
        rows, cols = w2d.shape
        if rows * cols == weight.numel() and off % 8 == 0 and flat.shape[1] % 8 == 0:
            return Planes(torch.as_strided(flat, (2, rows, cols), (flat.shape[1], cols, 1), off))
    return split_planes(w2d)


def _operand_grad(dy: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """The incoming fp32 gradient as an MFMA operand; reuses the bf16 copy the previous LayerNorm
    backward already wrote, if there is one."""
    if dy.dtype == dtype and dy.is_contiguous():
        return dy
    lp = getattr(dy, "_dm_lp_copy", None)      # attached by the BlockFn.backward that produced this very tensor object
    if lp is not None and lp.dtype == dtype and lp.numel() == dy.numel():
        return lp.reshape(dy.shape)
    return _as_operand(dy, dtype)


def _grad_out(param: torch.Tensor, shape, device):
    """Where a parameter gradient should be written: straight into the trainer's flat gradient buffer
    (`param._dm_grad_sink`, accumulate) when there is one, else a fresh tensor handed back to autograd."""
    sink = getattr(param, "_dm_grad_sink", None)
    if sink is not None:
        return sink.view(shape), True
    return torch.empty(shape, dtype=torch.float32, device=device), False


def _acc(param, direct) -> bool:
    """`accumulate` flag for a gradient write into `param`'s sink.  Parameters the trainer tracks (`_dm_gw`, set by FlatParams for the
    parameters of the fused blocks) are not zeroed at the start of a step: their FIRST write of the step stores, later ones (a block
    applied twice, gradient accumulation over several backward passes) add.  Untracked parameters: the buffer was zeroed, always add."""
    if not direct:
        return False
    st = getattr(param, "_dm_gw", None)
    if st is None or st[0]:
        return True
    st[0] = True
    return False


def _multi_use_sink(param, shape):
    """Gradient sink for the generic Functions (Linear, Mlp, LayerNorm), whose parameters may be used several times per step
    (the shared `norm`, the aux heads): their contributions accumulate into the trainer's flat buffer directly.  (The
    data-parallel exchange works on whole backward segments, so it never needs to know when one parameter is complete.)"""
    if param is None:
        return None
    sink = getattr(param, "_dm_grad_sink", None)
    return None if sink is None else sink.view(shape)


def _grad_done(param: torch.Tensor, g: torch.Tensor, direct: bool):
    """Value to return to autograd for this parameter (None when it went to the sink)."""
    if not direct:
        return g.view(param.shape) if g.shape != param.shape else g
    return None


# Weight-gradient GEMMs of a block do not feed anything else in that block's backward: with DM_WGRAD_STREAM=<max tokens> they are
# enqueued on a side stream (own split-K workspace) next to the dgrad chain for blocks of at most that many tokens, and joined
# before the block's backward returns.  Meant for the small stages, whose kernels cannot fill the chip on their own.
# Measured under graph replay: 6.96 ms/step with the 4096-token stages on the side stream, 6.75 ms with every block, 6.75 ms
# without -- the fork / join edges cost what the overlap gains, so the default is off (0).  Round 4, faster kernels, same verdict:
# 5.71-5.73 ms off, 5.88 ms with every block on the side stream, 5.87 ms with ONE join per backward pass instead of one per block,
# 5.85 ms with only the <= 4096-token stages (the large kernels own a CU's whole LDS / register file, so nothing co-resides).
_WGRAD_SIDE_TOKENS = int(os.environ.get("DM_WGRAD_STREAM", "0"))
# The same independence, used the other way round (round 5): for blocks of at most DM_WGRAD_GROUP tokens the four weight gradients are
# collected and issued at the END of the block's backward as ONE dm_gemm_grouped call -- 144 tiles in one launch of the 4-wave kernel, no K
# slices, no slab, no reduction launches (tools/mb_grouped_estimate.py: 78 -> 34 us per block at 1024 tokens, 142 -> 85 us at 4096,
# 250 -> 206 us at 12288; in the step the 16384-token blocks gain nothing -- every product fills the chip with its own slices).  Headline
# step 5.28 -> 5.12 ms, "bf16x3" 11.03 -> 10.89 ms (same box, tools/ab_grouped.sh).  bf16 operands or plane pairs ("bf16x3"); 0 = off.
_WGRAD_GROUP_TOKENS = int(os.environ.get("DM_WGRAD_GROUP", "12288"))
_WGRAD_PAIR = os.environ.get("DM_WGRAD_PAIR", "1") != "0"      # larger blocks: the proj and qkv gradients in one sliced launch
_side_streams = {}


def _side_stream(device):
    key = torch.device(device).index
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


class BlockFn(torch.autograd.Function):
    """Whole pre-norm block  x += proj(attn(LN1(x)));  x += fc2(GELU(fc1(LN2(x))))
    (CrossScaleBlock, nets/ShfitScaleFormer.py:181-184; vit_model.Block :182-185 with table=None).

    One autograd node per block: both residual adds live in GEMM epilogues, the residual-gradient adds in
    the LayerNorm backward kernels, GELU' in the fc2 dgrad epilogue; nothing is left to torch elementwise ops.
    """

    @staticmethod
    def forward(ctx, x, n1w, n1b, table, index32, qkv_w, qkv_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b,
                heads, eps, scale, dtype):
        _save_products(ctx)
        x = x.contiguous()
        B, N, Cc = x.shape
        M, Hd, D = B * N, fc1_w.shape[0], Cc // heads
        dev = x.device
        wq, wp = lp_weight(qkv_w, dtype, (3 * Cc, Cc)), lp_weight(proj_w, dtype, (Cc, Cc))
        w1, w2 = lp_weight(fc1_w, dtype, (Hd, Cc)), lp_weight(fc2_w, dtype, (Cc, Hd))
        x2d = x.view(M, Cc)
        # "bf16x3": every GEMM operand of the block as ONE hi / lo plane pair (Planes), made where the tensor is produced and used by
        # all its consumers -- the forward product, and in the backward pass the weight gradient (as the right operand) or both the
        # weight gradient and the data gradient (dy, as the left operand).  12 splits per block and step instead of 24, two pieces
        # written per split instead of three, and the saved activations ARE the plane pairs (same bytes as the fp32 tensors).
        # (Cc <= 1024: the LayerNorm kernels write / take the pair only for rows that fit their register form -- ViT-H's 1280-wide
        # rows keep fp32 activations and per-use splits)
        planes = (dtype == torch.float32 and _FP32_PRODUCTS == "bf16x3" and planes_ok(M, Cc) and planes_ok(M, Hd) and planes_ok(Hd, Cc)
                  and Cc <= _PAIR_LN_MAX_COLS and M * Cc * Cc >= _SPLIT_MIN_WORK)
        if planes:      # (the optimizer's mirror pairs when a trainer keeps them: no per-step split of the weights)
            wq, wp, w1, w2 = (pair_weight(P, w) for P, w in ((qkv_w, wq), (proj_w, wp), (fc1_w, w1), (fc2_w, w2)))
        y1, mean1, rstd1 = layernorm_fwd(x2d, n1w, n1b, eps, dtype, pair=planes)
        split, scube = _split_attention(table, index32, dtype, B, N, heads, D)
        # (plane pairs + the split-bf16 attention: the qkv product writes the two images the attention kernels read)
        qkv = (Planes(torch.empty((2, M, 3 * Cc), dtype=torch.bfloat16, device=dev)) if (planes and split)
               else torch.empty((M, 3 * Cc), dtype=dtype, device=dev))
        gemm(DM_NT, y1, wq, qkv, M, 3 * Cc, Cc, lda=Cc, ldb=Cc, ldc=3 * Cc, bias=qkv_b)
        bias = bias_t = None
        cube = _inkernel_cube(table, index32, B, N, heads, D, dtype)
        split_imgs = None
        if split:
            o_op = None
            if planes:
                o, lse, hi, lo, o_op = attention_fwd_split(qkv, None if table is None else table.contiguous(), scube, B, N, heads, D, scale, out_pair=True)
            else:
                o, lse, hi, lo = attention_fwd_split(qkv, None if table is None else table.contiguous(), scube, B, N, heads, D, scale)
            split_imgs = (hi, lo, scube)
        elif cube is not None:
            o, lse = attention_fwd_relpos(qkv, table.contiguous(), cube, B, N, heads, D, scale)
        else:
            if table is not None:
                bias, bias_t = relpos_bias_gather(table.contiguous(), index32, N, transposed=True)
            o, lse = attention_fwd(qkv, bias, B, N, heads, D, scale)
        x1 = torch.empty((M, Cc), dtype=torch.float32, device=dev)
        if not (planes and split):
            o_op = split_planes(o.view(M, Cc)) if planes else o.view(M, Cc)
        gemm(DM_NT, o_op, wp, x1, M, Cc, Cc, lda=Cc, ldb=Cc, ldc=Cc, bias=proj_b, residual=x2d)
        y2, mean2, rstd2 = layernorm_fwd(x1, n2w, n2b, eps, dtype, pair=planes)
        h = Planes(torch.empty((2, M, Hd), dtype=torch.bfloat16, device=dev)) if planes else torch.empty((M, Hd), dtype=dtype, device=dev)
        if any(ctx.needs_input_grad):
            pre = torch.empty((M, Hd), dtype=dtype, device=dev)       # GELU'(pre-activation), saved for backward
            gemm(DM_NT, y2, w1, h, M, Hd, Cc, lda=Cc, ldb=Cc, ldc=Hd, bias=fc1_b, epilogue=DM_EPI_GELU_GRAD, aux=pre, ldaux=Hd)
        else:                                                         # inference (torch.no_grad / frozen): nothing to save
            pre = None
            gemm(DM_NT, y2, w1, h, M, Hd, Cc, lda=Cc, ldb=Cc, ldc=Hd, bias=fc1_b, epilogue=DM_EPI_GELU)
        x2 = torch.empty((M, Cc), dtype=torch.float32, device=dev)
        gemm(DM_NT, h, w2, x2, M, Cc, Hd, lda=Hd, ldb=Hd, ldc=Cc, bias=fc2_b, residual=x1)
        if planes:          # (the bf16 [2, rows, cols] tensors travel through save_for_backward like any other)
            y1, y2, h, wq, wp, w1, w2, o_op = y1.t, y2.t, h.t, wq.t, wp.t, w1.t, w2.t, o_op.t
            if isinstance(qkv, Planes):
                qkv = qkv.t     # (its two planes are the images in ctx.split_imgs)
        ctx.planes = planes
        ctx.save_for_backward(x2d, mean1, rstd1, y1, qkv, o, lse, bias, bias_t, index32, x1, mean2, rstd2, y2, pre, h,
                              wq, wp, w1, w2, n1w, n2w, o_op if planes else None)
        ctx.dims = (B, N, Cc, heads, D, Hd, scale, None if table is None else table.shape[0])
        ctx.table_in_kernel = cube is not None
        ctx.split_imgs = split_imgs         # (two bf16 tensors of qkv's size, alive until this node's backward has run)
        ctx.params = (n1w, n1b, table, qkv_w, qkv_b, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b)
        return x2.view(B, N, Cc)

    @staticmethod
    @_replay_products
    def backward(ctx, dx2):
        (x, mean1, rstd1, y1, qkv, o, lse, bias, bias_t, index32, x1, mean2, rstd2, y2, pre, h,
         wq, wp, w1, w2, n1w, n2w, o_op) = ctx.saved_tensors
        planes = ctx.planes
        if planes:
            y1, y2, h, wq, wp, w1, w2, o_op = (Planes(t_) for t_ in (y1, y2, h, wq, wp, w1, w2, o_op))
        (P_n1w, P_n1b, P_table, P_qkv_w, P_qkv_b, P_proj_w, P_proj_b, P_n2w, P_n2b, P_fc1_w, P_fc1_b, P_fc2_w, P_fc2_b) = ctx.params
        B, N, Cc, heads, D, Hd, scale, n_bins = ctx.dims
        M = B * N
        dtype, dev = (torch.float32 if planes else y1.dtype), x.device
        lp = dtype != torch.float32
        lp_copy = getattr(dx2, "_dm_lp_copy", None)
        dy_pair = getattr(dx2, "_dm_planes", None)   # the plane pair the LayerNorm backward of the NEXT block wrote next to this very tensor
        if dy_pair is not None and dy_pair[1] != (dx2.data_ptr(), dx2._version):
            dy_pair = None                           # (the engine summed another contribution into the tensor in place: the pair is stale)
        elif dy_pair is not None:
            dy_pair = dy_pair[0]
        dx2 = dx2.contiguous().view(M, Cc)
        if lp_copy is not None:
            dx2._dm_lp_copy = lp_copy                # (.contiguous() is the same storage; the attribute rides along)
        dy = _operand_grad(dx2, dtype)
        if planes and isinstance(dy_pair, Planes) and (dy_pair.rows, dy_pair.cols) == (M, Cc):
            dy = dy_pair
        # ---- MLP ---------------------------------------------------------------------------
        dw2, k_w2 = _grad_out(P_fc2_w, (Cc, Hd), dev)
        db2, k_b2 = _grad_out(P_fc2_b, (Cc,), dev)
        side = _side_stream(dev) if 0 < M <= _WGRAD_SIDE_TOKENS else None

        pending = [] if (side is None and (lp or planes) and 0 < M <= _WGRAD_GROUP_TOKENS) else None
        # larger blocks: every product fills the chip with its own K slices, except the small proj gradient (12 tiles: 16 short slices) --
        # it waits for the qkv gradient and the two share one sliced launch (dm_gemm_grouped's third form)
        late = [] if (pending is None and side is None and (lp or planes) and _WGRAD_PAIR) else None

        def wgrad(*a, pair=False, **kw):
            if pending is not None:           # issued together at the end of this backward (gemm_grouped)
                pending.append((a, kw))
                return None
            if late is not None and pair:
                late.append((a, kw))
                return None
            if side is None:
                return gemm(*a, **kw)
            side.wait_stream(torch.cuda.current_stream())          # the operands' producers
            with torch.cuda.stream(side):
                return gemm(*a, ws_slot="gemm_side", **kw)

        def bias_grad(g2d, db, acc_b, direct):
            """(operand for the two products that read the gradient g2d, kwargs that make the weight-gradient call produce db).
            The column sums always ride on the weight gradient -- of a plane pair: colsum(hi) + colsum(lo), whether the pair came
            from its producer or from the split pass here, so the bias gradient does not depend on which of the two happened
            (a segmented backward hands over plain tensors at its cuts)."""
            if planes and not isinstance(g2d, Planes):
                g2d = split_planes(g2d)
            return g2d, dict(colsum_out=db, colsum_accumulate=acc_b)
        dy, cs = bias_grad(dy, db2, _acc(P_fc2_b, k_b2), k_b2)
        wgrad(DM_TN, dy, h, dw2, Cc, Hd, M, lda=Cc, ldb=Hd, ldc=Hd, accumulate=_acc(P_fc2_w, k_w2), **cs)
        dpre = Planes(torch.empty((2, M, Hd), dtype=torch.bfloat16, device=dev)) if planes else torch.empty((M, Hd), dtype=dtype, device=dev)
        gemm(DM_NN, dy, w2, dpre, M, Hd, Cc, lda=Cc, ldb=Hd, ldc=None if planes else Hd, epilogue=DM_EPI_MUL, aux=pre, ldaux=Hd)
        dw1, k_w1 = _grad_out(P_fc1_w, (Hd, Cc), dev)
        db1, k_b1 = _grad_out(P_fc1_b, (Hd,), dev)
        dpre, cs = bias_grad(dpre, db1, _acc(P_fc1_b, k_b1), k_b1)
        wgrad(DM_TN, dpre, y2, dw1, Hd, Cc, M, lda=Hd, ldb=Cc, ldc=Cc, accumulate=_acc(P_fc1_w, k_w1), **cs)
        dy2 = torch.empty((M, Cc), dtype=dtype, device=dev)
        gemm(DM_NN, dpre, w1, dy2, M, Cc, Hd, lda=Hd, ldb=Cc, ldc=Cc)
        dg2, k_n2 = _grad_out(P_n2w, (Cc,), dev)
        dbt2, k_n2b = _grad_out(P_n2b, (Cc,), dev)
        if k_n2 != k_n2b:      # mixed sinks: fall back to fresh tensors for both
            dg2, dbt2, k_n2, k_n2b = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev), False, False
        a_n2 = _acc(P_n2w, k_n2)
        _acc(P_n2b, k_n2b)                              # (gamma and beta are written by the same launch)
        r = layernorm_bwd(dy2, x1, n2w, mean2, rstd2, dres=dx2, dgamma=dg2, dbeta=dbt2, accumulate=a_n2, want_lp=lp, defer=bool(k_n2 and k_n2b),
                          want_pair=planes)
        dx1, dx1_lp = (r[0], r[1]) if (lp or planes) else (r[0], r[0])      # (planes: r[1] is the plane pair of dx1)
        # ---- attention -----------------------------------------------------------------------
        dwp, k_wp = _grad_out(P_proj_w, (Cc, Cc), dev)
        dbp, k_bp = _grad_out(P_proj_b, (Cc,), dev)
        dx1_op, cs = bias_grad(dx1_lp, dbp, _acc(P_proj_b, k_bp), k_bp)
        wgrad(DM_TN, dx1_op, o_op if planes else o.view(M, Cc), dwp, Cc, Cc, M, lda=Cc, ldb=Cc, ldc=Cc, accumulate=_acc(P_proj_w, k_wp), pair=True, **cs)
        do = torch.empty((M, Cc), dtype=dtype, device=dev)
        gemm(DM_NN, dx1_op, wp, do, M, Cc, Cc, lda=Cc, ldb=Cc, ldc=Cc)
        tab, cube = None, None
        if ctx.split_imgs is not None:                       # "bf16x3": the split-bf16 backward kernels (table read in the kernel, slab included)
            hi, lo, scube = ctx.split_imgs
            ctx.split_imgs = None
            want_table = P_table is not None
            dqkv, slab, rows = attention_bwd_split(hi, lo, None if P_table is None else P_table.detach().contiguous(), scube, o.view(B, N, Cc),
                                                   do.view(B, N, Cc), lse, B, N, heads, D, scale, index32 if want_table else None, n_bins or 0,
                                                   pair=planes)
        else:
            if ctx.table_in_kernel:                          # the forward kernel read the table itself; so do both backward passes
                tab, cube = P_table.detach().contiguous(), index32._dm_cube
                if _DENSE_BWD_ROWS:
                    bias, bias_t = relpos_bias_gather(tab, index32, N, transposed=True)
            want_table = bias is not None or tab is not None
            dqkv, slab, rows = attention_bwd(qkv, bias, o, do, lse, B, N, heads, D, scale,
                                             index32 if want_table else None, n_bins or 0, bias_t=bias_t, table=tab, cube=cube)
        dtable, k_t = None, False
        if want_table:
            dtable, k_t = _grad_out(P_table, (n_bins, heads), dev)
            relpos_bias_scatter(slab, dtable, B, heads, rows, n_bins, accumulate=_acc(P_table, k_t))
        dqkv2 = dqkv if isinstance(dqkv, Planes) else dqkv.view(M, 3 * Cc)
        dwq, k_wq = _grad_out(P_qkv_w, (3 * Cc, Cc), dev)
        dbq, k_bq = _grad_out(P_qkv_b, (3 * Cc,), dev)
        dqkv_op, cs = bias_grad(dqkv2, dbq, _acc(P_qkv_b, k_bq), k_bq)
        wgrad(DM_TN, dqkv_op, y1, dwq, 3 * Cc, Cc, M, lda=3 * Cc, ldb=Cc, ldc=Cc, accumulate=_acc(P_qkv_w, k_wq), pair=True, **cs)
        dy1 = torch.empty((M, Cc), dtype=dtype, device=dev)
        gemm(DM_NN, dqkv_op, wq, dy1, M, Cc, 3 * Cc, lda=3 * Cc, ldb=Cc, ldc=Cc)
        dg1, k_n1 = _grad_out(P_n1w, (Cc,), dev)
        dbt1, k_n1b = _grad_out(P_n1b, (Cc,), dev)
        if k_n1 != k_n1b:
            dg1, dbt1, k_n1, k_n1b = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev), False, False
        a_n1 = _acc(P_n1w, k_n1)
        _acc(P_n1b, k_n1b)
        r = layernorm_bwd(dy1, x, n1w, mean1, rstd1, dres=dx1, dgamma=dg1, dbeta=dbt1, accumulate=a_n1, want_lp=lp, defer=bool(k_n1 and k_n1b),
                          want_pair=planes)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)          # join: every weight gradient of this block is complete
        if pending:
            gemm_grouped(pending)
        if late:
            gemm_grouped(late)
        dx = r[0].view(B, N, Cc)
        if lp:
            # The bf16 copy the LayerNorm backward wrote rides on the tensor object autograd hands to the next node (the
            # engine preserves the Python object); a consumer that gets some other tensor simply casts.
            dx._dm_lp_copy = r[1]
        if planes:
            dx._dm_planes = (r[1], (dx.data_ptr(), dx._version))      # likewise the plane pair ("bf16x3"); a consumer that gets another tensor splits
        return (dx, _grad_done(P_n1w, dg1, k_n1), _grad_done(P_n1b, dbt1, k_n1b),
                _grad_done(P_table, dtable, k_t) if want_table else None, None,
                _grad_done(P_qkv_w, dwq, k_wq), _grad_done(P_qkv_b, dbq, k_bq),
                _grad_done(P_proj_w, dwp, k_wp), _grad_done(P_proj_b, dbp, k_bp),
                _grad_done(P_n2w, dg2, k_n2), _grad_done(P_n2b, dbt2, k_n2b),
                _grad_done(P_fc1_w, dw1, k_w1), _grad_done(P_fc1_b, db1, k_b1),
                _grad_done(P_fc2_w, dw2, k_w2), _grad_done(P_fc2_b, db2, k_b2), None, None, None, None)
