"""The folded (bf16x3) products of a stage-0 block on each GEMM family: which kernel should take which shape.
usage: python tools/mb_fold.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from deepmerge_amd import ops
from deepmerge_amd._lib import DM_NT, DM_NN, DM_TN, DM_EPI_MUL, DM_EPI_GELU_GRAD, DM_EPI_NONE
dev = "cuda:0"
T, C, H = int(os.environ.get("TOKENS", 16384)), 768, 3072
R = 3
def planes(r, c): return ops.Planes(torch.randn(2, r, c, device=dev).bfloat16())
def timeit(fn, it=20):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(it): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
FAMS = {"default": {}, "tiles": {"DM_GEMM_W4": "0", "DM_GEMM_256": "0", "DM_GEMM_RING": "0"}, "w4": {"DM_GEMM_W4": "2", "DM_GEMM_256": "0", "DM_GEMM_RING": "0"},
        "256": {"DM_GEMM_W4": "0", "DM_GEMM_256": "2", "DM_GEMM_RING": "0"}, "ring": {"DM_GEMM_W4": "0", "DM_GEMM_256": "0", "DM_GEMM_RING": "2"}}
def case(name, lay, M, N, K, out_pair, **kw):
    a_shape = (K, M) if lay == DM_TN else (M, K)
    b_shape = (N, K) if lay == DM_NT else (K, N)
    sets = [(planes(*a_shape), planes(*b_shape), planes(M, N) if out_pair else torch.empty(M, N, device=dev)) for _ in range(R)]
    aux = torch.randn(M, N, device=dev) if kw.pop("aux", False) else None
    bias = torch.randn(N, device=dev) if kw.pop("bias", False) else None
    res = torch.randn(M, N, device=dev) if kw.pop("res", False) else None
    row = []
    for fam, env in FAMS.items():
        for k in ("DM_GEMM_W4", "DM_GEMM_256", "DM_GEMM_RING"): os.environ.pop(k, None)
        os.environ.update(env)
        def run(i):
            A, B, Cc = sets[i % R]
            ops.gemm(lay, A, B, Cc, M, N, K, bias=bias, residual=res, aux=aux, ldaux=N if aux is not None else None, **kw)
        try:
            row.append(f"{fam} {timeit(run):6.1f}")
        except Exception as e:
            row.append(f"{fam} fail")
    fl = 2.0 * M * N * 3 * K
    print(f"{name:22s} {M}x{N}x3*{K}: " + " | ".join(row) + f"   (1000 TF = {fl/1e15*1e6:.0f} us)", flush=True)
case("qkv fwd (pair out)", DM_NT, T, 3 * C, C, True, bias=True)
case("proj fwd (+res)", DM_NT, T, C, C, False, bias=True, res=True)
case("fc1 fwd (pair, gelu')", DM_NT, T, H, C, True, bias=True, aux=True, epilogue=DM_EPI_GELU_GRAD)
case("fc2 fwd (+res)", DM_NT, T, C, H, False, bias=True, res=True)
case("fc2 dgrad (pair, mul)", DM_NN, T, H, C, True, aux=True, epilogue=DM_EPI_MUL)
case("fc1 dgrad", DM_NN, T, C, H, False)
case("proj dgrad", DM_NN, T, C, C, False)
case("qkv dgrad", DM_NN, T, C, 3 * C, False)
case("fc2 wgrad", DM_TN, C, H, T, False)
case("qkv wgrad", DM_TN, 3 * C, C, T, False)
case("proj wgrad", DM_TN, C, C, T, False)
