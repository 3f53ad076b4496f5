#!/usr/bin/env python3
"""Per-shape kernel timings on the GPU (hipEvents via torch on the launch stream), random data.
    python tools/microbench.py gemm | attn | rows
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepmerge_amd import ops  # noqa: E402
from deepmerge_amd._lib import DM_EPI_GELU, DM_NN, DM_NT, DM_TN  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def gemm(dt=torch.bfloat16):
    shapes = []
    for M in (16384, 4096, 1024, 61440):
        shapes += [("NT qkv", DM_NT, M, 2304, 768), ("NT proj", DM_NT, M, 768, 768), ("NT fc1+gelu", DM_NT, M, 3072, 768),
                   ("NT fc2", DM_NT, M, 768, 3072), ("NN dqkv", DM_NN, M, 768, 2304), ("NN dfc1", DM_NN, M, 768, 3072),
                   ("NN dfc2", DM_NN, M, 3072, 768), ("TN wqkv", DM_TN, 2304, 768, M), ("TN wproj", DM_TN, 768, 768, M),
                   ("TN wfc1", DM_TN, 3072, 768, M), ("TN wfc2", DM_TN, 768, 3072, M)]
    shapes += [("NT 4096^3", DM_NT, 4096, 4096, 4096), ("NT 8192^3", DM_NT, 8192, 8192, 8192)]
    for name, lay, M, N, K in shapes:
        if lay == DM_NT:
            A = torch.randn(M, K, device=DEV).to(dt); B = torch.randn(N, K, device=DEV).to(dt); lda, ldb = K, K
        elif lay == DM_NN:
            A = torch.randn(M, K, device=DEV).to(dt); B = torch.randn(K, N, device=DEV).to(dt); lda, ldb = K, N
        else:
            A = torch.randn(K, M, device=DEV).to(dt); B = torch.randn(K, N, device=DEV).to(dt); lda, ldb = M, N
        cdt = torch.float32 if lay == DM_TN or "proj" in name or "fc2" in name else dt
        C = torch.empty(M, N, device=DEV, dtype=cdt)
        kw = {}
        if "gelu" in name:
            kw = dict(epilogue=DM_EPI_GELU, aux=torch.empty(M, N, device=DEV, dtype=dt), bias=torch.randn(N, device=DEV))
        t = timeit(lambda: ops.gemm(lay, A, B, C, M, N, K, lda=lda, ldb=ldb, ldc=N, **kw))
        print(f"{name:14s} M={M:6d} N={N:5d} K={K:6d}  {t * 1e6:9.1f} us  {2.0 * M * N * K / t / 1e12:7.1f} TFLOP/s", flush=True)


def attn(dt=torch.bfloat16):
    for B, N in ((64, 256), (64, 64), (64, 16), (240, 256), (64, 192), (256, 197)):
        H, D = 12, 64
        qkv = torch.randn(B, N, 3, H, D, device=DEV).to(dt)
        nb = 1575
        table = torch.randn(nb, H, device=DEV)
        idx = torch.randint(0, nb, (N, N), device=DEV, dtype=torch.int32)
        bias, bias_t = ops.relpos_bias_gather(table, idx, N, transposed=True)
        out, lse = ops.attention_fwd(qkv, bias, B, N, H, D, 0.125)
        dout = torch.randn_like(out)
        tf = timeit(lambda: ops.attention_fwd(qkv, bias, B, N, H, D, 0.125))
        tb = timeit(lambda: ops.attention_bwd(qkv, bias, out, dout, lse, B, N, H, D, 0.125, idx, nb, bias_t=bias_t))
        f = 4.0 * B * H * N * N * D
        print(f"attn B={B:4d} N={N:4d}: fwd {tf * 1e6:8.1f} us {f / tf / 1e12:6.1f} TF/s | bwd {tb * 1e6:8.1f} us {2.5 * f / tb / 1e12:6.1f} TF/s", flush=True)


def rows():
    M, C = 16384, 768
    x = torch.randn(M, C, device=DEV); g = torch.ones(C, device=DEV); b = torch.zeros(C, device=DEV)
    t = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-5, torch.bfloat16))
    print(f"layernorm_fwd {M}x{C}: {t * 1e6:.1f} us  {(M * C * 6) / t / 1e9:.0f} GB/s")
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5, torch.bfloat16)
    dy = torch.randn(M, C, device=DEV).bfloat16()
    t = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dres=x))
    print(f"layernorm_bwd {M}x{C}: {t * 1e6:.1f} us  {(M * C * 14) / t / 1e9:.0f} GB/s")
    X = torch.randn(M, 3072, device=DEV).bfloat16(); o = torch.empty(3072, device=DEV)
    t = timeit(lambda: ops.colsum(X, o))
    print(f"colsum {M}x3072 bf16: {t * 1e6:.1f} us  {(M * 3072 * 2) / t / 1e9:.0f} GB/s")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    {"gemm": gemm, "attn": attn, "rows": rows}[which]()
