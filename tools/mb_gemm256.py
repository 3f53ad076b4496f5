"""Correctness + timing of the 256x256 pipeline (NT / NN / TN) vs the 128x128 kernel (run once per DM_GEMM_256 setting)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deepmerge_amd import ops
from deepmerge_amd._lib import DM_NT, DM_NN, DM_TN

dev = "cuda:0"
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

g = torch.Generator(device=dev); g.manual_seed(1)
def ints(shape): return torch.randint(-2, 3, shape, device=dev, generator=g).to(torch.bfloat16)
cases = [("NT", 4096, 3072, 768), ("NT", 4000, 3000, 704), ("NT", 16384, 3072, 768), ("NT", 8192, 8192, 8192),
         ("NN", 4096, 3072, 768), ("NN", 4000, 3000, 704), ("NN", 16384, 3072, 768), ("NN", 16384, 768, 3072), ("NN", 8192, 8192, 8192),
         ("TN", 768, 3072, 16384), ("TN", 3072, 768, 16384), ("TN", 2304, 768, 16384), ("TN", 768, 768, 16384), ("TN", 760, 3000, 4000),
         ("TN", 4096, 4096, 4096), ("TN", 8192, 8192, 8192)]
for (lay, M, N, K) in cases:
    if lay == "NT":
        a, b = ints((M, K)), ints((N, K)); want = a.float() @ b.float().T; L, lda, ldb = DM_NT, K, K
    elif lay == "NN":
        a, b = ints((M, K)), ints((K, N)); want = a.float() @ b.float(); L, lda, ldb = DM_NN, K, N
    else:
        a, b = ints((K, M)), ints((K, N)); want = a.float().T @ b.float(); L, lda, ldb = DM_TN, M, N
    odt = torch.float32 if lay == "TN" else torch.bfloat16
    out = torch.full((M, N), 7.0, device=dev, dtype=odt)
    ops.gemm(L, a, b, out, M, N, K, lda=lda, ldb=ldb, ldc=N)
    ok = torch.equal(out, want.to(odt))
    dt = timeit(lambda: ops.gemm(L, a, b, out, M, N, K, lda=lda, ldb=ldb, ldc=N))
    print(f"{lay} {M}x{N}x{K}: exact={ok}  {dt*1e6:8.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TFLOP/s", flush=True)
    if not ok:
        bad = (out != want.to(odt)).nonzero()
        print("  mismatches:", bad.shape[0], "first:", bad[:6].tolist(), "got", out[tuple(bad[0])].item(), "want", want[tuple(bad[0])].item())
