"""Correctness + timing of the 256x256 NT pipeline vs the 128x128 kernel (run once per DM_GEMM_256 setting)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deepmerge_amd import ops
from deepmerge_amd._lib import DM_NT, DM_EPI_GELU

dev = "cuda:0"
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

g = torch.Generator(device=dev); g.manual_seed(1)
for (M, N, K) in [(4096, 3072, 768), (4000, 3000, 704), (16384, 768, 768), (16384, 2304, 768), (16384, 3072, 768), (16384, 768, 3072), (8192, 8192, 8192), (4096, 4096, 4096)]:
    a = torch.randint(-2, 3, (M, K), device=dev, generator=g).to(torch.bfloat16)
    b = torch.randint(-2, 3, (N, K), device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    ops.gemm(DM_NT, a, b, out, M, N, K, lda=K, ldb=K, ldc=N)
    want = (a.float() @ b.float().T).to(torch.bfloat16)
    ok = torch.equal(out, want)
    ar = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16); br = torch.randn((N, K), device=dev, generator=g).to(torch.bfloat16)
    dt = timeit(lambda: ops.gemm(DM_NT, ar, br, out, M, N, K, lda=K, ldb=K, ldc=N))
    print(f"{M}x{N}x{K}: exact={ok}  {dt*1e6:8.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TFLOP/s", flush=True)
    if not ok:
        bad = (out != want).nonzero()
        print("  mismatches:", bad.shape[0], "first:", bad[:5].tolist())
