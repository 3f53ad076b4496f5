"""Does the 256x256 pipeline reach its warm-cache rate on cold operands when the tile count fills the CUs exactly?
(N = 1024 -> 256 tiles at M = 16384; compare N = 768 -> 192 tiles.)  Epilogue: bias + fp32 residual, fp32 out (the fc2 form)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deepmerge_amd import ops
from deepmerge_amd._lib import DM_NT
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(1)
def rnd(shape): return torch.randn(shape, device=dev, generator=g).to(torch.bfloat16)
for (M, N, K) in [(16384, 768, 3072), (16384, 1024, 3072), (16384, 2304, 768), (16384, 2048, 768), (16384, 3072, 768)]:
    R = 6
    sets = [(rnd((M, K)), rnd((N, K)), torch.randn((M, N), device=dev), torch.empty((M, N), device=dev), torch.randn(N, device=dev)) for _ in range(R)]
    def run(i):
        a, b, res, out, bias = sets[i % R]
        ops.gemm(DM_NT, a, b, out, M, N, K, lda=K, ldb=K, ldc=N, bias=bias, residual=res)
    for i in range(2 * R): run(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 36
    for i in range(n): run(i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"NT {M}x{N}x{K} +bias+residual fp32 (cold): {dt*1e6:7.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TFLOP/s", flush=True)
