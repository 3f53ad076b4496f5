"""Do the encoder's GEMMs slow down when their operands are cache-cold (as in the real step)?  Rotates over R operand sets
(R = 1: warm; R = 8: > 256 MB of MALL) with random-normal data; prints us per launch."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deepmerge_amd import ops
from deepmerge_amd._lib import DM_NT, DM_NN, DM_TN
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(1)
def rnd(shape): return torch.randn(shape, device=dev, generator=g).to(torch.bfloat16)
cases = [("NT", 16384, 3072, 768), ("NT", 16384, 768, 3072), ("NT", 16384, 2304, 768), ("NN", 16384, 3072, 768), ("TN", 768, 3072, 16384), ("TN", 3072, 768, 16384)]
for lay, M, N, K in cases:
    for R in (1, 2, 3, 8):
        sets = []
        for _ in range(R):
            if lay == "NT": a, b, L, lda, ldb = rnd((M, K)), rnd((N, K)), DM_NT, K, K
            elif lay == "NN": a, b, L, lda, ldb = rnd((M, K)), rnd((K, N)), DM_NN, K, N
            else: a, b, L, lda, ldb = rnd((K, M)), rnd((K, N)), DM_TN, M, N
            out = torch.empty((M, N), device=dev, dtype=torch.float32 if lay == "TN" else torch.bfloat16)
            sets.append((a, b, out))
        def run(i):
            a, b, out = sets[i % R]
            ops.gemm(L, a, b, out, M, N, K, lda=lda, ldb=ldb, ldc=N)
        for i in range(2 * R): run(i)
        torch.cuda.synchronize(); t = time.perf_counter()
        n = 40
        for i in range(n): run(i)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
        print(f"{lay} {M}x{N}x{K} sets={R}: {dt*1e6:7.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TFLOP/s", flush=True)
