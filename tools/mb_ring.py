"""Ablations of the ring GEMM kernel (DM_RING_DEBUG bits) on one shape, cold operands."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["DM_GEMM_RING"] = "2"; os.environ["DM_GEMM_256"] = "0"
import torch
from deepmerge_amd import ops
from deepmerge_amd._lib import DM_NT
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(1)
def rnd(shape, dt=torch.bfloat16): return torch.randn(shape, device=dev, generator=g).to(dt)
R = 3
def bench(M, N, K, dbg, wm):
    os.environ["DM_RING_DEBUG"] = str(dbg); os.environ["DM_GEMM_RING_WM"] = str(wm)
    sets = [(rnd((M, K)), rnd((N, K)), torch.empty((M, N), device=dev, dtype=torch.bfloat16)) for _ in range(R)]
    def run(i):
        a, b, o = sets[i % R]; ops.gemm(DM_NT, a, b, o, M, N, K, lda=K, ldb=K, ldc=N)
    for i in range(6): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(30): run(i)
    e1.record(); torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / 30 * 1e-3
    print(f"{M}x{N}x{K} wm={wm} debug={dbg}: {dt*1e6:7.1f} us {2.0*M*N*K/dt/1e12:7.1f} TF/s", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "pmc":
    bench(16384, 3072, 768, 0, 8)
    bench(16384, 768, 3072, 0, 8)
    os.environ["DM_GEMM_RING"] = "0"; os.environ["DM_GEMM_256"] = "2"
    bench(16384, 3072, 768, 0, 8)
    bench(16384, 768, 3072, 0, 8)
    sys.exit(0)
for (M, N, K) in ((16384, 3072, 768), (16384, 768, 3072)):
    for wm in (8,):
        for dbg in (0, 16, 6, 22, 5):
            bench(M, N, K, dbg, wm)
