"""K-loop ablations of the 4-wave persistent kernel (-DDM_W4_ABLATE build): epilogue off, then MFMA / DMA / fragment reads off."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["DM_GEMM_W4"] = "2"
import torch
from deepmerge_amd import ops
from deepmerge_amd._lib import DM_NT
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(1)
def rnd(shape): return torch.randn(shape, device=dev, generator=g).to(torch.bfloat16)
R, IT = 3, 30
PAD = int(os.environ.get("PAD", "0"))       # extra elements per row of A (leading dimension K + PAD): DRAM / L2 channel spread
def bench(M, N, K):
    sets = [(rnd((M, K + PAD)), rnd((N, K)), torch.empty((M, N), device=dev, dtype=torch.bfloat16)) for _ in range(R)]
    def run(i):
        a, b, o = sets[i % R]; ops.gemm(DM_NT, a, b, o, M, N, K, lda=K + PAD, ldb=K, ldc=N)
    for dbg in [int(x) for x in os.environ.get('DBGS', '0 1 5 13 29 61 33 17 9 45').split()]:
        os.environ["DM_W4_DEBUG"] = str(dbg)
        for i in range(6): run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(IT): run(i)
        e1.record(); torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) / IT * 1e-3
        print(f"{M}x{N}x{K} debug={dbg:2d} ({'noepi ' if dbg & 1 else ''}{'noglobal ' if dbg & 4 else ''}{'noldswrite ' if dbg & 8 else ''}{'nofrag ' if dbg & 16 else ''}{'nomfma' if dbg & 32 else ''}): {dt*1e6:7.1f} us", flush=True)
bench(16384, 3072, 768)
bench(16384, 768, 3072)
