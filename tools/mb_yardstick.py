"""Yardstick: the stage-0 GEMM shapes of the headline step (tokens M = 16384, C = 768, hidden 3072) through dm_gemm (default
routing) and through torch.matmul (hipBLASLt / rocBLAS on this image), cold-ish operands (3 rotating sets).  Not part of the
product: it tells how far the hand-written kernels are from the vendor library on the same box."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from deepmerge_amd import ops
from deepmerge_amd._lib import DM_NT, DM_NN, DM_TN
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(1)
def rnd(shape, dt=torch.bfloat16): return torch.randn(shape, device=dev, generator=g).to(dt)
R, IT = 3, 30
T = int(os.environ.get("TOKENS", 16384))
PAD = int(os.environ.get("PAD", 0))        # extra elements per operand / output row: leading dimensions off the power-of-two-ish strides
def padded(shape, dt=torch.bfloat16, fill=True):
    full = rnd((shape[0], shape[1] + PAD), dt) if fill else torch.empty((shape[0], shape[1] + PAD), device=dev, dtype=dt)
    return full[:, :shape[1]]

def timeit(run):
    for i in range(6): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(IT): run(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / IT * 1e-3

def case(name, layout, M, N, K):
    if layout == DM_NT:   sa, sb = (M, K), (N, K)
    elif layout == DM_NN: sa, sb = (M, K), (K, N)
    else:                 sa, sb = (K, M), (K, N)
    cdt = torch.float32 if layout == DM_TN else torch.bfloat16
    sets = [(padded(sa), padded(sb), padded((M, N), cdt, False), padded((M, N), torch.bfloat16, False)) for _ in range(R)]
    def mine(i):
        a, b, o, _ = sets[i % R]
        ops.gemm(layout, a, b, o, M, N, K, lda=sa[1] + PAD, ldb=sb[1] + PAD, ldc=N + PAD)
    def lib(i):
        a, b, _, o = sets[i % R]
        if layout == DM_NT:   torch.matmul(a, b.t(), out=o)
        elif layout == DM_NN: torch.matmul(a, b, out=o)
        else:                 torch.matmul(a.t(), b, out=o)
    tm, tl = timeit(mine), timeit(lib)
    if layout == DM_TN and os.environ.get("W4TN_AB"):
        os.environ["DM_GEMM_W4_TN"] = "0"
        t0 = timeit(mine)
        os.environ["DM_GEMM_W4_TN"] = "2"
        t2 = timeit(mine)
        os.environ.pop("DM_GEMM_W4_TN")
        print(f"      wgrad A/B: 256x256 pipeline {t0*1e6:7.1f} us | 4-wave kernel {t2*1e6:7.1f} us", flush=True)
    fl = 2.0 * M * N * K
    print(f"{name:12s} {M:6d}x{N:5d}x{K:6d}  dm_gemm {tm*1e6:7.1f} us {fl/tm/1e12:6.0f} TF/s | torch.matmul {tl*1e6:7.1f} us {fl/tl/1e12:6.0f} TF/s | ratio {tm/tl:5.2f}", flush=True)
    return tm, tl

tot_m = tot_l = 0.0
for nm, N_out, K_in in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    for kind, layout, (M, N, K) in (("fwd", DM_NT, (T, N_out, K_in)), ("dgrad", DM_NN, (T, K_in, N_out)), ("wgrad", DM_TN, (N_out, K_in, T))):
        a, b = case(f"{nm}.{kind}", layout, M, N, K)
        tot_m += a; tot_l += b
print(f"sum over one block's 12 GEMMs: dm_gemm {tot_m*1e6:.0f} us, torch.matmul {tot_l*1e6:.0f} us")
