"""Persistent 256x192 kernel (DM_GEMM_P192) against the previous routing on the stage-0 forward / dgrad shapes, cold-ish operands."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from deepmerge_amd import ops
from deepmerge_amd._lib import DM_NT, DM_NN
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(1)
def rnd(shape, dt=torch.bfloat16): return torch.randn(shape, device=dev, generator=g).to(dt)
R, IT = 3, 30
KEY = os.environ.get("KERNEL", "DM_GEMM_W4")           # or DM_GEMM_P192
os.environ["DM_GEMM_W4"] = "0"; os.environ["DM_GEMM_P192"] = "0"
T = int(os.environ.get("TOKENS", 16384))
def timeit(run):
    for i in range(6): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(IT): run(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / IT * 1e-3
def case(name, layout, M, N, K):
    sa, sb = ((M, K), (N, K)) if layout == DM_NT else ((M, K), (K, N))
    sets = [(rnd(sa), rnd(sb), torch.empty((M, N), device=dev, dtype=torch.bfloat16)) for _ in range(R)]
    def run(i):
        a, b, o = sets[i % R]
        ops.gemm(layout, a, b, o, M, N, K, lda=sa[1], ldb=sb[1], ldc=N)
    res = []
    for mode in ("0", "2"):
        os.environ[KEY] = mode
        res.append(timeit(run))
    os.environ[KEY] = "0"
    for dbg in os.environ.get("ABLATE", "").split():
        os.environ["DM_P192_DEBUG"] = dbg
        print(f"      debug={dbg}: {timeit(run)*1e6:7.1f} us", flush=True)
    os.environ["DM_P192_DEBUG"] = "0"
    fl = 2.0 * M * N * K
    print(f"{name:12s} {M:6d}x{N:5d}x{K:6d}  before {res[0]*1e6:7.1f} us {fl/res[0]/1e12:6.0f} TF/s | new {res[1]*1e6:7.1f} us {fl/res[1]/1e12:6.0f} TF/s | ratio {res[1]/res[0]:5.2f}", flush=True)
    return res
tb = tp = 0.0
for nm, N_out, K_in in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    for kind, layout, (M, N, K) in (("fwd", DM_NT, (T, N_out, K_in)), ("dgrad", DM_NN, (T, K_in, N_out))):
        a, b = case(f"{nm}.{kind}", layout, M, N, K)
        tb += a; tp += b
print(f"sum: before {tb*1e6:.0f} us, new {tp*1e6:.0f} us")
