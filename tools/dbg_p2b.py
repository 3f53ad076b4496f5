import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["DM_GEMM_P2"] = "2"
from deepmerge_amd import ops
from deepmerge_amd._lib import DM_NT
DEV = "cuda:0"
g = torch.Generator(device=DEV); g.manual_seed(3)
M, N, K = 16384, 3072, 768
sets = [(torch.randn((M, K), device=DEV, generator=g).to(torch.bfloat16), torch.randn((N, K), device=DEV, generator=g).to(torch.bfloat16), torch.empty((M, N), device=DEV, dtype=torch.bfloat16)) for _ in range(3)]
def run(i):
    a, b, o = sets[i % 3]; ops.gemm(DM_NT, a, b, o, M, N, K, lda=K, ldb=K, ldc=N)
for dbg in (0, 1, 3, 5, 7, 2, 4):
    os.environ["DM_P2_DEBUG"] = str(dbg)
    for i in range(6): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(30): run(i)
    e1.record(); torch.cuda.synchronize()
    print(f"debug={dbg} ({'noepi ' if dbg&1 else ''}{'nodma ' if dbg&2 else ''}{'nomfma' if dbg&4 else ''}): {e0.elapsed_time(e1)/30*1e3:7.1f} us", flush=True)
