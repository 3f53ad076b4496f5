import os, sys
os.environ["DM_PROF_SHAPES"] = "1"
sys.path.insert(0, "/root/repo")
import torch
from deepmerge_amd import ops, _lib
from deepmerge_amd._lib import DM_NT, DM_NN, DM_TN
dev = "cuda:0"
lib = _lib.lib()
def planes(r, c): return ops.Planes(torch.randn(2, r, c, device=dev).bfloat16())
def which(lay, M, N, K, out_pair, **kw):
    a_shape = (K, M) if lay == DM_TN else (M, K)
    b_shape = (N, K) if lay == DM_NT else (K, N)
    A, B = planes(*a_shape), planes(*b_shape)
    C = planes(M, N) if out_pair else torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev) if kw.pop("bias", False) else None
    lib.dm_prof_enable(1)
    ops.gemm(lay, A, B, C, M, N, K, bias=bias, **kw)
    torch.cuda.synchronize(); lib.dm_prof_enable(0)
    rows = (_lib.DmProfRow * 64)(); n = lib.dm_prof_collect(rows, 64)
    print([rows[i].name.decode() for i in range(n)])
for T in (16384, 61440):
    which(DM_NT, T, 2304, 768, True, bias=True)
    which(DM_NN, T, 768, 3072, False)
    which(DM_NN, T, 768, 768, False)
