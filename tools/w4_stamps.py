"""Per-phase cycle counts of the 4-wave GEMM kernel's folded (three-set) K step (library built with EXTRA=-DDM_W4_STAMP):
median over workgroups and steps.  usage: python tools/w4_stamps.py [TN|NN|NT]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["DM_GEMM_W4"] = "2"; os.environ["DM_GEMM_W4_TN"] = "2"; os.environ["DM_GEMM_256"] = "0"; os.environ["DM_GEMM_RING"] = "0"
from deepmerge_amd import ops, _lib
from deepmerge_amd._lib import DM_NT, DM_NN, DM_TN
dev = "cuda:0"
layout = sys.argv[1] if len(sys.argv) > 1 else "TN"
lay = {"NT": DM_NT, "NN": DM_NN, "TN": DM_TN}[layout]
M, N, K = {"TN": (768, 3072, 16384), "NN": (16384, 768, 3072), "NT": (16384, 768, 3072)}[layout]
def planes(r, c): return ops.Planes(torch.randn(2, r, c, device=dev).bfloat16())
A = planes(*((K, M) if layout == "TN" else (M, K))); B = planes(*((N, K) if layout == "NT" else (K, N)))
C = torch.empty(M, N, device=dev)
for _ in range(3):
    ops.gemm(lay, A, B, C, M, N, K)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.gemm(lay, A, B, C, M, N, K); e1.record(); torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(64 * 16 * 8, dtype=np.uint64)
lib.dm_debug_w4_stamps(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(64, 16, 8).astype(np.int64)
ok = t[:, :, 0] > 0
names = ["set 1 (hi.hi + Y staging + B lo frags)", "set 2 (hi.lo + A lo frags)", "lgkmcnt(0)", "barrier", "set 3 (lo.hi + X staging + next frags)"]
print(f"{layout} {M}x{N}x3*{K}: launch {e0.elapsed_time(e1) * 1e3:.1f} us (incl. reduce); {ok.sum()} steps stamped; ticks of s_memtime (100 MHz)")
d = [np.median((t[:, :, i + 1] - t[:, :, i])[ok]) for i in range(5)]
step = np.median((t[:, 1:, 0] - t[:, :-1, 0])[ok[:, 1:] & ok[:, :-1]])
for n, v in zip(names, d):
    print(f"  {n:45s} {v:7.1f} ticks")
print(f"  step to step {step:.1f} ticks = {step * 10:.0f} ns; three sets of 48 MFMAs need 2304 cycles = 960 ns at 2.4 GHz")
