"""Cycle counts inside the 4-wave GEMM kernel (library built with EXTRA=-DDM_W4_STAMP; s_memtime = shader-clock cycles on this hardware).
    python tools/w4_stamps.py [TN|NN|NT]            the folded (three-set) K step: cycles per MFMA set, wait, barrier; median over workgroups and steps
    python tools/w4_stamps.py [TN|NN|NT] phases     a plain bf16 product: every workgroup's entry / set-up / prologue / K loop / epilogue of its first tile"""
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
if len(sys.argv) > 2 and sys.argv[2] == "phases":
    a = torch.randn(*((K, M) if layout == "TN" else (M, K)), device=dev).bfloat16()
    b = torch.randn(*((N, K) if layout == "NT" else (K, N)), device=dev).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.float32 if layout == "TN" else torch.bfloat16)
    for _ in range(3):
        ops.gemm(lay, a, b, c, M, N, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm(lay, a, b, c, M, N, K); e1.record(); torch.cuda.synchronize()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    buf = np.zeros(512 * 8, dtype=np.uint64)
    lib.dm_debug_w4_kstamps(buf.ctypes.data_as(ctypes.c_void_p))
    t = buf.reshape(512, 8).astype(np.int64)
    G = int((t[:, 4] > 0).sum())
    t = t[:G]
    print(f"{layout} {M}x{N}x{K} bf16: launch {e0.elapsed_time(e1) * 1e3:.1f} us by hipEvents (one eager launch; incl. the slab reduction for TN); {G} workgroups stamped")
    for name, lo, hi in (("set-up (addresses, descriptors)", 0, 1), ("prologue (two K steps staged, fragments)", 1, 2),
                         ("K loop of the first tile", 2, 3), ("epilogue of the first tile (issue)", 3, 4), ("entry -> epilogue issued", 0, 4)):
        d = t[:, hi] - t[:, lo]
        print(f"  {name:50s} median {int(np.median(d)):9d}  min {d.min():9d}  max {d.max():9d} cycles  ({np.median(d) / 2400:.1f} us)")
    # s_memtime differs per XCD; slots 5 / 6 hold s_memrealtime (one 100 MHz clock for the chip) at entry / after the epilogue was issued
    ent, end = t[:, 5], t[:, 6]
    mhz = (t[:, 4] - t[:, 0]) / ((end - ent) / 100.0)
    print(f"  shader clock while the kernel runs (s_memtime cycles per s_memrealtime microsecond, per workgroup): median {np.median(mhz):.0f} MHz, min {mhz.min():.0f}, max {mhz.max():.0f}")
    print(f"  chip-wide (100 MHz clock): entries spread over {(ent.max() - ent.min()) / 100:.2f} us; first entry -> last epilogue issued {(end.max() - ent.min()) / 100:.2f} us; "
          f"ends spread over {(end.max() - end.min()) / 100:.2f} us")
    sys.exit(0)
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
print(f"{layout} {M}x{N}x3*{K}: launch {e0.elapsed_time(e1) * 1e3:.1f} us (incl. reduce); {ok.sum()} steps stamped; cycles of s_memtime")
d = [np.median((t[:, :, i + 1] - t[:, :, i])[ok]) for i in range(5)]
step = np.median((t[:, 1:, 0] - t[:, :-1, 0])[ok[:, 1:] & ok[:, :-1]])
for n, v in zip(names, d):
    print(f"  {n:45s} {v:7.1f} cycles")
print(f"  step to step {step:.1f} cycles = {step / 2.4:.0f} ns at 2.4 GHz; three sets of 48 MFMAs need 2304 cycles = 960 ns")
