"""Per-phase cycle counts of the 32-row attention forward (library built with EXTRA=-DDMQ_STAMP): median over workgroups, per sample."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deepmerge_amd import ops, _lib
dev = "cuda:0"
B, N, H, D = int(os.environ.get("B", 64)), int(os.environ.get("N", 256)), 12, 64
g = torch.Generator(device=dev); g.manual_seed(0)
qkv = torch.randn((B, N, 3, H, D), device=dev, generator=g).to(torch.bfloat16)
bias = None if os.environ.get("NOBIAS") == "1" else torch.randn((H, N, N), device=dev, generator=g) * 0.3
for _ in range(5):
    ops.attention_fwd(qkv, bias, B, N, H, D, 0.125)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(512 * 8 * 8, dtype=np.uint64)
rc = lib.dm_debug_q32_stamps(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(512, 8, 8).astype(np.int64)
live = t[:, 0, 0] > 0
t = t[live]
names = ["wait vmcnt", "barrier", "top..QK(0)", "tiles 0-3", "tiles 4-7", "epilogue", "loop tail -> next top"]
print(f"B={B} N={N} bias={bias is not None}: {live.sum()} workgroups stamped (cycles of s_memtime = 100 MHz ticks? see ratio below)")
for smp in range(min(8, 8)):
    ok = t[:, smp, 0] > 0
    if ok.sum() == 0:
        break
    d = [np.median(t[ok, smp, i + 1] - t[ok, smp, i]) for i in range(6)]
    nxt = np.median(t[ok, smp + 1, 0] - t[ok, smp, 6]) if smp + 1 < 8 and (t[ok, smp + 1, 0] > 0).all() else float("nan")
    tot = np.median(t[ok, smp, 6] - t[ok, smp, 0])
    print(f" sample {smp}: " + "  ".join(f"{n} {int(v)}" for n, v in zip(names, d)) + f"  | total {int(tot)}  to-next {nxt}")
first = t[:, 0, 0].min(); last = t[:, :, 6].max()
print(f" kernel span (first top .. last epilogue): {last - first} ticks")
