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
TABLE = os.environ.get("TABLE") == "1"          # the table-in-LDS form (8 waves, (head, sample) units as runs): the headline model's stage 0
if TABLE:
    cube = (N // 64, 8, 8)
    table = torch.randn(((2 * cube[0] - 1) * 225, H), device=dev, generator=g) * 0.3
for _ in range(5):
    if TABLE:
        ops.attention_fwd_relpos(qkv, table, cube, B, N, H, D, 0.125)
    else:
        ops.attention_fwd(qkv, bias, B, N, H, D, 0.125)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(512 * 8 * 8, dtype=np.uint64)
rc = lib.dm_debug_q32_stamps(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(512, 8, 8).astype(np.int64)
live = t[:, 0, 0] > 0
t = t[live]
names = ["wait vmcnt", "barrier", "top..QK(0)", "tiles 0-3", "tiles 4-7", "epilogue", "loop tail -> next top"]
print(f"B={B} N={N} bias={bias is not None} table={TABLE}: {live.sum()} workgroups stamped (cycles of s_memtime = 100 MHz ticks? see ratio below)")
for smp in range(min(8, 8)):
    ok = t[:, smp, 0] > 0
    if ok.sum() == 0:
        break
    d = [np.median(t[ok, smp, i + 1] - t[ok, smp, i]) for i in range(6)]
    nxt = np.median(t[ok, smp + 1, 0] - t[ok, smp, 6]) if smp + 1 < 8 and (t[ok, smp + 1, 0] > 0).all() else float("nan")
    tot = np.median(t[ok, smp, 6] - t[ok, smp, 0])
    print(f" sample {smp}: " + "  ".join(f"{n} {int(v)}" for n, v in zip(names, d)) + f"  | total {int(tot)}  to-next {nxt}")
# chip-wide 100 MHz clock (s_memrealtime): kernel entry / start of the unit loop / kernel exit of every workgroup, in microseconds from the first entry
rt = t[:, :3, 7].astype(np.float64) / 100.0
ok = (rt > 0).all(axis=1)
if ok.any():
    rt = rt[ok]; e0 = rt[:, 0].min()
    print(f" realtime clock, us from the first workgroup's entry: entries {np.median(rt[:, 0] - e0):.2f} (max {np.max(rt[:, 0] - e0):.2f}); "
          f"unit loop starts {np.median(rt[:, 1] - e0):.2f} (max {np.max(rt[:, 1] - e0):.2f}); exits median {np.median(rt[:, 2] - e0):.2f}, last {np.max(rt[:, 2] - e0):.2f}")
    print(f" per workgroup: entry -> loop {np.median(rt[:, 1] - rt[:, 0]):.2f} us, loop -> exit {np.median(rt[:, 2] - rt[:, 1]):.2f} us")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(20):
    (ops.attention_fwd_relpos(qkv, table, cube, B, N, H, D, 0.125) if TABLE else ops.attention_fwd(qkv, bias, B, N, H, D, 0.125))
ev[1].record(); torch.cuda.synchronize()
print(f" 20 eager launches back to back: {ev[0].elapsed_time(ev[1]) * 1000 / 20:.1f} us per launch (stamped build)")
