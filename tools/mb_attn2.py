"""Pipelined attention backward with / without bias (how much do the bias registers of the dK/dV kernel cost?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deepmerge_amd import ops, _lib
dev = "cuda:0"
B, N, H, D = 64, 256, 12, 64
g = torch.Generator(device=dev); g.manual_seed(0)
qkv = torch.randn((B, N, 3, H, D), device=dev, generator=g).to(torch.bfloat16)
bias = torch.randn((H, N, N), device=dev, generator=g) * 0.3
idx = torch.randint(0, 1575, (N, N), device=dev, dtype=torch.int32)
lib = _lib.lib()
for name, b_, ix in (("with bias + dS slab", bias, idx), ("with bias, no slab", bias, None), ("no bias", None, None)):
    out, lse = ops.attention_fwd(qkv, b_, B, N, H, D, 0.125)
    dout = torch.randn_like(out)
    f = lambda: ops.attention_bwd(qkv, b_, out, dout, lse, B, N, H, D, 0.125, ix, 1575 if ix is not None else 0)
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(30): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 30
    print(f"bwd {name:22s}: {dt*1e6:7.1f} us")
