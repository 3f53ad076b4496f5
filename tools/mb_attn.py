"""Attention forward / backward timing at the encoder's stage-0 shape (B samples x 12 heads x N tokens, bf16, with bias)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deepmerge_amd import ops
dev = "cuda:0"
B, N, H, D = int(os.environ.get("B", 64)), int(os.environ.get("N", 256)), 12, 64
g = torch.Generator(device=dev); g.manual_seed(0)
qkv = torch.randn((B, N, 3, H, D), device=dev, generator=g).to(torch.bfloat16)
bias = None if os.environ.get("NOBIAS") == "1" else torch.randn((H, N, N), device=dev, generator=g) * 0.3      # ViT (config 3) has no bias table
def timeit(f, n=30, reps=5):
    """Device time per call: n calls captured into one hipGraph and replayed (a Python call + two allocations per launch cost ~15 us of
    host time, more than the faster kernels take), best of `reps` replays."""
    for _ in range(3): f()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        f(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); g.replay(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / n)
    return best
t = timeit(lambda: ops.attention_fwd(qkv, bias, B, N, H, D, 0.125))
print(f"fwd B={B} N={N}: {t*1e6:7.1f} us  {4.0*B*H*N*N*D/t/1e12:6.1f} TFLOP/s")
if bias is not None and N % 64 == 0 and ops.relpos_inkernel(B, N, H, D, (N // 64, 8, 8), torch.bfloat16):
    cube = (N // 64, 8, 8)
    table = torch.randn(((2 * cube[0] - 1) * 225, H), device=dev, generator=g) * 0.3
    t = timeit(lambda: ops.attention_fwd_relpos(qkv, table, cube, B, N, H, D, 0.125))
    print(f"fwd (table in LDS) B={B} N={N}: {t*1e6:7.1f} us  {4.0*B*H*N*N*D/t/1e12:6.1f} TFLOP/s")
if os.environ.get("X3") == "1":
    q32 = qkv.float()
    t = timeit(lambda: ops.attention_fwd(q32, bias, B, N, H, D, 0.125))
    print(f"fwd fp32 kernels B={B} N={N}: {t*1e6:7.1f} us")
    cube3 = (N // 64, 8, 8) if (bias is not None and N % 64 == 0) else None
    tab3 = None if cube3 is None else torch.randn(((2 * cube3[0] - 1) * 225, H), device=dev, generator=g) * 0.3
    if ops.attention_split_ok(B, N, H, D, cube3) and (bias is None or cube3 is not None):
        t = timeit(lambda: ops.attention_fwd_split(q32, tab3, cube3, B, N, H, D, 0.125))
        print(f"fwd split-bf16 kernels B={B} N={N}: {t*1e6:7.1f} us  {3*4.0*B*H*N*N*D/t/1e12:6.1f} TFLOP/s (3 MFMAs per product, incl. the split pass)")
        o3, l3, hi3, lo3 = ops.attention_fwd_split(q32, tab3, cube3, B, N, H, D, 0.125)
        do3 = torch.randn_like(o3)
        idx3 = None if cube3 is None else torch.randint(0, tab3.shape[0], (N, N), device=dev, dtype=torch.int32)
        b32 = None if bias is None else bias
        t = timeit(lambda: ops.attention_bwd(q32, b32, o3, do3, l3, B, N, H, D, 0.125, idx3, 0 if tab3 is None else tab3.shape[0]), n=10)
        print(f"bwd fp32 kernels B={B} N={N}: {t*1e6:7.1f} us")
        t = timeit(lambda: ops.attention_bwd_split(hi3, lo3, tab3, cube3, o3, do3, l3, B, N, H, D, 0.125, idx3, 0 if tab3 is None else tab3.shape[0]), n=10)
        print(f"bwd split-bf16 kernels B={B} N={N}: {t*1e6:7.1f} us  {3*10.0*B*H*N*N*D/t/1e12:6.1f} TFLOP/s (incl. the dout split pass)")
if os.environ.get("FWD_ONLY") == "1":
    sys.exit(0)
out, lse = ops.attention_fwd(qkv, bias, B, N, H, D, 0.125)
dout = torch.randn_like(out)
NB = (2 * (N // 64) - 1) * 225 if N % 64 == 0 else 1575
idx = None if bias is None else torch.randint(0, NB, (N, N), device=dev, dtype=torch.int32)
t = timeit(lambda: ops.attention_bwd(qkv, bias, out, dout, lse, B, N, H, D, 0.125, idx, NB if bias is not None else 0))
print(f"bwd B={B} N={N}: {t*1e6:7.1f} us  {10.0*B*H*N*N*D/t/1e12:6.1f} TFLOP/s (incl. slab alloc)")
if bias is not None and N % 64 == 0 and ops.relpos_inkernel(B, N, H, D, (N // 64, 8, 8), torch.bfloat16):
    t = timeit(lambda: ops.attention_bwd(qkv, None, out, dout, lse, B, N, H, D, 0.125, idx, table.shape[0], table=table, cube=cube))
    print(f"bwd (table in LDS) B={B} N={N}: {t*1e6:7.1f} us  {10.0*B*H*N*N*D/t/1e12:6.1f} TFLOP/s (incl. slab alloc)")
