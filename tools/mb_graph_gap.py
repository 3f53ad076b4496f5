"""Per-node cost of a captured chain of launches (graph replay), tiny kernels and a mid-sized GEMM: what a launch boundary costs inside the
captured step.  usage: python tools/mb_graph_gap.py"""
import torch, time
from deepmerge_amd import ops
from deepmerge_amd._lib import DM_NT
dev = "cuda:0"
x = torch.randn(64, device=dev)
def timed(fn, n_nodes, label, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label:60s} {dt*1e6:9.1f} us per replay, {dt*1e6/n_nodes:7.2f} us per node", flush=True)
    return dt
def tiny(n):
    def f():
        for _ in range(n): ops.cast(x, torch.bfloat16)
    return f
timed(tiny(100), 100, "100 x dm_cast(64 elements)")
timed(tiny(400), 400, "400 x dm_cast(64 elements)")
def tiny_torch(n):
    def f():
        y = x
        for _ in range(n): y = y + 1
    return f
timed(tiny_torch(100), 100, "100 x torch add(64 elements)")
M, N, K = 16384, 768, 768
a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16(); c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
def gemms(n, pad):
    def f():
        for _ in range(n):
            ops.gemm(DM_NT, a, w, c, M, N, K)
            for _ in range(pad): ops.cast(x, torch.bfloat16)
    return f
t0 = timed(gemms(40, 0), 40, "40 x gemm 16384x768x768")
t1 = timed(gemms(40, 1), 80, "40 x (gemm + 1 tiny)")
t2 = timed(gemms(40, 4), 200, "40 x (gemm + 4 tiny)")
print(f"extra per tiny launch next to a GEMM: {(t1-t0)/40*1e6:.2f} us (1), {(t2-t0)/160*1e6:.2f} us (4)")
