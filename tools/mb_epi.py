"""What do the fused epilogues and the tile choice cost on the step's shapes, with cache-cold operands?
    DM_GEMM_256=0|2 python tools/mb_epi.py          (0: register-staged tiles only; 2: force the 256x256 pipeline)
Rotates over R operand sets so every launch reads operands from HBM, as in the real step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deepmerge_amd import ops
from deepmerge_amd._lib import DM_NT, DM_NN, DM_TN, DM_EPI_NONE, DM_EPI_GELU, DM_EPI_GELU_GRAD, DM_EPI_MUL
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(1)
def rnd(shape, dt=torch.bfloat16): return torch.randn(shape, device=dev, generator=g).to(dt)
R = 3

def bench(name, lay, M, N, K, variant, tile=None):
    sets = []
    for _ in range(R):
        if lay == DM_NT: a, b, lda, ldb = rnd((M, K)), rnd((N, K)), K, K
        elif lay == DM_NN: a, b, lda, ldb = rnd((M, K)), rnd((K, N)), K, N
        else: a, b, lda, ldb = rnd((K, M)), rnd((K, N)), M, N
        kw = {}
        cdt = torch.bfloat16
        if variant == "gelu_grad": kw = dict(epilogue=DM_EPI_GELU_GRAD, aux=torch.empty((M, N), device=dev, dtype=torch.bfloat16), bias=rnd((N,), torch.float32))
        elif variant == "gelu": kw = dict(epilogue=DM_EPI_GELU, bias=rnd((N,), torch.float32))
        elif variant == "bias": kw = dict(bias=rnd((N,), torch.float32))
        elif variant == "mul": kw = dict(epilogue=DM_EPI_MUL, aux=rnd((M, N)))
        elif variant == "res_f32": kw = dict(bias=rnd((N,), torch.float32), residual=rnd((M, N), torch.float32)); cdt = torch.float32
        elif variant == "f32": cdt = torch.float32
        out = torch.empty((M, N), device=dev, dtype=cdt)
        sets.append((a, b, out, lda, ldb, kw))
    if tile: os.environ["DM_GEMM_FORCE_TILE"] = str(tile)
    else: os.environ.pop("DM_GEMM_FORCE_TILE", None)
    def run(i):
        a, b, out, lda, ldb, kw = sets[i % R]
        ops.gemm(lay, a, b, out, M, N, K, lda=lda, ldb=ldb, ldc=N, **kw)
    for i in range(2 * R): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for i in range(n): run(i)
    e1.record(); torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / n * 1e-3
    print(f"{name:10s} {['NT','NN','TN'][lay]} {M}x{N}x{K} {variant:9s} tile={tile or 'auto':>4}: {dt*1e6:7.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TF/s", flush=True)

which = sys.argv[1] if len(sys.argv) > 1 else "epi"
if which == "w4set":          # the multi-tile K = 768 products (w4 under DM_GEMM_W4=3) with their step epilogues
    bench("fc1", DM_NT, 16384, 3072, 768, "gelu_grad")
    bench("fc1", DM_NT, 16384, 3072, 768, "none")
    bench("dfc2", DM_NN, 16384, 3072, 768, "mul")
    bench("qkv", DM_NT, 16384, 2304, 768, "bias")
elif which == "proj":         # the one-round N = 768 products at 16384 tokens: tile choices
    for tile in (None, 128, 64):
        bench("proj", DM_NT, 16384, 768, 768, "res_f32", tile)
        bench("projd", DM_NN, 16384, 768, 768, "none", tile)
elif which == "epi":
    for v in ("none", "bias", "gelu", "gelu_grad"):
        for tile in (None, 128):
            bench("fc1", DM_NT, 16384, 3072, 768, v, tile)
    for v in ("none", "mul"):
        for tile in (None, 128):
            bench("dfc2", DM_NN, 16384, 3072, 768, v, tile)
    for v in ("none", "f32", "res_f32"):
        for tile in (None, 128):
            bench("fc2", DM_NT, 16384, 768, 3072, v, tile)
            bench("proj", DM_NT, 16384, 768, 768, v, tile)
    for tile in (None, 128):
        bench("qkv", DM_NT, 16384, 2304, 768, "bias", tile)
else:
    for M in (4096, 1024):
        for tile in (64, 128):
            bench("qkv", DM_NT, M, 2304, 768, "bias", tile)
            bench("proj", DM_NT, M, 768, 768, "res_f32", tile)
            bench("fc1", DM_NT, M, 3072, 768, "gelu_grad", tile)
            bench("fc2", DM_NT, M, 768, 3072, "res_f32", tile)
            bench("dfc2", DM_NN, M, 3072, 768, "mul", tile)
            bench("dfc1", DM_NN, M, 768, 3072, "none", tile)
            bench("dqkv", DM_NN, M, 768, 2304, "none", tile)
            bench("wfc1", DM_TN, 3072, 768, M, "f32", tile)
            bench("wfc2", DM_TN, 768, 3072, M, "f32", tile)
            bench("wqkv", DM_TN, 2304, 768, M, "f32", tile)
