import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["DM_GEMM_P2"] = "2"
from deepmerge_amd import ops
from deepmerge_amd._lib import DM_NT, DM_EPI_GELU, DM_EPI_GELU_GRAD
DEV = "cuda:0"
g = torch.Generator(device=DEV); g.manual_seed(3)
def check(M, N, K):
    a = torch.randint(-2, 3, (M, K), device=DEV, generator=g).to(torch.bfloat16)
    b = torch.randint(-2, 3, (N, K), device=DEV, generator=g).to(torch.bfloat16)
    bias = torch.randint(-3, 4, (N,), device=DEV, generator=g).float()
    ref = a.float() @ b.float().T
    out = torch.zeros((M, N), device=DEV, dtype=torch.bfloat16)
    ops.gemm(DM_NT, a, b, out, M, N, K, lda=K, ldb=K, ldc=N)
    bad = (out != ref.to(torch.bfloat16))
    print(M, N, K, "plain bad", int(bad.sum()), "of", bad.numel())
    if bad.any():
        idx = bad.nonzero(); print(idx[:6].tolist(), idx[-3:].tolist(), "rows", len(idx[:,0].unique()), "cols", len(idx[:,1].unique()))
        i, j = idx[0].tolist(); print("got", out[i, j].item(), "want", ref[i, j].item())
    out2 = torch.zeros((M, N), device=DEV, dtype=torch.bfloat16)
    ops.gemm(DM_NT, a, b, out2, M, N, K, lda=K, ldb=K, ldc=N, bias=bias)
    bad = (out2 != (ref + bias).to(torch.bfloat16)); print("  bias bad", int(bad.sum()))
    a2 = (a.float() * 0.125).to(torch.bfloat16)
    d = torch.zeros((M, N), device=DEV, dtype=torch.bfloat16); h = torch.zeros((M, N), device=DEV, dtype=torch.bfloat16)
    ops.gemm(DM_NT, a2, b, h, M, N, K, lda=K, ldb=K, ldc=N, bias=bias, epilogue=DM_EPI_GELU_GRAD, aux=d, ldaux=N)
    pre = (ref * 0.125 + bias).double()
    wh = torch.nn.functional.gelu(pre).float()
    wd = (0.5 * (1 + torch.erf(pre / 2 ** 0.5)) + pre * torch.exp(-0.5 * pre * pre) / (2 * torch.pi) ** 0.5).float()
    print("  gelu_grad h err", float((h.float() - wh).abs().max()), "d err", float((d.float() - wd).abs().max()))
for shp in [(2048, 256, 512), (4096, 3072, 768), (16384, 3072, 768), (9000, 2304, 768), (16384, 2304, 768)]:
    check(*shp)
torch.cuda.synchronize()
# timing
def bench(M, N, K, variant):
    R = 3
    sets = []
    for _ in range(R):
        a = torch.randn((M, K), device=DEV, generator=g).to(torch.bfloat16); b = torch.randn((N, K), device=DEV, generator=g).to(torch.bfloat16)
        kw = {}
        if variant == "gelu_grad": kw = dict(epilogue=DM_EPI_GELU_GRAD, aux=torch.empty((M, N), device=DEV, dtype=torch.bfloat16), bias=torch.randn((N,), device=DEV))
        elif variant == "bias": kw = dict(bias=torch.randn((N,), device=DEV))
        sets.append((a, b, torch.empty((M, N), device=DEV, dtype=torch.bfloat16), kw))
    def run(i):
        a, b, o, kw = sets[i % R]; ops.gemm(DM_NT, a, b, o, M, N, K, lda=K, ldb=K, ldc=N, **kw)
    for mode in ("2", "0", "2", "0"):
        os.environ["DM_GEMM_P2"] = mode
        for i in range(6): run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30): run(i)
        e1.record(); torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) / 30 * 1e-3
        print(f"{M}x{N}x{K} {variant:9s} P2={mode}: {dt*1e6:7.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TF/s", flush=True)
bench(16384, 3072, 768, "gelu_grad"); bench(16384, 3072, 768, "none"); bench(16384, 2304, 768, "bias"); bench(4096, 3072, 768, "gelu_grad")
