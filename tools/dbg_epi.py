import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deepmerge_amd import ops
from deepmerge_amd._lib import DM_NT, DM_EPI_GELU, DM_EPI_GELU_GRAD, DM_EPI_MUL
DEV = "cuda:0"
g = torch.Generator(device=DEV); g.manual_seed(3)
M, N, K = 4096, 3072, 768
a = torch.randint(-2, 3, (M, K), device=DEV, generator=g).to(torch.bfloat16)
b = torch.randint(-2, 3, (N, K), device=DEV, generator=g).to(torch.bfloat16)
bias = torch.randint(-3, 4, (N,), device=DEV, generator=g).float()
ref = a.float() @ b.float().T
a2 = (a.float() * 0.125).to(torch.bfloat16)
for forced in ("128", None):
    if forced: os.environ["DM_GEMM_FORCE_TILE"] = forced; os.environ["DM_GEMM_256"] = "0"; os.environ["DM_GEMM_W4"]="0"
    else:
        os.environ.pop("DM_GEMM_FORCE_TILE", None); os.environ["DM_GEMM_256"] = "2"
    pre = torch.zeros((M, N), device=DEV, dtype=torch.bfloat16)
    h = torch.zeros((M, N), device=DEV, dtype=torch.bfloat16)
    ops.gemm(DM_NT, a2, b, h, M, N, K, lda=K, ldb=K, ldc=N, bias=bias, epilogue=DM_EPI_GELU, aux=pre, ldaux=N)
    want = (ref * 0.125 + bias).to(torch.bfloat16)
    bad = (pre != want)
    print("forced", forced, "bad", int(bad.sum()), "of", bad.numel())
    if bad.any():
        idx = bad.nonzero()
        print(idx[:10].tolist(), idx[-3:].tolist())
        rows = idx[:, 0].unique(); cols = idx[:, 1].unique()
        print("rows", rows[:20].tolist(), len(rows), "cols", cols[:20].tolist(), len(cols))
        i, j = idx[0].tolist()
        print("got", pre[i, j].item(), "want", want[i, j].item(), "h", h[i, j].item())
    wh = torch.nn.functional.gelu(ref * 0.125 + bias)
    print("h err", (h.float() - wh).abs().max().item())
