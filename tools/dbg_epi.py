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
FAMILIES = {      # environment that routes the product to one kernel family (read per call by dm_gemm)
    "128x128": {"DM_GEMM_FORCE_TILE": "128", "DM_GEMM_256": "0", "DM_GEMM_W4": "0", "DM_GEMM_RING": "0"},
    "256x256": {"DM_GEMM_256": "2", "DM_GEMM_W4": "0", "DM_GEMM_RING": "0"},
    "ring": {"DM_GEMM_RING": "2", "DM_GEMM_256": "0", "DM_GEMM_W4": "0"},
    "w4": {"DM_GEMM_W4": "2", "DM_GEMM_256": "0", "DM_GEMM_RING": "0"},
}
KEYS = ("DM_GEMM_FORCE_TILE", "DM_GEMM_256", "DM_GEMM_W4", "DM_GEMM_RING")
want = (ref * 0.125 + bias).to(torch.bfloat16)
wh = torch.nn.functional.gelu(ref * 0.125 + bias)
for fam in (sys.argv[1:] or list(FAMILIES)):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(FAMILIES[fam])
    for epi, name in ((DM_EPI_GELU, "gelu+pre"), (DM_EPI_GELU_GRAD, "gelu+gelu'")):
        pre = torch.zeros((M, N), device=DEV, dtype=torch.bfloat16)
        h = torch.zeros((M, N), device=DEV, dtype=torch.bfloat16)
        ops.gemm(DM_NT, a2, b, h, M, N, K, lda=K, ldb=K, ldc=N, bias=bias, epilogue=epi, aux=pre, ldaux=N)
        if epi == DM_EPI_GELU:
            bad = (pre != want)
        else:                                  # the saved derivative: compare two runs with each other (a corrupted store differs run to run)
            pre2 = torch.zeros_like(pre); h2 = torch.zeros_like(h)
            ops.gemm(DM_NT, a2, b, h2, M, N, K, lda=K, ldb=K, ldc=N, bias=bias, epilogue=epi, aux=pre2, ldaux=N)
            bad = (pre != pre2) | (h != h2)
        herr = (h.float() - wh).abs().max().item()
        print("family", fam, name, "bad", int(bad.sum()), "of", bad.numel(), "h err", round(herr, 4))
        if bad.any():
            idx = bad.nonzero()
            print("  first", idx[:6].tolist(), "rows", len(idx[:, 0].unique()), "cols", len(idx[:, 1].unique()))
