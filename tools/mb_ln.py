import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deepmerge_amd import ops
dev = "cuda:0"
rows, C = 16384, 768
x = torch.randn((rows, C), device=dev); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5, torch.bfloat16)
dy = torch.randn((rows, C), device=dev).to(torch.bfloat16); dres = torch.randn((rows, C), device=dev)
dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
tf = t(lambda: ops.layernorm_fwd(x, g, b, 1e-5, torch.bfloat16))
tb = t(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, dgamma=dg, dbeta=db, accumulate=True, want_lp=True))
print(f"LN fwd {tf*1e6:.1f} us ({rows*C*6/tf/1e12:.2f} TB/s)   LN bwd+reduce {tb*1e6:.1f} us ({rows*C*16/tb/1e12:.2f} TB/s)")
