import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deepmerge_amd import ops
n = 48_700_000
dev = "cuda:0"
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
lp = torch.empty(n, device=dev, dtype=torch.bfloat16)
for _ in range(3): ops.adam_step(p, g, m, v, 3, param_lp=lp)
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(30): ops.adam_step(p, g, m, v, 3 + i, param_lp=lp)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 30
print(f"adam {dt*1e6:.1f} us  {n*30/dt/1e12:.2f} TB/s (30 B/param incl. bf16 mirror)")
