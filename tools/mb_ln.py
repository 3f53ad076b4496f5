"""LayerNorm forward / backward timing at the encoder's stage-0 shape (rows x 768, bf16 upstream gradient, residual-gradient add,
bf16 copy of dx) against the bytes each must move.  DM_LN_WG sets the backward's workgroup count (= partial rows)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deepmerge_amd import ops
dev = "cuda:0"
rows, cols = int(os.environ.get("ROWS", 16384)), 768
g = torch.Generator(device=dev); g.manual_seed(0)
NSET = 6                                     # rotate operand sets so that nothing stays in the Infinity Cache
xs = [torch.randn((rows, cols), device=dev, generator=g) for _ in range(NSET)]
dys = [torch.randn((rows, cols), device=dev, generator=g).to(torch.bfloat16) for _ in range(NSET)]
drs = [torch.randn((rows, cols), device=dev, generator=g) for _ in range(NSET)]
gamma = torch.randn(cols, device=dev, generator=g); beta = torch.randn(cols, device=dev, generator=g)
y, mean, rstd = ops.layernorm_fwd(xs[0], gamma, beta, 1e-6, torch.bfloat16)
def timeit(f, n=30):
    for i in range(3): f(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n): f(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
t = timeit(lambda i: ops.layernorm_fwd(xs[i % NSET], gamma, beta, 1e-6, torch.bfloat16))
print(f"fwd  rows={rows}: {t*1e6:6.1f} us  {rows*cols*6/t/1e12:5.2f} TB/s (4 B in + 2 B out per element)")
t = timeit(lambda i: ops.layernorm_bwd(dys[i % NSET], xs[i % NSET], gamma, mean, rstd, dres=drs[i % NSET], want_lp=True))
print(f"bwd  rows={rows}: {t*1e6:6.1f} us  {rows*cols*16/t/1e12:5.2f} TB/s (2 + 4 + 4 B in, 4 + 2 B out per element; incl. the partial reduce launch and output allocation)")
