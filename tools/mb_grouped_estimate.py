"""Is a grouped weight-gradient launch worth building?  The four weight gradients of one block at T tokens (2304x768, 768x768, 3072x768,
768x3072 outputs, contraction T) as routed today, against ONE 4-wave-kernel launch with the same 144 tiles of 256 x 192 and no K slices
(a 2304 x 3072 x T product forced onto dm_gemm_w4 stands in for the grouped launch: same tiles, same K loop, no slab, no reduction).

    python tools/mb_grouped_estimate.py [T ...]          default 1024 4096 16384
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from deepmerge_amd import ops
from deepmerge_amd._lib import DM_TN

DEV = "cuda:0"
g = torch.Generator(device=DEV); g.manual_seed(2)


def rnd(shape):
    return (torch.randn(shape, device=DEV, generator=g) * 0.5).to(torch.bfloat16)


def timeit(run, it=30):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for T in [int(a) for a in sys.argv[1:]] or [1024, 4096, 16384]:
    C, H = 768, 3072
    shapes = [(3 * C, C), (C, C), (H, C), (C, H)]
    ops_ = []
    for (m, n) in shapes:
        a, b = rnd((T, m)), rnd((T, n))
        ops_.append((a, b, torch.empty((m, n), device=DEV), torch.empty((m,), device=DEV), m, n))

    def separate():
        for a, b, c, cs, m, n in ops_:
            ops.gemm(DM_TN, a, b, c, m, n, T, lda=m, ldb=n, ldc=n, colsum_out=cs)
    for k in ("DM_GEMM_W4_TN",):
        os.environ.pop(k, None)
    t_sep = timeit(separate)
    calls = [((DM_TN, a, b, c, m, n, T), dict(lda=m, ldb=n, ldc=n, colsum_out=cs)) for a, b, c, cs, m, n in ops_]
    t_grp = timeit(lambda: ops.gemm_grouped(calls))
    A, B, Cc, cs = rnd((T, 2304)), rnd((T, 3072)), torch.empty((2304, 3072), device=DEV), torch.empty((2304,), device=DEV)
    os.environ["DM_GEMM_W4_TN"] = "2"
    t_one = timeit(lambda: ops.gemm(DM_TN, A, B, Cc, 2304, 3072, T, lda=2304, ldb=3072, ldc=3072, colsum_out=cs))
    os.environ.pop("DM_GEMM_W4_TN")
    t_one_routed = timeit(lambda: ops.gemm(DM_TN, A, B, Cc, 2304, 3072, T, lda=2304, ldb=3072, ldc=3072, colsum_out=cs))
    print(f"T = {T:6d}: four weight gradients as routed {t_sep:7.1f} us | dm_gemm_grouped {t_grp:7.1f} us | one launch, 144 tiles, no K slices (4-wave kernel) {t_one:7.1f} us"
          f" | the same product as routed {t_one_routed:7.1f} us", flush=True)
