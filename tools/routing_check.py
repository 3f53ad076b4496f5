"""Routing self-check: for token counts M the routing was NOT tuned on, is the kernel family that
`dm_gemm` picks within 5 % of the best forced family?  One block's twelve bf16 products with the epilogues the training step uses
(bias, fp32 residual, GELU' saved, x saved GELU', fused column sums), cold-ish operands (R rotating sets), hipEvent time per launch.

    python tools/routing_check.py [M ...]        default: 3840 7680 15360 61440 384000 (a per-GPU batch of 60 .. the 2000-point eval batch)

ONLY=qkv.dgrad,fc1.dgrad restricts the products.  Prints one line per (M, product): routed time, every legal forced family's time, best / routed ratio; `<<<` marks products where the
routed kernel is more than 5 % slower than the best one."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from deepmerge_amd import ops
from deepmerge_amd._lib import DM_EPI_GELU_GRAD, DM_EPI_MUL, DM_EPI_NONE, DM_NN, DM_NT, DM_TN

DEV = "cuda:0"
g = torch.Generator(device=DEV); g.manual_seed(1)
KEYS = ("DM_GEMM_FORCE_TILE", "DM_GEMM_256", "DM_GEMM_W4", "DM_GEMM_W4_TN", "DM_GEMM_RING", "DM_GEMM_FWD_SPLIT")
OFF = {"DM_GEMM_256": "0", "DM_GEMM_W4": "0", "DM_GEMM_W4_TN": "0", "DM_GEMM_RING": "0"}
FAMILIES = {          # name -> (environment, layouts it applies to)
    "128x128": ({**OFF, "DM_GEMM_FORCE_TILE": "128"}, (DM_NT, DM_NN, DM_TN)),
    "64x64": ({**OFF, "DM_GEMM_FORCE_TILE": "64"}, (DM_NT, DM_NN, DM_TN)),
    "256x256": ({**OFF, "DM_GEMM_256": "2"}, (DM_NT, DM_NN, DM_TN)),
    "ring": ({**OFF, "DM_GEMM_RING": "2"}, (DM_NT,)),
    "w4": ({**OFF, "DM_GEMM_W4": "2", "DM_GEMM_W4_TN": "2"}, (DM_NT, DM_NN, DM_TN)),
}


def rnd(shape, dt=torch.bfloat16):
    return (torch.randn(shape, device=DEV, generator=g) * 0.5).to(dt)


def set_env(env):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)


def timeit(run, it):
    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(it):
        run(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3          # us


def products(M, C=768, Hd=3072):
    """(name, layout, M, N, K, kind): kind selects the epilogue the step uses for that product."""
    out = []
    for nm, n_out, k_in, fwd_kind, dg_kind in (("qkv", 3 * C, C, "bias", "plain"), ("proj", C, C, "res", "plain"),
                                               ("fc1", Hd, C, "gelu", "plain"), ("fc2", C, Hd, "res", "mul")):
        out.append((f"{nm}.fwd", DM_NT, M, n_out, k_in, fwd_kind))
        out.append((f"{nm}.dgrad", DM_NN, M, k_in, n_out, dg_kind))
        out.append((f"{nm}.wgrad", DM_TN, n_out, k_in, M, "wgrad"))
    return out


def case(name, layout, M, N, K, kind, R, it):
    sa, sb = ((M, K), (N, K)) if layout == DM_NT else ((M, K), (K, N)) if layout == DM_NN else ((K, M), (K, N))
    sets = []
    for _ in range(R):
        d = {"a": rnd(sa), "b": rnd(sb)}
        if kind == "res":
            d["c"], d["res"], d["bias"] = torch.empty((M, N), device=DEV), rnd((M, N), torch.float32), rnd((N,), torch.float32)
        elif kind == "gelu":
            d["c"], d["aux"], d["bias"] = torch.empty((M, N), device=DEV, dtype=torch.bfloat16), torch.empty((M, N), device=DEV, dtype=torch.bfloat16), rnd((N,), torch.float32)
        elif kind == "mul":
            d["c"], d["aux"] = torch.empty((M, N), device=DEV, dtype=torch.bfloat16), rnd((M, N))
        elif kind == "wgrad":
            d["c"], d["cs"] = torch.empty((M, N), device=DEV), torch.empty((M,), device=DEV)
        else:
            d["c"] = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
            if kind == "bias":
                d["bias"] = rnd((N,), torch.float32)
        sets.append(d)

    def run(i):
        d = sets[i % R]
        kw = dict(lda=sa[1], ldb=sb[1], ldc=N)
        if kind == "res":
            ops.gemm(layout, d["a"], d["b"], d["c"], M, N, K, bias=d["bias"], residual=d["res"], ldr=N, **kw)
        elif kind == "gelu":
            ops.gemm(layout, d["a"], d["b"], d["c"], M, N, K, bias=d["bias"], epilogue=DM_EPI_GELU_GRAD, aux=d["aux"], ldaux=N, **kw)
        elif kind == "mul":
            ops.gemm(layout, d["a"], d["b"], d["c"], M, N, K, epilogue=DM_EPI_MUL, aux=d["aux"], ldaux=N, **kw)
        elif kind == "wgrad":
            ops.gemm(layout, d["a"], d["b"], d["c"], M, N, K, colsum_out=d["cs"], **kw)
        else:
            ops.gemm(layout, d["a"], d["b"], d["c"], M, N, K, bias=d.get("bias"), **kw)
    set_env({})
    routed = timeit(run, it)
    times = {}
    for fam, (env, lays) in FAMILIES.items():
        if layout not in lays:
            continue
        set_env(env)
        try:
            times[fam] = timeit(run, it)
        except ValueError:            # a family that does not take the shape (DM_ERR_UNSUPPORTED)
            pass
    set_env({})
    routed2 = timeit(run, it)         # (second look at the routed kernel: the box drifts by ~1 %)
    routed = min(routed, routed2)
    best = min(times, key=times.get)
    ratio = routed / times[best]
    fl = 2.0 * M * N * K
    flag = "   <<<" if ratio > 1.05 else ""
    print(f"{name:11s} {M:6d}x{N:5d}x{K:6d}  routed {routed:8.1f} us {fl / routed / 1e6:6.0f} TF/s | " +
          "  ".join(f"{f} {t:7.1f}" for f, t in times.items()) + f" | best {best} ratio {ratio:4.2f}{flag}", flush=True)
    del sets
    torch.cuda.empty_cache()
    return ratio


if __name__ == "__main__":
    Ms = [int(a) for a in sys.argv[1:]] or [3840, 7680, 15360, 61440, 384000]
    worst = {}
    for M in Ms:
        big = M > 100000
        print(f"== M = {M} tokens ==", flush=True)
        only = [x for x in os.environ.get("ONLY", "").split(",") if x]
        for p in products(M):
            if only and p[0] not in only:
                continue
            r = case(*p, R=1 if big else 3, it=5 if big else 20)
            worst[(M, p[0])] = r
    off = {k: v for k, v in worst.items() if v > 1.05}
    print(f"{len(worst)} products checked, {len(off)} routed more than 5 % behind the best forced family: " +
          ", ".join(f"{k[1]}@{k[0]} {v:.2f}" for k, v in sorted(off.items(), key=lambda kv: -kv[1])))
