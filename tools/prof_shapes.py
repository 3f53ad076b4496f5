"""Per-shape GEMM / attention table (hipEvents around every launch of a few eager steps, DM_PROF_SHAPES=1: one row per product shape).
    python tools/prof_shapes.py [bench.py args]          the headline model through bench.py (e.g. --depth 6,4,2 --pairs 120 = config 5)
    python tools/prof_shapes.py --config 3               BASELINE configs[2]: ViT-B/16 pair encoder, 128 pairs / step
"""
import json, os, subprocess, sys

os.environ["DM_PROF_SHAPES"] = "1"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def table(ms, tf, steps=1):
    tot = sum(ms.values())
    print(f"{'kernel':64s} {'ms/step':>8s} {'TFLOP/s':>8s}")
    for k in sorted(ms, key=lambda k: -ms[k]):
        print(f"{k:64s} {ms[k] / steps:8.3f} {tf.get(k, 0):8.1f}")
    print(f"{'total':64s} {tot / steps:8.3f}")


if "--config" in sys.argv and sys.argv[sys.argv.index("--config") + 1] == "3":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    from deepmerge_amd import _lib
    from deepmerge_amd.Losses import Loss
    from deepmerge_amd.trainer import PairTrainer
    from deepmerge_amd.vit_model import vit_base_patch16_224_in21k
    B, dev = 128, "cuda:0"
    net = vit_base_patch16_224_in21k(num_classes=100, has_logits=False, numerics="bf16").to(dev)

    class Pair(torch.nn.Module):
        def __init__(self, n):
            super().__init__(); self.n = n; self.numerics = "bf16"
        def forward(self, a, _1, b, _2):
            return self.n(a, b)
    tr = PairTrainer(Pair(net), margin=1.0, lr=1e-4)
    g = torch.Generator().manual_seed(0)
    x1 = torch.rand(B, 3, 224, 224, generator=g).to(dev); x2 = torch.rand(B, 3, 224, 224, generator=g).to(dev)
    flag = (torch.arange(B) % 2).to(dev)
    for _ in range(2):
        tr.step(x1, None, x2, None, flag)
    torch.cuda.synchronize()
    lib = _lib.lib()
    lib.dm_prof_enable(1)
    steps = 3
    for _ in range(steps):
        tr.step(x1, None, x2, None, flag)
    torch.cuda.synchronize()
    lib.dm_prof_enable(0)
    rows = (_lib.DmProfRow * 512)()
    n = lib.dm_prof_collect(rows, 512)
    ms = {rows[i].name.decode(): rows[i].total_ms for i in range(n)}
    tf = {rows[i].name.decode(): rows[i].total_flops / (rows[i].total_ms * 1e-3) / 1e12 for i in range(n) if rows[i].total_ms > 0}
    print("config 3: ViT-B/16 224x224x3, 128 pairs / step, bf16 (eager steps, hipEvents per launch)")
    table(ms, tf, steps)
else:
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras"] + sys.argv[1:],
                         env=dict(os.environ), capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)["roofline"]
    table(r["all_kernels_ms_per_step"], r["all_kernels_TFLOPs"])
