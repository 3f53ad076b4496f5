"""Which torch (aten) kernels run inside one training step, and from which source line?  (glue around the HIP library)"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from bench import synth_batch
from deepmerge_amd.nets.ShfitScaleFormer import ShfitScaleFormer_v3
from deepmerge_amd.trainer import PairTrainer
dev = "cuda:0"
scales, in_c = [32, 64, 128, 256], 4
torch.manual_seed(0)
net = ShfitScaleFormer_v3(cube_size=[8, 8], input_image_scales=list(scales), depth=[3, 2, 1], in_c=in_c, numerics="bf16").to(dev)
tr = PairTrainer(net, lr=1e-4)
batch = synth_batch(32, scales, in_c, dev, 1000)
for _ in range(3): tr.step(*batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(*batch)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith("aten::") or e.device_time_total <= 0 or e.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in e.cpu_children):
        continue
    where = next((f for f in (e.stack or []) if "deepmerge_amd" in f or "bench.py" in f), "?")
    where = where.split("/repo/")[-1]
    agg[(e.name, where)][0] += 1
    agg[(e.name, where)][1] += e.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for (name, where), (n, t) in rows[:40]:
    print(f"{name:28s} n={n:3d} {t:8.1f} us  {where}")
