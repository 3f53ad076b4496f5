"""Torch (aten) kernels inside one training step with their input shapes (to find the glue left around the HIP library)."""
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
tr.enable_graph(warmup=2)
for _ in range(4): tr.step(*batch)
batch = tr.graph_inputs()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr._eager_step(*batch)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith("aten::") or e.device_time_total <= 0:
        continue
    if e.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in e.cpu_children):
        continue
    agg[(e.name, str(e.input_shapes)[:110])][0] += 1
    agg[(e.name, str(e.input_shapes)[:110])][1] += e.device_time_total
for (name, shp), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{name:24s} n={n:3d} {t:8.1f} us  {shp}")
