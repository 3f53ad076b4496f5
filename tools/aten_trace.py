"""Which torch (aten) kernels run inside one training step?  (glue around the HIP library)"""
import os, sys
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    tr.step(*batch)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda r: -r[2])
for k, c, t in rows[:25]:
    print(f"{k:40s} n={c:4d}  {t:9.1f} us")
