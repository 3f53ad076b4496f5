"""Host-side cost of one eager step (enqueue only, no synchronisation inside the loop) against its GPU time."""
import os, sys, time
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
n = 30
t0 = time.perf_counter()
for _ in range(n): tr.step(*batch)
t_enq = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / n
print(f"enqueue {t_enq*1e3:.2f} ms/step   total {t_all*1e3:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): tr.step(*batch)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
