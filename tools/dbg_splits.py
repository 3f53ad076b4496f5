"""Which operands still go through a split pass in one bf16x3 training step (shape, with / without column sums, caller line)."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import bench
from deepmerge_amd import ops
from deepmerge_amd.nets.ShfitScaleFormer import ShfitScaleFormer_v3
from deepmerge_amd.trainer import PairTrainer
dev = "cuda:0"
scales, in_c = [32, 64, 128, 256], 4
torch.manual_seed(0)
net = ShfitScaleFormer_v3(cube_size=[8, 8], input_image_scales=list(scales), depth=[3, 2, 1], in_c=in_c, numerics="bf16x3").to(dev)
tr = PairTrainer(net, margin=1.0, lr=1e-4)
batch = bench.synth_batch(32, scales, in_c, dev, 1000)
tr.step(*batch); torch.cuda.synchronize()
seen = collections.Counter()
orig = ops.split_planes
def spy(x, colsum_out=None, *a, **k):
    fr = [f for f in traceback.extract_stack()[:-1] if f.filename.endswith("ops.py")][-1]
    seen[(tuple(x.shape), colsum_out is not None, fr.lineno, fr.line.strip()[:70])] += 1
    return orig(x, colsum_out, *a, **k)
ops.split_planes = spy
tr.step(*batch); torch.cuda.synchronize()
for k, v in sorted(seen.items(), key=lambda kv: -kv[0][0][0] * kv[0][0][1] * kv[1]):
    print(v, k)
