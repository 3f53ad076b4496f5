"""bf16 (throughput) vs fp32 (parity) training trajectories on the same synthetic batches: per-step loss."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from bench import synth_batch
from deepmerge_amd.nets.ShfitScaleFormer import ShfitScaleFormer_v3
from deepmerge_amd.trainer import PairTrainer
dev = "cuda:0"
scales, in_c = [32, 64, 128, 256], 4
batches = [synth_batch(16, scales, in_c, dev, 100 + i) for i in range(4)]
out = {}
for mode in ("fp32", "bf16", "bf16-graph"):
    torch.manual_seed(0)
    net = ShfitScaleFormer_v3(cube_size=[8, 8], input_image_scales=list(scales), depth=[3, 2, 1], in_c=in_c, numerics=mode.split("-")[0]).to(dev)
    tr = PairTrainer(net, lr=1e-4)
    if mode.endswith("graph"):
        tr.enable_graph(warmup=2)
    losses = []
    for step in range(16):
        losses.append(float(tr.step(*batches[step % 4])))
    out[mode] = losses
    print(mode, " ".join(f"{l:.4f}" for l in losses), flush=True)
rel = max(abs(a - b) / max(abs(a), 1e-6) for a, b in zip(out["fp32"], out["bf16"]))
print("max relative loss difference bf16 vs fp32 over 16 steps:", f"{rel:.3e}")
print("bf16 eager == bf16 graph:", out["bf16"] == out["bf16-graph"])
