"""Which launches of the training step come from torch itself (copies, fills, elementwise kernels) and which Python line issues them.

    PYTHONPATH=. python tools/torch_ops_in_step.py [--numerics bf16] [--pairs 32] [--graph]

One step under a TorchDispatchMode: every aten operator that reaches the dispatcher with a device tensor is listed with the innermost
frame inside deepmerge_amd/ (forward) or the autograd node that issued it (backward) and its count per step.  View / metadata operators
are dropped.  The library's own kernels (ctypes calls) do not appear: they are not aten operators.  --graph lists the capturing step.
"""
import argparse
import collections
import traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench
from deepmerge_amd.nets.ShfitScaleFormer import ShfitScaleFormer_v3
from deepmerge_amd.trainer import PairTrainer

VIEWS = ("view", "reshape", "slice", "select", "transpose", "permute", "expand", "as_strided", "unsqueeze", "squeeze", "detach", "alias",
         "t.default", "_unsafe_view", "empty", "split", "unbind", "narrow", "sym_", "size", "stride", "is_", "_local_scalar", "lift",
         "new_empty", "result_type", "numel", "dim", "storage_offset", "_reshape_alias", "unfold", "chunk", "flatten")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not any(name.startswith(v) for v in VIEWS):
            on_dev = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values()))
            if on_dev or name.startswith(("zeros", "ones", "full", "rand", "arange")):
                frames = [f for f in traceback.extract_stack() if "deepmerge_amd/" in f.filename or f.filename.endswith("bench.py")]
                where = f"{frames[-1].filename.split('deepmerge_amd/')[-1]}:{frames[-1].lineno} {frames[-1].name}" if frames else "(autograd engine)"
                shapes = ",".join(str(tuple(a.shape)) for a in args if isinstance(a, torch.Tensor))[:48]
                self.sites[(name, where, shapes)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--numerics", default="bf16")
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--graph", action="store_true")
    args = ap.parse_args()
    dev = "cuda:0"
    scales, in_c, depth = [32, 64, 128, 256], 4, [3, 2, 1]
    torch.manual_seed(0)
    net = ShfitScaleFormer_v3(cube_size=[8, 8], input_image_scales=list(scales), depth=list(depth), in_c=in_c, numerics=args.numerics).to(dev)
    trainer = PairTrainer(net, margin=1.0, lr=1e-4)
    batch = bench.synth_batch(args.pairs, scales, in_c, dev, 1000)
    if args.graph:
        trainer.enable_graph(warmup=1)
    trainer.step(*batch)
    torch.cuda.synchronize()
    with Log() as log:
        trainer.step(*batch)                          # (--graph: the capturing step)
    torch.cuda.synchronize()
    for (name, where, shapes), n in sorted(log.sites.items(), key=lambda kv: (kv[0][1], -kv[1])):
        print(f"{n:4d}  {name:32s} {where:56s} {shapes}")


if __name__ == "__main__":
    main()
