#!/usr/bin/env python3
"""Train_SMT counterpart on synthetic data (Train_SMT.py:212-351): epochs of steps, MultiStepLR([40, 80], 0.2) stepped per
epoch, FRESH pairs every step generated on the device -- patch pyramids cut from resident uint8 tiles by the sync-free feed
(deepmerge_amd/feed.py: one dm_pair_batch_gather launch per scale straight into the captured step's inputs), positives = two
jittered windows of the same spot, negatives = spots of different tiles -- through PairTrainer (hipGraph replay).  Prints a loss curve; BASELINE configs[4] shape by default (depth [6,4,2], 4 scales x 4 ch,
120 pairs per GPU).        python tools/train_synth.py [--epochs 3] [--steps-per-epoch 20] [--pairs 120] [--depth 6,4,2]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepmerge_amd.nets.ShfitScaleFormer import ShfitScaleFormer_v3  # noqa: E402
from deepmerge_amd.feed import PairFeed, PairTable  # noqa: E402
from deepmerge_amd.patches import point_batch  # noqa: E402
from deepmerge_amd.trainer import PairTrainer, multistep_lr  # noqa: E402

DEV = "cuda:0"


def smooth_tiles(n, bands, size, g):
    """Spatially correlated uint8 tiles (so that two nearby windows look alike and two tiles do not)."""
    low = torch.rand((n, bands, size // 32, size // 32), generator=g, device=DEV)
    t = torch.nn.functional.interpolate(low, size=(size, size), mode="bilinear", align_corners=False)
    t = t + 0.08 * torch.rand((n, bands, size, size), generator=g, device=DEV)
    return (t.clamp(0, 1) * 255).to(torch.uint8)


def draw_pairs(n_tiles, size, B, g):
    """The sample table of one batch, drawn on the device (same draws, in the same order, as the per-tile loop of rounds 1-4 --
    `make_pairs_looped` below -- so the two feeds see identical data)."""
    flag = (torch.arange(B, device=DEV) % 2 == 0).to(torch.int64)
    tl = torch.randint(0, n_tiles, (B,), generator=g, device=DEV)
    tr = torch.where(flag == 1, tl, (tl + 1 + torch.randint(0, n_tiles - 1, (B,), generator=g, device=DEV)) % n_tiles)
    xy_l = torch.randint(96, size - 96, (B, 2), generator=g, device=DEV)
    xy_r = torch.where(flag[:, None] == 1, xy_l + torch.randint(-6, 7, (B, 2), generator=g, device=DEV),
                       torch.randint(96, size - 96, (B, 2), generator=g, device=DEV))
    inner = torch.randint(16, 65, (B,), generator=g, device=DEV)
    obj = inner + torch.randint(8, 49, (B,), generator=g, device=DEV)
    feats_l = torch.exp(torch.empty((B, 15), device=DEV).uniform_(-2.0, 3.0, generator=g))
    feats_r = torch.where(flag[:, None] == 1, feats_l * 1.05, torch.exp(torch.empty((B, 15), device=DEV).uniform_(-2.0, 3.0, generator=g)))
    return tl, tr, xy_l, xy_r, inner, obj, feats_l, feats_r, flag


def pair_table(draw):
    """deepmerge_amd.feed.PairTable of a draw: [left rows; right rows], int32 columns -- device ops only, nothing read back."""
    tl, tr, xy_l, xy_r, inner, obj, feats_l, feats_r, flag = draw
    i32 = lambda a, b: torch.cat((a, b), 0).to(torch.int32)
    return PairTable(tile_id=i32(tl, tr), xy=i32(xy_l, xy_r), inner=i32(inner, inner), obj=i32(obj, obj),
                     region=torch.cat((feats_l, feats_r), 0), flag=flag)


MAX_WINDOW = [64, 112, 160, 208]       # bounds of (inner, obj, obj + interval, obj + 2 interval) for the ranges drawn above


def make_pairs_looped(tiles, B, scales, g):
    """The feed of rounds 1-4 (`--feed looped`, kept for the A/B and the bit-for-bit test): one gather launch per (tile, scale), tile
    membership by `nonzero`, window sides through the host."""
    n, bands, size, _ = tiles.shape
    tl, tr, xy_l, xy_r, inner, obj, feats_l, feats_r, flag = draw_pairs(n, size, B, g)

    def side(tid, xy, feats):
        patches = [torch.empty((B, bands, s, s), device=DEV) for s in scales]
        designed = torch.empty((B, 1, 19), device=DEV)
        for t in range(n):                                  # one gather launch per (tile, scale)
            sel = torch.nonzero(tid == t).reshape(-1)
            if sel.numel() == 0:
                continue
            p, d = point_batch(tiles[t], xy[sel].to(torch.int32), inner[sel].cpu(), obj[sel].cpu(), feats[sel], scales=scales)
            for i in range(len(scales)):
                patches[i][sel] = p[i]
            designed[sel] = d
        return patches, designed
    pl, dl = side(tl, xy_l, feats_l)
    pr, dr = side(tr, xy_r, feats_r)
    return pl, dl, pr, dr, flag


def main():  # noqa: C901
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--steps-per-epoch", type=int, default=20)
    ap.add_argument("--pairs", type=int, default=120)
    ap.add_argument("--depth", type=str, default="6,4,2")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--numerics", type=str, default="bf16", choices=["bf16", "fp32", "bf16x3"])
    ap.add_argument("--milestones", type=str, default="40,80")
    ap.add_argument("--feed", type=str, default="rows", choices=["rows", "patches", "looped"],
                    help="rows: deepmerge_amd.feed.PairFeed writing the patch-embed operand rows (default); patches: the same feed writing "
                         "fp32 patch tensors; looped: the per-tile Python loop of rounds 1-4")
    ap.add_argument("--loss-every", type=int, default=0, help="read the loss back every k steps (0 = once per epoch)")
    args = ap.parse_args()
    scales, bands = [32, 64, 128, 256], 4
    depth = [int(d) for d in args.depth.split(",")]
    ms = tuple(int(m) for m in args.milestones.split(","))
    g = torch.Generator(device=DEV); g.manual_seed(0)
    torch.manual_seed(0)
    tiles = smooth_tiles(6, bands, 1024, g)
    net = ShfitScaleFormer_v3(cube_size=[8, 8], input_image_scales=list(scales), depth=depth, in_c=bands, numerics=args.numerics).to(DEV)
    tr = PairTrainer(net, margin=1.0, lr=args.lr)
    tr.enable_graph(warmup=1)
    feed = None if args.feed == "looped" else PairFeed(tiles, scales, args.pairs, MAX_WINDOW, rows=args.feed == "rows", numerics=args.numerics, trainer=tr)
    curve = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    WARM = 3                                                 # eager step(s) + the capturing step: not steady state (one extra sync after them)
    t_warm, done = None, 0
    for epoch in range(args.epochs):
        lr = multistep_lr(args.lr, epoch, ms)
        tot = torch.zeros((), device=DEV)
        for k in range(args.steps_per_epoch):
            if done == WARM:
                torch.cuda.synchronize(); t_warm = time.perf_counter()
            done += 1
            if feed is None:
                batch = make_pairs_looped(tiles, args.pairs, scales, g)
            else:
                batch = feed.fill(pair_table(draw_pairs(tiles.shape[0], tiles.shape[2], args.pairs, g)))
            tot += tr.step(*batch, lr=lr)                    # (device-side sum: the reference's per-step .item(), Train_SMT.py:301, is a sync)
            if args.loss_every and (k + 1) % args.loss_every == 0:
                float(tot)
        curve.append(round(float(tot) / args.steps_per_epoch, 5))
        if feed is not None:
            feed.check()
        print(f"epoch {epoch}: lr {lr:.2e} mean loss {curve[-1]:.5f}", flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = args.epochs * args.steps_per_epoch
    print(json.dumps({"tool": "train_synth", "feed": args.feed, "numerics": args.numerics, "depth": depth, "pairs_per_step": args.pairs, "steps": n,
                      "loss_curve": curve, "hip_graph": tr.graph_error is None, "pairs_per_s_incl_data_generation": round(n * args.pairs / dt, 1),
                      "pairs_per_s_steady_incl_data_generation": None if t_warm is None else round((n - WARM) * args.pairs / (t0 + dt - t_warm), 1),
                      "steady_note": f"after the first {WARM} steps (eager warm-up + hipGraph capture)"}))


if __name__ == "__main__":
    main()
