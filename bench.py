#!/usr/bin/env python3
"""Headline benchmark: superpixel-pairs/s of one Siamese training step on MI355X.

A "step" = forward of both sides (one 2B batch) + contrastive Loss + backward + gradient all-reduce
(N > 1, RCCL over xGMI) + fused Adam, on synthetic 256x256x4 multi-scale patches resident in HBM.
Workload at every N (weak scaling): BASELINE.json configs[1] -- ShfitScaleFormer_v3 "tiny"
(depth [3,2,1]) on 4 scales [32,64,128,256] x 4 channels, 32 pairs per GPU, bf16 operands.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs B] [--depth 3,2,1] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for the roofline / cpu_baseline fields).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_TBPS = 8.0
PEAK_F32_TFLOPS = 157.3


def synth_batch(B, scales, in_c, device, seed):
    """Synthetic pair batch of the reference's tensor contract (MyUtils1.py:41-77): per side a list of
    [B, in_c, s, s] float32 patches in [0,1] on a uint8 grid, designed features [B,1,19], flag [B]."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    left = [torch.randint(0, 256, (B, in_c, s, s), generator=g, dtype=torch.uint8).float().div_(255.0) for s in scales]
    right = [torch.randint(0, 256, (B, in_c, s, s), generator=g, dtype=torch.uint8).float().div_(255.0) for s in scales]
    flag = (torch.arange(B) % 2 == 0).to(torch.int64)
    for i in range(len(scales)):        # positives: jittered copy of the left crop (so some d < margin)
        right[i][flag == 1] = (left[i][flag == 1] * 0.9 + 0.1 * right[i][flag == 1])
    ld = torch.exp(torch.empty(B, 1, 19).uniform_(-4.6, 6.9, generator=g))
    rd = torch.where(flag.view(B, 1, 1) == 1, ld * 1.05, torch.exp(torch.empty(B, 1, 19).uniform_(-4.6, 6.9, generator=g)))
    mv = lambda t: t.to(device)
    return [mv(t) for t in left], mv(ld), [mv(t) for t in right], mv(rd), mv(flag)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(cfg_scales, in_c, depth, pairs, steps):
    """The oracle (CPU restatement, plain PyTorch fp32) timed on this box's host cores: forward both
    sides + loss + backward + Adam on a bounded sample of the same workload.  Reported, not a target."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from oracle import adam as OA
    from oracle import losses as OL
    from oracle import s2former as O
    from util import model_params
    cores = usable_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads, {pairs} pairs x {steps + 1} steps")
    cfg = O.S2Config(scales=tuple(cfg_scales), in_c=in_c, depth=tuple(depth))
    p = model_params(cfg)
    fparams = {k: v for k, v in p.items() if v.dtype.is_floating_point}
    m = {k: torch.zeros_like(v) for k, v in fparams.items()}
    v2 = {k: torch.zeros_like(v) for k, v in fparams.items()}
    left, ld, right, rd, flag = synth_batch(pairs, cfg_scales, in_c, "cpu", 1234)
    times = []
    for step in range(1, steps + 2):
        t0 = time.perf_counter()
        for t in fparams.values():
            t.grad = None
        fa, fb = O.forward_pair(p, left, ld, right, rd, cfg)
        loss = OL.contrastive_loss(fa, fb, flag, 1.0)
        loss.backward()
        with torch.no_grad():
            for k, t in fparams.items():
                if t.grad is not None:
                    OA.adam_step(t, t.grad, m[k], v2[k], step)
        times.append(time.perf_counter() - t0)
        log(f"cpu_baseline: step {step} took {times[-1]:.2f} s")
    dt = sum(times[1:]) / max(1, len(times) - 1)      # first step is warm-up
    return {"value": pairs / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{steps} timed steps (+1 warm-up) of {pairs} pairs, same model/inputs shape, torch CPU fp32 oracle"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=32, help="pairs per GPU per step")
    ap.add_argument("--depth", type=str, default="3,2,1")
    ap.add_argument("--numerics", type=str, default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--backend", type=str, default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for --gpus > 1 (gloo only to rehearse the DP path with several ranks on ONE GPU)")
    ap.add_argument("--graph", type=str, default="auto", choices=["auto", "on", "off"],
                    help="replay the step from a captured hipGraph (auto: on for one GPU; the data-parallel exchange runs eagerly)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    if args.backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())     # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import deepmerge_amd
    from deepmerge_amd import _lib
    from deepmerge_amd.nets.ShfitScaleFormer import ShfitScaleFormer_v3
    from deepmerge_amd.trainer import PairTrainer
    from deepmerge_amd.workload import pair_step_flops

    scales, in_c = [32, 64, 128, 256], 4
    depth = [int(d) for d in args.depth.split(",")]
    torch.manual_seed(0)
    net = ShfitScaleFormer_v3(cube_size=[8, 8], input_image_scales=list(scales), depth=list(depth), in_c=in_c,
                              numerics=args.numerics).to(dev)
    trainer = PairTrainer(net, margin=1.0, lr=1e-4)
    batch = synth_batch(args.pairs, scales, in_c, dev, 1000 + rank)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = (args.graph == "on") or (args.graph == "auto" and world == 1)
    n_warm = args.warmup
    if use_graph:
        n_warm = max(args.warmup, 2)                     # >= 1 eager step (lazy caches) + the capturing step, all untimed
        trainer.enable_graph(warmup=n_warm - 1)
    log(f"model built on {dev}; {n_warm} warm-up steps" + (" (the last one captures the hipGraph)" if use_graph else ""))
    for i in range(n_warm):
        trainer.step(*batch)
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")
    lib = _lib.lib()
    sync()
    if not use_graph:
        lib.dm_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(*batch)
    sync()
    dt = time.perf_counter() - t0
    lib.dm_prof_enable(0)
    log(f"timed region: {args.steps} steps in {dt:.3f} s")
    prof_steps, prof_note = args.steps, "hipEvents around every launch of the timed region"
    if use_graph:
        # graph replays carry no per-kernel events: time the same kernels on the same stream in 3 eager steps right after
        loss = loss.clone()
        prof_steps, prof_note = 3, "hipEvents around every launch of 3 eager steps run right after the timed region (graph replays carry no events)"
        lib.dm_prof_enable(1)
        for _ in range(prof_steps):
            trainer._eager_step(*batch)
        torch.cuda.synchronize()
        lib.dm_prof_enable(0)
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    rows = (_lib.DmProfRow * 256)()
    n = lib.dm_prof_collect(rows, 256)
    prof = {rows[i].name.decode(): (rows[i].launches, rows[i].total_ms, rows[i].total_flops, rows[i].total_bytes) for i in range(n)}

    if rank == 0:
        flop_pair = pair_step_flops(scales, in_c, depth)
        pairs_total = world * args.pairs * args.steps
        value = pairs_total / dt
        roof = None
        if prof:
            dom = max(prof, key=lambda k: prof[k][1])
            launches, ms, flops, _ = prof[dom]
            peak = PEAK_BF16_TFLOPS if "bf16" in dom else PEAK_F32_TFLOPS
            ach = flops / (ms * 1e-3) / 1e12
            traffic = None      # HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json), if any
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                if dom in pmc:
                    traffic = round(pmc[dom]["fetch_bytes"] + pmc[dom]["write_bytes"])
            except Exception:
                pass
            roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "algorithmic_bytes_per_launch": round(prof[dom][3] / launches),
                    "launches": launches,
                    # roofline position of this launch mix: FLOP per algorithmic HBM byte against the 312 FLOP/B ridge
                    # (2.5 PFLOP/s / 8 TB/s); attainable = min(peak, intensity * 8 TB/s)
                    "intensity_flop_per_byte": round(flops / max(prof[dom][3], 1), 1),
                    "attainable_TFLOPs": round(min(peak, flops / max(prof[dom][3], 1) * PEAK_HBM_TBPS), 1),
                    "avg_launch_us": round(1e3 * ms / launches, 2),
                    "all_kernels_TFLOPs": {k: round(v[2] / (v[1] * 1e-3) / 1e12, 1) for k, v in prof.items() if v[1] > 0},
                    "all_kernels_ms_per_step": {k: round(v[1] / prof_steps, 3) for k, v in prof.items()},
                    "measured_over": prof_note}
        out = {
            "metric": "superpixel-pairs/sec (train step)", "value": round(value, 2), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.numerics == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"ShfitScaleFormer_v3 depth {depth}, scales {scales} x {in_c}ch (256x256x4 patches), "
                                   f"{args.pairs} pairs/GPU/step, fwd+loss+bwd+allreduce+Adam (BASELINE configs[1])",
                       "pairs_per_gpu": args.pairs, "global_batch": world * args.pairs, "parallelism": f"dp{world}",
                       "gflop_per_pair_step": round(flop_pair / 1e9, 2)},
            "model_tflops_per_gpu": round(value / world * flop_pair / 1e12, 2),
            "loss": float(loss.item()), "backend": args.backend if world > 1 else None, "hip_graph": bool(use_graph),
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(scales, in_c, depth, args.cpu_pairs, args.cpu_steps)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
