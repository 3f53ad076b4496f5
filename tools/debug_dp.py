import os, sys, socket, torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from test_gpu_dp import _build, _batch

def worker(rank, world, port, out, nb, sinks):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    net = _build("fp32")
    from deepmerge_amd.trainer import PairTrainer, shard_slice
    tr = PairTrainer(net, lr=1e-4, n_buckets=nb)
    if not sinks:
        for p in tr.fp.params:
            if hasattr(p, "_dm_grad_sink"): del p._dm_grad_sink
    left, ld, right, rd, flag = _batch(8)
    sl = shard_slice(8, rank, world)
    mv = lambda t: t[sl].to("cuda:0")
    tr.step([mv(t) for t in left], mv(ld), [mv(t) for t in right], mv(rd), mv(flag))
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({n: (p.grad / world).cpu() for n, p in net.named_parameters()}, out)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    from deepmerge_amd.trainer import PairTrainer
    net = _build("fp32"); tr = PairTrainer(net, lr=1e-4)
    left, ld, right, rd, flag = _batch(8); mv = lambda t: t.to("cuda:0")
    tr.step([mv(t) for t in left], mv(ld), [mv(t) for t in right], mv(rd), mv(flag))
    ref = {n: p.grad.cpu().clone() for n, p in net.named_parameters()}
    for nb, sinks in ((1, True), (3, True)):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        out = f"/tmp/dp_{nb}_{sinks}.pt"
        mp.spawn(worker, args=(2, port, out, nb, sinks), nprocs=2, join=True)
        got = torch.load(out)
        bad = [(float((got[n] - ref[n]).abs().max() / (ref[n].abs().max() + 1e-12)), n) for n in ref]
        bad.sort(reverse=True)
        print(f"n_buckets={nb} sinks={sinks}: worst", [(round(a, 4), n) for a, n in bad[:6]], flush=True)
