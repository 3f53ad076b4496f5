import os, sys, socket, torch, torch.distributed as dist, torch.multiprocessing as mp
def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    n = 24_000_000
    for trial in range(3):
        x = torch.zeros(n, device="cuda:0")
        # enqueue a long chain of kernels writing x, then all-reduce views of it asynchronously
        for _ in range(20):
            x += (rank + 1)
        works = [dist.all_reduce(x[i * (n // 3):(i + 1) * (n // 3)], async_op=True) for i in range(3)]
        for w in works: w.wait()
        y = x * 1.0
        torch.cuda.synchronize()
        want = 20.0 * 3
        print(f"rank {rank} trial {trial}: min {float(y.min())} max {float(y.max())} want {want}", flush=True)
    dist.barrier(); dist.destroy_process_group()
if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
