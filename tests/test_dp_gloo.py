"""CPU, world_size 2 over gloo: the data-parallel exchange of deepmerge_amd.trainer.PairTrainer
(flat parameter/gradient buffers, bucketed all-reduce launched from autograd hooks, averaging folded
into Adam) reproduces the single-process gradient of the global batch.  The compute kernels are HIP-only,
so a tiny CPU encoder built from the oracle stands in for the network; the exchange code is the product's."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def tiny_cfg():
    from oracle import s2former as O
    return O.S2Config(scales=(8, 16), in_c=2, depth=(1, 1, 1), grid=8, dim=32, heads=2, hidden=64, out_dim=12, n_designed=19)


class OracleNet(torch.nn.Module):
    """CPU stand-in with the v3 forward signature, parameters named/shaped by the oracle's manifest."""

    def __init__(self, cfg):
        super().__init__()
        from oracle import s2former as O
        self.cfg = cfg
        g = torch.Generator().manual_seed(7)
        self.names, self.bufs = [], {}
        for k, (shape, dt) in O.param_spec(cfg).items():
            if dt == "float32":
                self.names.append(k)
                self.register_parameter(k.replace(".", "__"), torch.nn.Parameter(torch.randn(shape, generator=g) * 0.1))
            else:
                self.bufs[k] = torch.from_numpy(O.relpos_index(cfg.cube(int(k[6]))))

    def forward(self, l, ld, r, rd):
        from oracle import s2former as O
        p = {k: getattr(self, k.replace(".", "__")) for k in self.names}
        p.update(self.bufs)
        return O.forward_pair(p, l, ld, r, rd, self.cfg)


def make_batch(B, cfg, seed):
    g = torch.Generator().manual_seed(seed)
    l = [torch.rand(B, cfg.in_c, s, s, generator=g) for s in cfg.scales]
    r = [torch.rand(B, cfg.in_c, s, s, generator=g) for s in cfg.scales]
    ld, rd = torch.rand(B, 1, 19, generator=g), torch.rand(B, 1, 19, generator=g)
    flag = (torch.arange(B) % 2).to(torch.int64)
    return l, ld, r, rd, flag


def cpu_adam(param, grad, m, v, step, lr, beta1, beta2, eps, grad_scale):
    from oracle import adam as OA
    OA.adam_step(param, grad * grad_scale, m, v, step, lr=lr, beta1=beta1, beta2=beta2, eps=eps)


def cpu_criterion(a, b, flag):
    from oracle import losses as OL
    return OL.contrastive_loss(a, b, flag, 50.0)    # large margin so the hinge branch is live


def _worker(rank, world, port, out, steps=2):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from deepmerge_amd.trainer import PairTrainer, shard_slice
    cfg = tiny_cfg()
    net = OracleNet(cfg)
    tr = PairTrainer(net, lr=1e-3, n_buckets=3, criterion=cpu_criterion, adam_fn=cpu_adam)
    assert tr.world == world and len(tr.bucket_slices) >= 2
    l, ld, r, rd, flag = make_batch(8, cfg, 99)
    sl = shard_slice(8, rank, world)
    losses = []
    for _ in range(steps):
        losses.append(float(tr.step([t[sl] for t in l], ld[sl], [t[sl] for t in r], rd[sl], flag[sl])))
    if rank == 0:
        torch.save({"flat": tr.fp.flat.clone(), "grad": tr.fp.grad.clone() / world, "losses": losses}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_global_batch(tmp_path):
    from deepmerge_amd.trainer import PairTrainer, shard_slice
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    cfg = tiny_cfg()
    net = OracleNet(cfg)
    tr = PairTrainer(net, lr=1e-3, criterion=cpu_criterion, adam_fn=cpu_adam)
    batch = make_batch(8, cfg, 99)
    for _ in range(2):
        loss = tr.step(*batch)
    # averaged shard gradients == gradient of the global-batch mean loss; same weights after 2 Adam steps
    np.testing.assert_allclose(got["grad"].numpy(), tr.fp.grad.numpy(), rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(got["flat"].numpy(), tr.fp.flat.numpy(), rtol=1e-4, atol=2e-5)
    assert shard_slice(8, 1, 2) == slice(4, 8)
    with pytest.raises(ValueError):
        shard_slice(9, 0, 2)


def test_four_rank_step_with_calibrated_bucket_hooks(tmp_path):
    """world_size 4, three steps: from the second step on the bucket hooks count only the parameters that received a gradient
    (buckets are exchanged from inside backward), which must stay equivalent to the single-process global batch."""
    from deepmerge_amd.trainer import PairTrainer
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(4, _free_port(), out, 3), nprocs=4, join=True)
    got = torch.load(out)
    cfg = tiny_cfg()
    tr = PairTrainer(OracleNet(cfg), lr=1e-3, criterion=cpu_criterion, adam_fn=cpu_adam)
    batch = make_batch(8, cfg, 99)
    for _ in range(3):
        tr.step(*batch)
    np.testing.assert_allclose(got["grad"].numpy(), tr.fp.grad.numpy(), rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(got["flat"].numpy(), tr.fp.flat.numpy(), rtol=2e-4, atol=5e-5)


def test_flat_params_views_and_buckets():
    from deepmerge_amd.trainer import FlatParams, multistep_lr
    net = OracleNet(tiny_cfg())
    names = [n for n, _ in net.named_parameters()]
    fp = FlatParams(net)
    assert fp.params[0] is dict(net.named_parameters())[names[-1]], "reverse (backward) order"
    for p, o in zip(fp.params, fp.offsets):
        assert p.data_ptr() == fp.flat.data_ptr() + 4 * o and p.grad.data_ptr() == fp.grad.data_ptr() + 4 * o and o % 64 == 0
    b = fp.buckets(4)
    assert b[0].start == 0 and b[-1].stop == fp.total and all(x.stop == y.start for x, y in zip(b[:-1], b[1:]))
    # unused parameters keep zero grads and are left untouched by Adam (m = v = 0)
    assert multistep_lr(1e-4, 0) == 1e-4 and abs(multistep_lr(1e-4, 40) - 2e-5) < 1e-12 and abs(multistep_lr(1e-4, 85) - 4e-6) < 1e-12
