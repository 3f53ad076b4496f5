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


def _worker(rank, world, port, out, steps=2, compress=None):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from deepmerge_amd.trainer import PairTrainer, shard_slice
    cfg = tiny_cfg()
    net = OracleNet(cfg)
    tr = PairTrainer(net, lr=1e-3, n_buckets=3, criterion=cpu_criterion, adam_fn=cpu_adam, compress_grads=compress)
    assert tr.world == world and len(tr.bucket_slices) >= 2
    l, ld, r, rd, flag = make_batch(8, cfg, 99)
    sl = shard_slice(8, rank, world)
    losses = []
    for _ in range(steps):
        losses.append(float(tr.step([t[sl] for t in l], ld[sl], [t[sl] for t in r], rd[sl], flag[sl])))
    if rank == 0:
        torch.save({"flat": tr.fp.flat.clone(), "grad": tr.fp.grad.clone() / world, "losses": losses, "bytes": tr.stats["allreduce_bytes"]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bf16_compressed_buckets(tmp_path):
    """compress_grads="bf16" (opt-in): buckets travel as bf16 -- half the bytes; the averaged gradient equals the exact
    one to bf16 rounding of the ranks' contributions and of their sum."""
    from deepmerge_amd.trainer import PairTrainer
    out, out32 = str(tmp_path / "c.pt"), str(tmp_path / "f.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, 1, "bf16"), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, _free_port(), out32, 1, None), nprocs=2, join=True)
    got, ref = torch.load(out), torch.load(out32)
    assert got["bytes"] * 2 == ref["bytes"]
    g, r = got["grad"], ref["grad"]
    assert not torch.equal(g, r)
    assert float((g - r).abs().max()) <= 2.0 ** -7 * float(r.abs().max())
    assert float((g - r).norm() / r.norm()) < 2.0 ** -8         # (entry-wise relative error is unbounded: the two ranks' terms can cancel)
    with pytest.raises(ValueError):
        PairTrainer(OracleNet(tiny_cfg()), compress_grads="fp8", criterion=cpu_criterion, adam_fn=cpu_adam)


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


def test_four_rank_step_bucketed_exchange(tmp_path):
    """world_size 4, three steps, a model without cut support: one backward, then the bucketed exchange of the whole buffer;
    must stay equivalent to the single-process global batch."""
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


class CutNet(torch.nn.Module):
    """CPU stand-in WITH cut support (the v3 contract: `_dp_cut(x, owner)` after every stage-0 block): 3 "stage-0" blocks, a tail,
    a parameter that is only used on odd steps (the used-parameter set changes between steps) and one that is never used."""
    _dp_cut = None

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        mk = lambda i, o: torch.nn.Linear(i, o)
        self.embed = mk(24, 32)
        self.blocks0 = torch.nn.ModuleList([mk(32, 32) for _ in range(3)])
        self.tail = mk(32, 12)
        self.sometimes = mk(32, 32)
        self.never = mk(4, 4)
        for p in self.parameters():
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        self.odd = False

    def once(self, x):
        x = torch.tanh(self.embed(x))
        for blk in self.blocks0:
            x = x + torch.tanh(blk(x))
            if self._dp_cut is not None:
                x = self._dp_cut(x, blk)
        if self.odd:
            x = x + 0.1 * torch.tanh(self.sometimes(x))
        return self.tail(x)

    def forward(self, l, ld, r, rd):
        return self.once(l[0]), self.once(r[0])


def _cut_batch(B):
    g = torch.Generator().manual_seed(11)
    return [torch.randn(B, 24, generator=g)], None, [torch.randn(B, 24, generator=g)], None, (torch.arange(B) % 2).to(torch.int64)


def _cut_worker(rank, world, port, out, steps):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from deepmerge_amd.trainer import PairTrainer, shard_slice
    net = CutNet()
    tr = PairTrainer(net, lr=1e-3, criterion=cpu_criterion, adam_fn=cpu_adam)
    assert tr.world == world and tr.segmented
    l, _, r, _, flag = _cut_batch(8)
    sl = shard_slice(8, rank, world)
    for i in range(steps):
        net.odd = bool(i % 2)
        tr.step([l[0][sl]], None, [r[0][sl]], None, flag[sl])
    if rank == 0:
        torch.save({"flat": tr.fp.flat.clone(), "grad": tr.fp.grad.clone() / world, "buckets": [(b.start, b.stop) for b in tr.bucket_slices],
                    "calls": tr.stats["allreduce_calls"]}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_segmented_backward_exchange(tmp_path, world):
    """The data-parallel schedule of the v3 family: backward in segments (one per stage-0 block + tail), one bucket per segment
    exchanged right after its segment, used-parameter set changing from step to step.  Result == single-process global batch."""
    from deepmerge_amd.trainer import PairTrainer
    out = str(tmp_path / "rank0.pt")
    steps = 4
    mp.spawn(_cut_worker, args=(world, _free_port(), out, steps), nprocs=world, join=True)
    got = torch.load(out)
    net = CutNet()
    tr = PairTrainer(net, lr=1e-3, criterion=cpu_criterion, adam_fn=cpu_adam)
    assert not tr.segmented                                   # one process: plain backward
    batch = _cut_batch(8)
    for i in range(steps):
        net.odd = bool(i % 2)
        tr.step(*batch)
    np.testing.assert_allclose(got["grad"].numpy(), tr.fp.grad.numpy(), rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(got["flat"].numpy(), tr.fp.flat.numpy(), rtol=2e-4, atol=2e-5)
    # 4 buckets (tail incl. `sometimes` / `never`, block 2, block 1, block 0 + embed), contiguous, whole buffer
    b = got["buckets"]
    assert len(b) == 4 and b[0][0] == 0 and b[-1][1] == tr.fp.total and all(x[1] == y[0] for x, y in zip(b[:-1], b[1:]))
    assert got["calls"] == 4 * steps
    # the segmented schedule on ONE process gives bit-identical gradients to the plain backward
    net2 = CutNet()
    tr2 = PairTrainer(net2, lr=1e-3, criterion=cpu_criterion, adam_fn=cpu_adam, segmented=True)
    for i in range(steps):
        net2.odd = bool(i % 2)
        tr2.step(*batch)
    assert torch.equal(tr2.fp.grad, tr.fp.grad) and torch.equal(tr2.fp.flat, tr.fp.flat)


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
