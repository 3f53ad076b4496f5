"""Data-parallel pair trainer: the MI355X counterpart of the loop body of the reference's
`train()` (Train_SMT.py:226-307: forward both sides -> Loss -> zero_grad -> backward -> Adam.step),
one process per GPU, gradients averaged with RCCL (`torch.distributed`, backend "nccl") over xGMI.

Memory layout (288 GB HBM3E per GPU makes replication free: 88 M params = 0.35 GB fp32):
  * ONE flat fp32 buffer holds every parameter, one holds every gradient, two hold Adam's m / v;
    each nn.Parameter (and its .grad) is a view into them.  Parameters are laid out in REVERSE
    forward order, so the gradients that backward produces first sit at the front of the buffer.
  * the gradient all-reduce runs on a few large contiguous buckets of that buffer (xGMI rings are
    per-link bound: few large transfers, not many small ones), each launched as soon as backward has
    written every gradient in it, overlapping the rest of backward;
  * Adam is a single fused launch over the flat buffers (dm_adam_step), which also applies the
    1/world_size averaging.
Pairs shard by contiguous equal slices of the global batch; the loss is a mean over pairs, so the
average of per-rank gradients equals the single-GPU gradient of the global batch.
Parameters that never receive a gradient (final_features.*, head.* on the designed-feature path,
SURVEY 8a M11) keep a zero gradient; with m = v = 0 Adam leaves them untouched, which matches
torch.optim.Adam skipping `grad is None` parameters.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import ops
from .Losses import Loss


def multistep_lr(base_lr: float, epoch: int, milestones=(40, 80), gamma: float = 0.2) -> float:
    """MultiStepLR([40, 80], 0.2) stepped per epoch (Train_SMT.py:194, :351)."""
    return base_lr * gamma ** sum(1 for m in milestones if epoch >= m)


def shard_slice(global_batch: int, rank: int, world: int) -> slice:
    """Contiguous equal shard of the pair batch owned by `rank` (both sides of a pair stay together)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} must divide evenly over {world} ranks "
                         f"(equal shards make mean-of-means equal the global mean)")
    per = global_batch // world
    return slice(rank * per, (rank + 1) * per)


class FlatParams:
    """Re-homes a module's parameters and gradients into flat fp32 buffers (views keep autograd working)."""

    def __init__(self, module: torch.nn.Module, align: int = 64, lp_mirror: bool = True):
        params = [p for p in module.parameters() if p.requires_grad]
        params.reverse()                      # backward order: last-used parameters first
        self.params = params
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + align - 1) // align * align
        dev = params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets, self.total = offs, total
        # bf16 mirror of the weights (MFMA operands), rewritten by the fused Adam kernel each step
        self.flat_lp = torch.zeros(total, dtype=torch.bfloat16, device=dev) if lp_mirror and dev.type == "cuda" else None
        for p, o in zip(params, offs):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view_as(p)
            p.grad = self.grad[o:o + n].view_as(p)
            if self.flat_lp is not None:
                p._dm_lp = self.flat_lp[o:o + n].view_as(p)
            if dev.type == "cuda":
                # fused backward kernels accumulate straight into the flat gradient buffer (ops._grad_out)
                p._dm_grad_sink = self.grad[o:o + n].view_as(p)
        self.refresh_lp()

    def refresh_lp(self):
        """Re-derive the bf16 mirror from the fp32 masters (after load_state_dict or any external update)."""
        if self.flat_lp is not None:
            self.flat_lp.copy_(ops.cast(self.flat, torch.bfloat16))

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):      # re-attach views if something replaced them
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def buckets(self, n_buckets: int) -> List[slice]:
        """Contiguous ranges of the flat buffer with roughly equal sizes, cut at parameter boundaries."""
        target = self.total / max(1, n_buckets)
        cuts, acc = [0], 0
        for p, o in zip(self.params, self.offsets):
            if o - cuts[-1] >= target and len(cuts) < n_buckets:
                cuts.append(o)
        cuts.append(self.total)
        return [slice(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]


def _v3_forward():
    from .nets.ShfitScaleFormer import ShfitScaleFormer_v3
    return ShfitScaleFormer_v3.forward


class PairTrainer:
    """One training step of the Siamese encoder on this rank's shard of the pair batch."""

    def __init__(self, net: torch.nn.Module, margin: float = 1.0, lr: float = 1e-4, lamda: float = 0.1, belta: float = 0,
                 betas=(0.9, 0.999), eps: float = 1e-8, n_buckets: int = 4, process_group=None, criterion=None, adam_fn=None):
        """`criterion` / `adam_fn` default to the HIP loss and fused Adam; tests of the exchange logic may
        inject stand-ins with the same signatures."""
        self.net = net
        self.criterion = criterion if criterion is not None else Loss(margin, lamda, belta)
        self.adam_fn = adam_fn if adam_fn is not None else ops.adam_step
        self.lr, self.betas, self.eps = lr, betas, eps
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.fp = FlatParams(net, lp_mirror=(getattr(net, "numerics", "bf16") == "bf16"))
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.step_count = 0
        self.bucket_slices = self.fp.buckets(n_buckets if self.world > 1 else 1)
        self._pending = []
        self._hooks_installed = False
        if self.world > 1:
            self._install_bucket_hooks()
        self._graph = None          # captured step (enable_graph)
        self._graph_warm = 0
        self._capture = None        # {"side": stream, "hyper": tensor} while a graph with overlapped Adam is being captured

    # -- gradient exchange -----------------------------------------------------------------------
    def _install_bucket_hooks(self):
        """Launch a bucket's all-reduce from autograd as soon as its last gradient has been accumulated."""
        fp = self.fp
        self._bucket_of, self._remaining0 = {}, []
        for bi, sl in enumerate(self.bucket_slices):
            members = [p for p, o in zip(fp.params, fp.offsets) if sl.start <= o < sl.stop]
            self._remaining0.append(len(members))
            for p in members:
                self._bucket_of[p] = bi
        self._remaining = list(self._remaining0)

        self._seen = set()
        self._calibrated = False

        def hook(p):
            # idempotent per step: a parameter whose gradient went straight to the sink is reported by the fused
            # backward AND may be reported again by autograd's own post-accumulate hook
            if id(p) in self._seen:
                return
            self._seen.add(id(p))
            bi = self._bucket_of[p]
            self._remaining[bi] -= 1
            if self._remaining[bi] == 0:
                self._launch_bucket(bi)
        for p in fp.params:
            p.register_post_accumulate_grad_hook(hook)
            p._dm_grad_ready = hook          # same notification when a fused backward wrote the sink directly
        self._hooks_installed = True

    def _launch_bucket(self, bi: int):
        sl = self.bucket_slices[bi]
        if self.world > 1:
            self._pending.append((bi, dist.all_reduce(self.fp.grad[sl], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)))
        elif self._capture is not None:
            # single GPU, graph capture: this bucket's gradients are final and its weights are not read again in this
            # backward -> its Adam update runs on a side stream next to the rest of the backward (Adam is HBM-bound, the
            # GEMMs are not)
            cap = self._capture
            cap["side"].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap["side"]):
                self._adam_slice(sl, cap["hyper"])
            self._pending.append((bi, None))

    def _adam_slice(self, sl: slice, hyper_dev):
        lp = self.fp.flat_lp[sl] if self.fp.flat_lp is not None else None
        ops.adam_step_dev(self.fp.flat[sl], self.fp.grad[sl], self.m[sl], self.v[sl], hyper_dev, beta1=self.betas[0], beta2=self.betas[1],
                          eps=self.eps, grad_scale=1.0, param_lp=lp)

    def _finish_exchange(self):
        launched = {bi for bi, _ in self._pending}
        for bi in range(len(self.bucket_slices)):        # buckets whose params got no gradient this step
            if bi not in launched:
                self._launch_bucket(bi)
        for _, work in self._pending:
            if work is not None:
                work.wait()
        self._pending.clear()
        if not self._calibrated:
            # parameters that never receive a gradient (final_features.*, head.* on the designed-feature path) would
            # hold their bucket back until the end of backward: after the first step, count only the used ones
            counts = [0] * len(self.bucket_slices)
            for p in self.fp.params:
                if id(p) in self._seen:
                    counts[self._bucket_of[p]] += 1
            self._remaining0 = counts
            self._calibrated = True
        self._remaining = list(self._remaining0)
        self._seen.clear()

    # -- the step ----------------------------------------------------------------------------------
    # -- hipGraph replay of the whole step ---------------------------------------------------------
    def enable_graph(self, warmup: int = 3, overlap_adam: bool = False, adam_buckets: int = 4):
        """Capture forward + loss + backward + Adam of one step into a hipGraph (torch.cuda.CUDAGraph) after `warmup`
        eager steps, and replay it afterwards: ~330 launches per step collapse into one submission, which removes the
        host-side gaps between the many small kernels.  Needs fixed input shapes; single-GPU only (the bucketed
        exchange of the data-parallel path stays eager)."""
        # overlap_adam: run Adam bucket by bucket on a side stream (a second branch of the graph) as soon as a bucket's gradients
        # are final.  Measured on MI355X: 7.10 - 7.26 ms/step against 6.84 ms with the single-branch graph -- the fork / join
        # edges and the HBM contention cost more than the 0.26 ms of Adam they hide -- so it is off by default
        # (DM_ADAM_OVERLAP=<buckets> switches it on for experiments).
        if self.world > 1:
            raise RuntimeError("graph replay covers the single-GPU step; the data-parallel exchange runs eagerly")
        if self.fp.flat.device.type != "cuda":
            raise RuntimeError("graph capture needs the parameters on the GPU")
        env = os.environ.get("DM_ADAM_OVERLAP")
        if env is not None:
            overlap_adam, adam_buckets = env != "0", (int(env) if env.isdigit() and int(env) > 1 else adam_buckets)
        self._graph = {"warmup": warmup, "g": None, "overlap_adam": bool(overlap_adam)}
        self._graph_warm = 0
        if overlap_adam:
            # Adam per bucket of the flat buffer, launched from the gradient-ready hooks on a side stream of the captured graph
            # (a graph branch that runs next to the remaining backward).  Bucket boundaries are 16-byte aligned views.
            self.bucket_slices = self.fp.buckets(adam_buckets)
            if not self._hooks_installed:
                self._install_bucket_hooks()

    def _graph_step(self, left, left_designed, right, right_designed, flag, lr):
        st = self._graph
        args = (list(left), left_designed, list(right), right_designed, flag)
        if st["g"] is None:
            if self._graph_warm < st["warmup"]:           # eager steps first: lazy caches, workspaces, kernel attributes
                self._graph_warm += 1
                return self._eager_step(*args, lr)
            dev = self.fp.flat.device
            # models that take the two sides pre-stacked ([left; right] along the batch) get static buffers of that form, so
            # the step's inputs are copied once and the captured graph holds no torch.cat
            batched = type(self.net).__dict__.get("forward_pair_batched") is not None or \
                (hasattr(self.net, "forward_pair_batched") and type(self.net).forward is _v3_forward())
            if batched:
                both = [torch.cat((l, r), 0) for l, r in zip(left, right)]
                Bp = left[0].shape[0]
                st["left"], st["right"] = [t[:Bp] for t in both], [t[Bp:] for t in both]
                st["both"] = both
                if left_designed is not None:
                    st["dboth"] = torch.cat((left_designed, right_designed), 0)
                    st["ld"], st["rd"] = st["dboth"][:Bp], st["dboth"][Bp:]
                else:
                    st["dboth"] = st["ld"] = st["rd"] = None
            else:
                st["both"] = None
                st["left"] = [t.clone() for t in left]
                st["right"] = [t.clone() for t in right]
                st["ld"] = None if left_designed is None else left_designed.clone()
                st["rd"] = None if right_designed is None else right_designed.clone()
            st["flag"] = flag.clone()
            st["hyper"] = torch.zeros(2, dtype=torch.float32, device=dev)
            st["shapes"] = [tuple(t.shape) for t in st["left"] + st["right"]]
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                self.fp.zero_grad()
                if st["both"] is not None:
                    fa, fb = self.net.forward_pair_batched(st["both"], st["dboth"])
                else:
                    fa, fb = self.net(st["left"], st["ld"], st["right"], st["rd"])
                loss = self.criterion(fa, fb, st["flag"])
                if st["overlap_adam"]:
                    self._capture = {"side": torch.cuda.Stream(), "hyper": st["hyper"]}
                    try:
                        loss.backward()                    # hooks launch Adam bucket by bucket on the side stream
                        self._finish_exchange()            # buckets that saw no gradient (unused parameters)
                        torch.cuda.current_stream().wait_stream(self._capture["side"])
                    finally:
                        self._capture = None
                else:
                    loss.backward()
                    self._adam_slice(slice(0, self.fp.total), st["hyper"])
                st["loss"] = loss.detach()
            st["g"] = g
            # capture only records: nothing above has executed yet, the replay below is this step
        if [tuple(t.shape) for t in list(left) + list(right)] != st["shapes"]:
            raise ValueError("graph replay needs the input shapes it was captured with; call enable_graph() again for a new batch size")
        for dst, src in zip(st["left"] + st["right"], list(left) + list(right)):
            dst.copy_(src, non_blocking=True)
        if st["ld"] is not None:
            st["ld"].copy_(left_designed, non_blocking=True)
            st["rd"].copy_(right_designed, non_blocking=True)
        st["flag"].copy_(flag, non_blocking=True)
        self.step_count += 1
        st["hyper"].copy_(ops.adam_hyper(self.step_count, self.lr if lr is None else lr, self.betas[0], self.betas[1]), non_blocking=True)
        st["g"].replay()
        return st["loss"]

    def step(self, left: Sequence[torch.Tensor], left_designed, right: Sequence[torch.Tensor], right_designed, flag,
             lr: Optional[float] = None) -> torch.Tensor:
        """forward -> Loss -> zero_grad -> backward -> (all-reduce) -> Adam.  Returns the local loss tensor
        (no host sync; the reference's per-step `.item()` at Train_SMT.py:301 is left to the caller).
        After enable_graph() the same work is replayed from a captured hipGraph (the returned tensor is then a
        static buffer that the next step overwrites)."""
        if self._graph is not None:
            self.net.train()
            return self._graph_step(left, left_designed, right, right_designed, flag, lr)
        return self._eager_step(left, left_designed, right, right_designed, flag, lr)

    def _eager_step(self, left, left_designed, right, right_designed, flag, lr=None) -> torch.Tensor:
        self.net.train()
        self.fp.zero_grad()
        fa, fb = self.net(left, left_designed, right, right_designed)
        loss = self.criterion(fa, fb, flag)
        loss.backward()
        if self.world > 1:
            self._finish_exchange()
        self.step_count += 1
        extra = {"param_lp": self.fp.flat_lp} if self.fp.flat_lp is not None else {}
        self.adam_fn(self.fp.flat, self.fp.grad, self.m, self.v, self.step_count, lr=self.lr if lr is None else lr,
                     beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, grad_scale=1.0 / self.world, **extra)
        if self.world == 1 and self._hooks_installed:
            self._finish_exchange()        # gradient-ready hooks are installed for the graph's overlapped Adam: reset their counters
        return loss.detach()
