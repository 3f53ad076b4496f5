"""Data-parallel pair trainer: the MI355X counterpart of the loop body of the reference's
`train()` (Train_SMT.py:226-307: forward both sides -> Loss -> zero_grad -> backward -> Adam.step),
one process per GPU, gradients averaged with RCCL (`torch.distributed`, backend "nccl") over xGMI.

Memory layout (288 GB HBM3E per GPU makes replication free: 88 M params = 0.35 GB fp32):
  * ONE flat fp32 buffer holds every parameter, one holds every gradient, two hold Adam's m / v;
    each nn.Parameter (and its .grad) is a view into them.  Parameters are laid out in REVERSE
    forward order, so the gradients that backward produces first sit at the front of the buffer.
  * the gradient all-reduce runs on a few large contiguous buckets of that buffer (xGMI rings are
    per-link bound: few large transfers, not many small ones): the backward pass runs in segments (cut
    after every stage-0 block) and a segment's bucket is exchanged while the earlier layers' backward runs;
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


_SINK_VIOLATION = ("first-write gradient sink violated: a parameter of a fused block received a gradient outside the block kernels (its "
                   ".grad was replaced, or autograd accumulated into it); build the trainer with first_write=False for models that use "
                   "these parameters elsewhere")


class FlatParams:
    """Re-homes a module's parameters and gradients into flat fp32 buffers (views keep autograd working)."""

    def __init__(self, module: torch.nn.Module, align: int = 64, lp_mirror: bool = True, first_write: Optional[bool] = None,
                 pair_mirror: bool = False):
        """first_write (default: on unless DM_GRAD_FIRST_WRITE=0): for models whose fused blocks are the only writers of their
        parameters' gradients (`_dm_first_write_blocks` on the model class, `_dm_fused_block` on the block class) those gradients
        are not zeroed at the start of a step; the first write of the step stores instead (ops._acc).  Saves the 195 MB memset and
        the read of every weight gradient by its own first accumulation."""
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
        # "bf16x3": the weights as hi / lo plane pairs [2, total] (plane 0 = bf16(w), plane 1 = bf16(w - plane 0)), rewritten by the Adam
        # kernel; ops.pair_weight hands the blocks [2, rows, cols] views of it (plane stride = total)
        self.flat_pair = (torch.zeros((2, total), dtype=torch.bfloat16, device=dev)
                          if pair_mirror and dev.type == "cuda" and total % 8 == 0 and total < (1 << 29) else None)
        for p, o in zip(params, offs):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view_as(p)
            p.grad = self.grad[o:o + n].view_as(p)
            if self.flat_lp is not None:
                p._dm_lp = self.flat_lp[o:o + n].view_as(p)
            if self.flat_pair is not None:
                p._dm_pair_src = (self.flat_pair, o)
            if dev.type == "cuda":
                # fused backward kernels accumulate straight into the flat gradient buffer (ops._grad_out)
                p._dm_grad_sink = self.grad[o:o + n].view_as(p)
        # ---- first-write gradient sinks ----------------------------------------------------------------------------------
        if first_write is None:
            first_write = os.environ.get("DM_GRAD_FIRST_WRITE", "1") != "0"
        self.tracked = []                     # (param, offset, numel) of the parameters that are not zeroed by zero_grad()
        if first_write and dev.type == "cuda":
            ids = set()
            for model in module.modules():          # the module itself or a model wrapped inside it (adapters around the pair signature)
                if getattr(model, "_dm_first_write_blocks", False):
                    for sub in model.modules():
                        if getattr(sub, "_dm_fused_block", False):
                            ids.update(id(q) for q in sub.parameters())
            for p, o in zip(params, offs):
                if id(p) in ids:
                    p._dm_gw = [False]
                    self.tracked.append((p, o, p.numel()))
            names = {id(q): n for n, q in module.named_parameters()}
            for p, _, _ in self.tracked:
                p._dm_name = names.get(id(p), "?")
        # what zero_grad() still has to clear: the maximal runs of untracked parameters (alignment gaps included)
        self._zero_ranges, tr = [], {o for _, o, _ in self.tracked}
        run = None
        for p, o in zip(params, offs):
            end = o + (p.numel() + align - 1) // align * align
            if o in tr:
                if run is not None:
                    self._zero_ranges.append(tuple(run)); run = None
            else:
                run = [o, end] if run is None else [run[0], end]
        if run is not None:
            self._zero_ranges.append(tuple(run))
        self.refresh_lp()

    def refresh_lp(self):
        """Re-derive the bf16 mirror from the fp32 masters (after load_state_dict or any external update)."""
        if self.flat_lp is not None:
            self.flat_lp.copy_(ops.cast(self.flat, torch.bfloat16))
        if self.flat_pair is not None:
            cols = 64
            while cols < 8192 and self.total % (2 * cols) == 0:
                cols *= 2
            self.flat_pair.copy_(ops.split_planes(self.flat.view(self.total // cols, cols)).t.view(2, self.total))

    def zero_grad(self):
        if self.tracked:
            for lo, hi in self._zero_ranges:
                self.grad[lo:hi].zero_()
            for p, _, _ in self.tracked:
                p._dm_gw[0] = False
        else:
            self.grad.zero_()
        for p, o in zip(self.params, self.offsets):      # re-attach views if something replaced them
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)
        # Tracked parameters go through the backward pass WITHOUT a .grad: the fused blocks write their sinks through
        # `_dm_grad_sink` and hand autograd nothing, so a .grad that exists afterwards was produced by autograd -- by some other
        # use of the parameter -- and finish_grads() refuses it (it would otherwise have been added to last step's values).
        for p, _, _ in self.tracked:
            p.grad = None

    def finish_grads(self, lo: int = 0, hi: Optional[int] = None) -> int:
        """Call after the backward pass (of the range [lo, hi) of the flat buffer) and before its gradients are used: a tracked
        parameter that received no gradient this step still holds last step's values -- clear it.  Returns how many there were."""
        hi = self.total if hi is None else hi
        n = 0
        for p, o, cnt in self.tracked:
            if not lo <= o < hi:
                continue
            # The first-write contract: a tracked parameter's gradient is written by the fused-block kernels only, through the raw
            # pointer of its sink view (they store on the first write of a step).  zero_grad() detached the parameter's .grad, so
            # one that exists now came from autograd -- another use of the parameter.  (No hook on the parameter: a hook keeps its
            # AccumulateGrad node alive across steps, on the stream it was created on, which breaks stream capture.)
            if p.grad is not None and p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                raise RuntimeError(f"{_SINK_VIOLATION} [parameter {getattr(p, '_dm_name', '?')} {tuple(p.shape)}]")
            p.grad = self.grad[o:o + cnt].view_as(p)         # what optimizers / callers read after the step
            if not p._dm_gw[0]:
                self.grad[o:o + cnt].zero_()
                p._dm_gw[0] = True
                n += 1
        return n

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


class _Cuts:
    """Autograd cut points of one forward pass.  The model calls `net._dp_cut(x)` at its segment boundaries (after every stage-0
    block for the v3 family); the value that flows on is a detached leaf, so the backward pass can be run segment by segment:
    loss.backward() stops at the last cut, then `x.backward(leaf.grad)` continues through the segment before it, and so on."""

    def __init__(self):
        self.pairs = []            # (tensor produced by the segment, detached leaf the next segment consumed), forward order
        self.owners = []           # the module after which each cut was taken

    def __call__(self, x, owner=None):
        leaf = x.detach().requires_grad_(True)
        self.pairs.append((x, leaf))
        self.owners.append(owner)
        return leaf


def _unit(loss: torch.Tensor):
    """The gradient to start a scalar fp32 loss's backward pass with: ops.unit_grad -- the loss functions recognise it and skip the
    scalar multiply (and autograd creates no ones_like); anything else starts the default way."""
    return ops.unit_grad(loss.device) if (loss.dim() == 0 and loss.dtype == torch.float32) else None


class PairTrainer:
    """One training step of the Siamese encoder on this rank's shard of the pair batch.

    Data-parallel exchange (world > 1): the backward pass runs in SEGMENTS (see _Cuts); the flat gradient buffer is cut into
    one bucket per segment (parameters are laid out in reverse forward order, so a segment's parameters are one contiguous
    range), and a bucket's all-reduce is launched as soon as its segment's backward has been enqueued -- it then runs on the
    collective's stream next to the backward of the earlier layers.  Which parameters received a gradient plays no role (a
    bucket is exchanged whole; an unused parameter contributes zeros), so the schedule is identical on every rank and every
    step by construction.  With enable_graph() every compute piece between two exchange launches (forward + loss + first backward
    segment, each further backward segment, Adam) is a captured hipGraph; the collectives themselves stay eager."""

    def __init__(self, net: torch.nn.Module, margin: float = 1.0, lr: float = 1e-4, lamda: float = 0.1, belta: float = 0,
                 betas=(0.9, 0.999), eps: float = 1e-8, n_buckets: int = 4, process_group=None, criterion=None, adam_fn=None,
                 segmented: Optional[bool] = None, first_write: Optional[bool] = None, compress_grads: Optional[str] = None,
                 overlap_adam: bool = False):
        """`compress_grads="bf16"` (opt-in, lossy; SURVEY 8e "optionally bf16-compressed buckets in throughput mode"): a bucket is
        rounded to bf16, summed across ranks in bf16 and widened again before Adam -- half the bytes on the links for ~2^-9 relative
        rounding per rank's contribution; the default exchanges fp32.
        `first_write`: see FlatParams.  `criterion` / `adam_fn` default to the HIP loss and fused Adam; tests of the exchange logic may
        inject stand-ins with the same signatures.  `segmented`: run the backward pass in segments (default: world > 1 and
        the model supports cuts); `n_buckets` is the bucket count for models without cut support."""
        self.net = net
        self.criterion = criterion if criterion is not None else Loss(margin, lamda, belta)
        ops.unit_grad(next(net.parameters()).device)      # (exists before any capture: creating it is a fill launch)
        self.adam_fn = adam_fn if adam_fn is not None else ops.adam_step
        self.lr, self.betas, self.eps = lr, betas, eps
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # DM_DP_FORCE=1 (rehearsal): run the data-parallel schedule -- segmented backward, bucket all-reduces through the
        # initialised backend, per-segment graphs -- even with ONE rank, so that a one-GPU box exercises the RCCL calls
        self.force_dp = os.environ.get("DM_DP_FORCE") == "1" and dist.is_available() and dist.is_initialized()
        self.fp = FlatParams(net, lp_mirror=(getattr(net, "numerics", "bf16") == "bf16"), first_write=first_write,
                             pair_mirror=(getattr(net, "numerics", "bf16") == "bf16x3" and os.environ.get("DM_X3_PAIR_MIRROR", "1") != "0"))
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.step_count = 0
        self.can_cut = hasattr(net, "_dp_cut")
        self.dp = self.world > 1 or self.force_dp
        # one GPU, graph replay: `overlap_adam=True` (or DM_ADAM_OVERLAP=1) runs the backward in segments inside ONE graph with each
        # bucket's Adam update as a side branch (_capture_overlapped).  Off by default: measured 5.88 -> 6.15 ms per step on the headline
        # model -- the streaming update takes more from the GEMMs it runs beside (L2 / fabric) than its own 0.23 ms.
        self.overlap_adam = (bool(overlap_adam) or os.environ.get("DM_ADAM_OVERLAP", "0") == "1") and not self.dp and self.can_cut
        if segmented is None:
            segmented = self.dp or self.overlap_adam
        self.segmented = (bool(segmented) or self.overlap_adam) and self.can_cut
        self.n_buckets = max(1, n_buckets)
        self.bucket_slices = self.fp.buckets(self.n_buckets if self.dp else 1)     # re-derived from the cuts when segmented
        self._pending = []
        self._graph = None          # captured step (enable_graph)
        self._graph_warm = 0
        self.exchange = True        # False: skip the collectives (bench.py uses it to price the exposed exchange time)
        self.trace = os.environ.get("DM_DP_TRACE") == "1"
        if compress_grads not in (None, "bf16"):
            raise ValueError(f"compress_grads must be None or 'bf16', got {compress_grads!r}")
        self.compress_grads = compress_grads
        self._cbuf = torch.empty_like(self.fp.grad, dtype=torch.bfloat16) if (compress_grads and self.dp) else None
        self.stats = {"allreduce_calls": 0, "allreduce_bytes": 0}
        self.graph_error = None     # set when enable_graph() had to fall back to eager launches

    # -- gradient exchange -----------------------------------------------------------------------
    def _log(self, msg):
        if self.trace:
            import sys
            import time
            rank = dist.get_rank(self.pg) if self.world > 1 else 0
            print(f"[dp {time.strftime('%H:%M:%S')}.{int(time.time() * 1000) % 1000:03d} rank {rank}] {msg}", file=sys.stderr, flush=True)

    def _launch_bucket(self, bi: int):
        sl = self.bucket_slices[bi]
        self.fp.finish_grads(sl.start, sl.stop)
        if self.dp and self.exchange:
            self._log(f"launch bucket {bi} [{sl.start}:{sl.stop}] ({(sl.stop - sl.start) * 4 / 1e6:.1f} MB)")
            if self._cbuf is not None:
                self._cbuf[sl].copy_(self.fp.grad[sl])                   # fp32 -> bf16 (round to nearest even)
                self._pending.append((bi, dist.all_reduce(self._cbuf[sl], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)))
            else:
                self._pending.append((bi, dist.all_reduce(self.fp.grad[sl], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)))
            self.stats["allreduce_calls"] += 1
            self.stats["allreduce_bytes"] += (sl.stop - sl.start) * (2 if self._cbuf is not None else 4)

    def _wait_exchange(self):
        for bi, work in self._pending:
            self._log(f"wait bucket {bi}")
            work.wait()
            if self._cbuf is not None:
                sl = self.bucket_slices[bi]
                self.fp.grad[sl].copy_(self._cbuf[sl])
        self._log("exchange complete")
        self._pending.clear()

    def _finish_buckets(self, update):
        """Per bucket, in launch order: wait for its all-reduce, then `update(bucket index)` (Adam on that range).  The optimizer
        work of the early buckets (the last layers) thus runs while the later buckets -- the first layers, whose gradients are
        ready last -- are still being exchanged: only the last bucket's update stays behind the exchange.  (A collective's
        wait() orders the compute stream behind it; it does not block the host for RCCL.)"""
        pending = dict(self._pending)
        for bi in range(len(self.bucket_slices)):
            work = pending.get(bi)
            if work is not None:
                self._log(f"wait bucket {bi}")
                work.wait()
                if self._cbuf is not None:
                    sl = self.bucket_slices[bi]
                    self.fp.grad[sl].copy_(self._cbuf[sl])               # the bf16 sum, widened for the fp32 Adam
            update(bi)
        self._log("exchange complete")
        self._pending.clear()

    def _segment_buckets(self, cuts: _Cuts):
        """One bucket per distinct cut owner + one for the tail.  A cut taken after module `owner` ends the segment that
        contains `owner`: in the flat buffer (reverse forward order) that segment starts at the lowest offset of owner's
        parameters.  Returns (slices, ready_after): bucket j may be exchanged after backward piece ready_after[j] (piece 0 =
        loss.backward() down to the last cuts, piece k = the k-th cut from the end).  A model that runs its encoder once per
        side meets every owner twice; its bucket is ready after the LAST piece that touches it."""
        off = {id(p): o for p, o in zip(self.fp.params, self.fp.offsets)}
        start_of, last_piece = {}, {}
        for k, owner in enumerate(reversed(cuts.owners)):
            ps = [off[id(p)] for p in owner.parameters() if id(p) in off] if owner is not None else []
            if not ps:
                raise RuntimeError("a DP cut needs an owner module with trainable parameters to place the bucket boundary")
            start_of[id(owner)] = min(ps)
            last_piece[id(owner)] = k + 1
        order = sorted(start_of, key=lambda o: start_of[o])
        starts = [start_of[o] for o in order]
        if starts and starts[0] <= 0:
            raise RuntimeError(f"DP cuts do not follow the flat parameter order: {starts}")
        edges = [0] + starts + [self.fp.total]
        slices = [slice(a, b) for a, b in zip(edges[:-1], edges[1:])]
        ready = [0] + [last_piece[o] for o in order]
        if len(ready) > 1:
            ready[-1] = len(cuts.pairs)          # the last range also holds everything in front of the first cut
        if any(y < x for x, y in zip(ready[:-1], ready[1:])):
            raise RuntimeError(f"DP cuts: buckets would complete out of order {ready}")
        return slices, ready

    def _launch_ready(self, piece: int):
        for bi, r in enumerate(self._ready_after):
            if r == piece:
                self._launch_bucket(bi)

    def _backward_segments(self, loss, cuts: _Cuts):
        """Generator over the backward pieces: yields the index of the piece that has just been enqueued."""
        loss.backward(_unit(loss))
        yield 0
        for k, (x, leaf) in enumerate(reversed(cuts.pairs)):
            x.backward(leaf.grad)
            yield k + 1

    # -- hipGraph replay of the whole step ---------------------------------------------------------
    def enable_graph(self, warmup: int = 3):
        """Capture the step's compute into hipGraphs (torch.cuda.CUDAGraph) after `warmup` eager steps and replay them afterwards:
        ~330 launches per step collapse into one submission per piece, which removes the host-side gaps between the many small
        kernels.  One GPU: a single graph (forward + loss + backward + Adam).  Data parallel: one graph per backward segment
        (the first also holds zero_grad + forward + loss) and one for Adam, with the bucket all-reduces launched eagerly in
        between -- the exchange overlaps the graphs that follow.  Needs fixed input shapes."""
        if self.fp.flat.device.type != "cuda":
            raise RuntimeError("graph capture needs the parameters on the GPU")
        self._graph = {"warmup": warmup, "g": None}
        self._graph_warm = 0

    def graph_inputs(self):
        """The static input tensors of the captured step, (left list, left_designed, right list, right_designed, flag), or None
        before capture.  A data pipeline that writes the next batch straight into them (e.g. the patch gather kernels) and then
        calls step() with these very tensors skips the per-step device-to-device input copies."""
        st = self._graph
        if not st or st.get("g") is None:
            return None
        return st["left"], st["ld"], st["right"], st["rd"], st["flag"]

    def graph_inputs_both(self):
        """(both, designed_both, flag) of the captured step for models that take the two sides stacked along the batch ([left; right],
        2B samples; the v3 family): the buffers a feed fills in ONE gather launch per scale (deepmerge_amd/feed.py).  None before
        capture or for other models."""
        st = self._graph
        if not st or st.get("g") is None or st.get("both") is None:
            return None
        return st["both"], st["dboth"], st["flag"]

    def _static_inputs(self, st, left, left_designed, right, right_designed, flag):
        # models that take the two sides pre-stacked ([left; right] along the batch) get static buffers of that form, so
        # the step's inputs are copied once and the captured graph holds no torch.cat
        batched = type(self.net).__dict__.get("forward_pair_batched") is not None or \
            (hasattr(self.net, "forward_pair_batched") and type(self.net).forward is _v3_forward())
        if batched:
            both = [ops.cat_batch(l, r) for l, r in zip(left, right)]      # (image tensors or ops.PatchCols: patch-embed rows from a feed)
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
            if any(isinstance(t, ops.PatchCols) for t in list(left) + list(right)):
                raise ValueError("ops.PatchCols inputs need a model with forward_pair_batched (the v3 family)")
            st["left"] = [t.clone() for t in left]
            st["right"] = [t.clone() for t in right]
            st["ld"] = None if left_designed is None else left_designed.clone()
            st["rd"] = None if right_designed is None else right_designed.clone()
        st["flag"] = flag.to(torch.float32).clone() if not flag.is_floating_point() else flag.clone()     # (the loss kernel reads fp32 flags: cast once here, not per step)
        st["hyper"] = torch.zeros(2, dtype=torch.float32, device=self.fp.flat.device)
        st["shapes"] = [tuple(t.shape) for t in st["left"] + st["right"]]

    def _forward_loss(self, st):
        with ops.module_products(self.net):          # forward_pair_batched is not `__call__`: the module's forward hooks do not fire
            if st["both"] is not None:
                fa, fb = self.net.forward_pair_batched(st["both"], st["dboth"])
            else:
                fa, fb = self.net(st["left"], st["ld"], st["right"], st["rd"])
            return self.criterion(fa, fb, st["flag"])

    def _capture(self, st):
        torch.cuda.synchronize()
        if not self.segmented:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.fp.zero_grad()
                loss = self._forward_loss(st)
                loss.backward(_unit(loss))
                self.fp.finish_grads()
                if not self.dp:
                    self._adam_slice(slice(0, self.fp.total), st["hyper"], 1.0)
                st["loss"] = loss.detach()
            st["pieces"] = [g]
            if self.dp:                              # no cut support: exchange the whole buffer after the one backward graph
                st["adam"] = self._capture_adam(st, g.pool())
            return
        if self.overlap_adam:
            return self._capture_overlapped(st)
        cuts = _Cuts()
        self.net._dp_cut = cuts
        pieces = []
        try:
            g0 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g0, capture_error_mode="thread_local"):
                self.fp.zero_grad()
                loss = self._forward_loss(st)
                loss.backward(_unit(loss))
                st["loss"] = loss.detach()
            pieces.append(g0)
            for x, leaf in reversed(cuts.pairs):
                gk = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gk, pool=g0.pool(), capture_error_mode="thread_local"):
                    x.backward(leaf.grad)
                pieces.append(gk)
        finally:
            self.net._dp_cut = None
        self.fp.finish_grads()      # (a tracked parameter no captured piece writes is cleared once here and stays clear: nothing replays a write to it)
        st["pieces"], st["cuts"] = pieces, cuts                            # the cut tensors keep the autograd segments alive
        self.bucket_slices, self._ready_after = self._segment_buckets(cuts)
        st["adam"] = self._capture_adam(st, g0.pool())
        # the retired-workspace list of ops.workspace keeps every scratch buffer these graphs point into alive

    def _capture_overlapped(self, st):
        """One GPU, segmented backward (`PairTrainer(segmented=True)`): ONE graph in which the Adam update of a bucket is a side
        branch that starts as soon as the backward piece that completes the bucket's gradients has been enqueued.  Adam is a
        pure streaming kernel (7 fp32 passes + the bf16 mirror over the bucket, HBM-bound) and the backward pieces that follow
        are matrix-pipe work on other layers' weights, so the two share the chip; only the last bucket's update (the first
        layers, whose gradients complete last) runs alone at the end of the step.  Same arithmetic as the unsegmented step."""
        cuts = _Cuts()
        self.net._dp_cut = cuts
        side = torch.cuda.Stream(device=self.fp.flat.device)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                main = torch.cuda.current_stream()
                self.fp.zero_grad()
                loss = self._forward_loss(st)
                self.bucket_slices, self._ready_after = self._segment_buckets(cuts)
                forked = False
                for piece in self._backward_segments(loss, cuts):
                    ready = [bi for bi, r in enumerate(self._ready_after) if r == piece]
                    if not ready:
                        continue
                    side.wait_stream(main)               # the piece's kernels (incl. its batched reductions) are enqueued on `main`
                    forked = True
                    with torch.cuda.stream(side):
                        for bi in ready:
                            sl = self.bucket_slices[bi]
                            self.fp.finish_grads(sl.start, sl.stop)
                            self._adam_slice(sl, st["hyper"], 1.0)
                if forked:
                    main.wait_stream(side)
                st["loss"] = loss.detach()
        finally:
            self.net._dp_cut = None
        st["pieces"], st["cuts"] = [g], cuts

    def _capture_adam(self, st, pool):
        """One captured Adam launch per bucket (see _finish_buckets)."""
        graphs = []
        for sl in self.bucket_slices:
            ga = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, pool=pool, capture_error_mode="thread_local"):
                self._adam_slice(sl, st["hyper"], 1.0 / self.world)
            graphs.append(ga)
        return graphs

    def _graph_step(self, left, left_designed, right, right_designed, flag, lr):
        st = self._graph
        args = (list(left), left_designed, list(right), right_designed, flag)
        if st["g"] is None:
            if self._graph_warm < st["warmup"]:           # eager steps first: lazy caches, workspaces, kernel attributes
                self._graph_warm += 1
                return self._eager_step(*args, lr)
            self._static_inputs(st, *args)
            try:
                self._capture(st)
            except Exception as e:                        # e.g. a collective backend whose helper threads break stream capture
                import sys
                if os.environ.get("DM_GRAPH_DEBUG") == "1":
                    raise
                print(f"[deepmerge_amd] hipGraph capture failed ({type(e).__name__}: {e}); the step stays eager", file=sys.stderr, flush=True)
                torch.cuda.synchronize()
                if getattr(self.net, "_dp_cut", None) is not None:
                    self.net._dp_cut = None
                self._graph = None
                self.graph_error = f"{type(e).__name__}: {e}"
                return self._eager_step(*args, lr)
            st["g"] = True
            # capture only records: nothing above has executed yet, the replay below is this step
        if [tuple(t.shape) for t in list(left) + list(right)] != st["shapes"]:
            raise ValueError("graph replay needs the input shapes it was captured with; call enable_graph() again for a new batch size")
        def put(dst, src):                                    # a loader that fills graph_inputs() directly pays no copy
            if isinstance(dst, ops.PatchCols) or isinstance(src, ops.PatchCols):
                if not (isinstance(dst, ops.PatchCols) and isinstance(src, ops.PatchCols)):
                    raise ValueError("the step was captured with patch-embed rows (ops.PatchCols) for this input: pass the same kind")
                dst, src = dst.cols, src.cols
            if src is not dst and (src.data_ptr() != dst.data_ptr() or src.dtype != dst.dtype):
                dst.copy_(src, non_blocking=True)
        for dst, src in zip(st["left"] + st["right"], list(left) + list(right)):
            put(dst, src)
        if st["ld"] is not None:
            put(st["ld"], left_designed)
            put(st["rd"], right_designed)
        put(st["flag"], flag)
        self.step_count += 1
        st["hyper"].copy_(ops.adam_hyper(self.step_count, self.lr if lr is None else lr, self.betas[0], self.betas[1]), non_blocking=True)
        if not self.dp and (not self.segmented or self.overlap_adam):
            st["pieces"][0].replay()                     # Adam is inside the one graph (a side branch per bucket with overlap_adam)
            return st["loss"]
        if not self.segmented:                       # one backward graph, then the bucketed exchange of the whole buffer
            st["pieces"][0].replay()
            for bi in range(len(self.bucket_slices)):
                self._launch_bucket(bi)
        else:
            for piece, g in enumerate(st["pieces"]):
                g.replay()
                self._launch_ready(piece)                # the exchange runs next to the graphs replayed after it
        self._finish_buckets(lambda bi: st["adam"][bi].replay())
        return st["loss"]

    def _adam_slice(self, sl: slice, hyper_dev, grad_scale: float):
        lp = self.fp.flat_lp[sl] if self.fp.flat_lp is not None else None
        lo = None
        if self.fp.flat_pair is not None:
            lp, lo = self.fp.flat_pair[0][sl], self.fp.flat_pair[1][sl]
        ops.adam_step_dev(self.fp.flat[sl], self.fp.grad[sl], self.m[sl], self.v[sl], hyper_dev, beta1=self.betas[0], beta2=self.betas[1],
                          eps=self.eps, grad_scale=grad_scale, param_lp=lp, param_lo=lo)

    # -- the step ----------------------------------------------------------------------------------
    def step(self, left: Sequence[torch.Tensor], left_designed, right: Sequence[torch.Tensor], right_designed, flag,
             lr: Optional[float] = None) -> torch.Tensor:
        """forward -> Loss -> zero_grad -> backward -> (all-reduce) -> Adam.  Returns the local loss tensor
        (no host sync; the reference's per-step `.item()` at Train_SMT.py:301 is left to the caller).
        After enable_graph() the same work is replayed from captured hipGraphs (the returned tensor is then a
        static buffer that the next step overwrites)."""
        if self._graph is not None:
            self.net.train()
            return self._graph_step(left, left_designed, right, right_designed, flag, lr)
        return self._eager_step(left, left_designed, right, right_designed, flag, lr)

    def _eager_step(self, left, left_designed, right, right_designed, flag, lr=None) -> torch.Tensor:
        self.net.train()
        self.fp.zero_grad()
        cuts = None
        if self.segmented:
            cuts = _Cuts()
            self.net._dp_cut = cuts
        try:
            fa, fb = self.net(left, left_designed, right, right_designed)
        finally:
            if cuts is not None:
                self.net._dp_cut = None
        loss = self.criterion(fa, fb, flag)
        if cuts is not None and cuts.pairs:
            self.bucket_slices, self._ready_after = self._segment_buckets(cuts)
            for piece in self._backward_segments(loss, cuts):
                self._launch_ready(piece)
        else:
            loss.backward(_unit(loss))
            if self.dp:
                for bi in range(len(self.bucket_slices)):
                    self._launch_bucket(bi)
        self.step_count += 1

        def update(sl):
            if self.fp.flat_pair is not None and self.adam_fn is ops.adam_step:      # "bf16x3": the update also rewrites the weights' plane pairs
                hyper = ops.adam_hyper(self.step_count, self.lr if lr is None else lr, self.betas[0], self.betas[1]).to(self.fp.flat.device, non_blocking=True)
                ops.adam_step_dev(self.fp.flat[sl], self.fp.grad[sl], self.m[sl], self.v[sl], hyper, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                                  grad_scale=1.0 / self.world, param_lp=self.fp.flat_pair[0][sl], param_lo=self.fp.flat_pair[1][sl])
                return
            extra = {"param_lp": self.fp.flat_lp[sl]} if self.fp.flat_lp is not None else {}
            self.adam_fn(self.fp.flat[sl], self.fp.grad[sl], self.m[sl], self.v[sl], self.step_count, lr=self.lr if lr is None else lr,
                         beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, grad_scale=1.0 / self.world, **extra)
        if self.dp:
            self._finish_buckets(lambda bi: update(self.bucket_slices[bi]))
        else:
            self.fp.finish_grads()
            update(slice(0, self.fp.total))
        if self.fp.flat_pair is not None and self.adam_fn is not ops.adam_step:      # a caller-supplied update: re-derive the pairs from the masters
            self.fp.refresh_lp()
        return loss.detach()
