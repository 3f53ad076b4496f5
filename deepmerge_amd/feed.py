"""Sync-free training feed: a pair batch goes from uint8 tiles resident in HBM and a DEVICE sample table straight into the inputs of
`PairTrainer.step` -- the MI355X counterpart of the batch assembly the reference runs on the host for every step
(Train_SMT.py:212-262 iterating DataLoaders whose items come from MyUtils1.py:41-77: `get_scales` :130-156, the crop / zero pad /
INTER_AREA resize / 255 chain :116-223, the designed-feature row :60-77).

Per step: ONE gather launch per scale over all 2B samples of the batch (both sides, any mix of tiles: the tile id is a column of the
table), window arithmetic and the designed-feature rows inside that launch, results written directly into the static buffers the
captured step reads (`PairTrainer.graph_inputs_both()`), by default as the patch-embed GEMM's bf16 operand rows (`ops.PatchCols`) so
that neither the fp32 patch tensors nor the im2col pass exist in the training step.  Nothing is read back: no `.cpu()`, `.item()`,
`nonzero` or host-side maximum; an out-of-range sample raises a device flag that `check()` reads when the caller reads the loss.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Union

import torch

from . import ops


@dataclass
class PairTable:
    """The samples of one pair batch, on the device.  Rows 0 .. B-1 are the left sides, rows B .. 2B-1 the right sides of pairs
    0 .. B-1 (the [left; right] stacking `forward_pair_batched` consumes)."""
    tile_id: torch.Tensor       # int32 [2B]      tile each sample is cut from
    xy: torch.Tensor            # int32 [2B, 2]   pixel coordinates of the sample point (MyUtils1.py:67-73 after geo -> pixel)
    inner: torch.Tensor         # int32 [2B]      inner window side (MyUtils1.py:130-156)
    obj: torch.Tensor           # int32 [2B]      object window side
    region: torch.Tensor        # float32 [2B, 15] region attributes (MyUtils1.py:79-114)
    flag: torch.Tensor          # [B]             1 = same object (merge), 0 = different

    @staticmethod
    def stack(left: "PairTable", right: "PairTable") -> "PairTable":
        return PairTable(*(torch.cat((getattr(left, f), getattr(right, f)), 0).contiguous() for f in ("tile_id", "xy", "inner", "obj", "region")),
                         flag=left.flag)


class PairFeed:
    def __init__(self, tiles: torch.Tensor, scales: Sequence[int], pairs: int, max_window: Union[int, Sequence[int]], grid: int = 8,
                 rows: bool = True, numerics: str = "bf16", resize: str = "opencv", trainer=None):
        """tiles: uint8 [T, bands, H, W] on the GPU.  scales: the model's input scales (up to 4: window (inner, obj, obj + interval,
        obj + 2 interval)[i] is resized to scales[i]).  max_window: an upper bound of the window sides the table will hold (one
        number, or one per scale -- it sizes the gather's LDS staging, <= 384; a sample beyond it raises the error flag).
        rows=True: patch-embed operand rows in the activation dtype of `numerics`; False: float32 [2B, bands, s, s] patches.
        trainer: once its step is captured, the feed writes straight into the graph's input buffers."""
        if tiles.dtype != torch.uint8 or tiles.dim() != 4 or not tiles.is_cuda:
            raise ValueError("tiles must be a uint8 [T, bands, H, W] tensor on the GPU")
        if not 1 <= len(scales) <= 4:
            raise ValueError("1..4 scales (the reference's get_scales derives four window sides)")
        self.tiles = tiles.contiguous()
        self.scales, self.pairs, self.grid, self.rows, self.resize = [int(s) for s in scales], int(pairs), int(grid), bool(rows), resize
        self.max_window = [int(max_window)] * len(scales) if isinstance(max_window, int) else [int(m) for m in max_window]
        if len(self.max_window) != len(self.scales):
            raise ValueError("max_window: one bound, or one per scale")
        dev, bands, P = tiles.device, tiles.shape[1], 2 * self.pairs
        dtype = ops.act_dtype(numerics) if rows else torch.float32
        self.both: List = []
        for s in self.scales:
            if rows:
                ps = s // grid
                self.both.append(ops.PatchCols(torch.empty((P * grid * grid, bands * ps * ps), dtype=dtype, device=dev), P, s, ps, bands))
            else:
                self.both.append(torch.empty((P, bands, s, s), dtype=torch.float32, device=dev))
        self.dboth = torch.empty((P, 1, 19), dtype=torch.float32, device=dev)
        self.flag = torch.empty((self.pairs,), dtype=torch.float32, device=dev)
        self.error_flag = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.trainer, self._bound = trainer, False

    def _bind(self):
        """After the trainer captured its step: adopt the graph's static inputs as this feed's output buffers."""
        if self.trainer is None or self._bound:
            return
        gi = self.trainer.graph_inputs_both()
        if gi is None:
            return
        both, dboth, flag = gi
        same = all(type(a) is type(b) and tuple(a.shape) == tuple(b.shape) for a, b in zip(both, self.both))
        if not same or dboth is None or len(both) != len(self.both):
            raise RuntimeError("the captured step's inputs do not match this feed (scales, rows / patches, batch size)")
        self.both, self.dboth, self.flag, self._bound = list(both), dboth, flag, True

    def fill(self, table: PairTable):
        """Enqueue the gathers for `table`; returns the arguments of PairTrainer.step: (left, left_designed, right, right_designed, flag)."""
        self._bind()
        B = self.pairs
        if table.xy.shape[0] != 2 * B:
            raise ValueError(f"the table holds {table.xy.shape[0]} samples, this feed was built for {2 * B} (2 x {B} pairs)")
        for i, s in enumerate(self.scales):
            out = self.both[i].cols if self.rows else self.both[i]
            first = i == 0
            ops.pair_batch_gather(self.tiles, table.tile_id, table.xy, table.inner, table.obj, i, s, self.max_window[i], out,
                                  grid=self.grid if self.rows else 0, resize=self.resize, region_features=table.region if first else None,
                                  designed=self.dboth if first else None, error_flag=self.error_flag)
        self.flag.copy_(table.flag, non_blocking=True)
        left, right = [t[:B] for t in self.both], [t[B:] for t in self.both]
        return left, self.dboth[:B], right, self.dboth[B:], self.flag

    def check(self):
        """Host read of the error flag (synchronises: call it where the loss is read, every k steps).  Raises if any sample since the
        last check had a tile id or window side out of range -- those samples were fed as zeros."""
        if int(self.error_flag.item()) != 0:
            self.error_flag.zero_()
            raise ValueError("PairFeed: a sample's tile id or window side was out of range (tile_id outside [0, T), or a window side "
                             "outside 1..max_window); it was fed as zeros")
