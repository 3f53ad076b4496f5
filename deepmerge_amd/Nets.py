"""Drop-in for the reference's MNIST sandbox nets (`Nets.py`): `MLP` (:11-35), `FC` (:37-45) and `RNN` (:48-111).

Not on the DeepMerge hot path ("API surface only"); the Linear layers run on the HIP library's generic
fp32 GEMM (their widths 250 / 10 are not MFMA-tileable), the leaky ReLUs are torch's elementwise kernels on the device.
`RNN` (a 4-layer bidirectional GRU + a self-attention pooling, round 2): input projections of all time steps as one GEMM per
layer and direction, one small GEMM + one `dm_gru_cell` launch per step, the attention pooling on the generic attention kernels
(one head of 160).  fp32 throughout; the reference's debug prints are not reproduced.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import ops


def _linear(layer: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    return ops.LinearFn.apply(x.float().contiguous(), layer.weight, layer.bias, None, torch.float32)


class MLP(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(784, 250)
        self.fc2 = nn.Linear(250, 250)
        self.fc3 = nn.Linear(250, 10)

    def forward(self, x):
        fc1_map = F.leaky_relu(_linear(self.fc1, x))
        fc2_map = F.leaky_relu(_linear(self.fc2, fc1_map))
        fc3_map = F.leaky_relu(_linear(self.fc3, fc2_map))
        return fc3_map, fc2_map


class FC(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc3 = nn.Linear(250, 10)

    def forward(self, x):
        return F.leaky_relu(_linear(self.fc3, x))


class RNN(nn.Module):
    """Nets.py:48-111.  `self.rnn` (torch.nn.GRU) is the parameter container: same state_dict keys as the reference
    (`rnn.weight_ih_l0`, `rnn.weight_hh_l0_reverse`, ..., `out.weight`)."""

    def __init__(self):
        super().__init__()
        self.rnn = nn.GRU(input_size=28, hidden_size=80, num_layers=4, batch_first=True, bidirectional=True)
        self.dropout = nn.Dropout(0.5)
        self.out = nn.Linear(160, 10)

    def _direction(self, x, layer, reverse):
        """x [B, T, in] -> h_t for every step, [B, T, 80] (time order of the input, as torch.nn.GRU returns it)."""
        sfx = f"_l{layer}" + ("_reverse" if reverse else "")
        w_ih, w_hh = getattr(self.rnn, "weight_ih" + sfx), getattr(self.rnn, "weight_hh" + sfx)
        b_ih, b_hh = getattr(self.rnn, "bias_ih" + sfx), getattr(self.rnn, "bias_hh" + sfx)
        B, T, _ = x.shape
        H = w_hh.shape[1]
        gi = ops.LinearFn.apply(x.reshape(B * T, -1), w_ih, b_ih, None, torch.float32).view(B, T, 3 * H)
        h = torch.zeros((B, H), dtype=torch.float32, device=x.device)
        outs = [None] * T
        for t in (range(T - 1, -1, -1) if reverse else range(T)):
            gh = ops.LinearFn.apply(h, w_hh, b_hh, None, torch.float32)
            h = ops.GRUCellFn.apply(gi[:, t], gh, h)
            outs[t] = h
        return torch.stack(outs, dim=1)

    def attention_net(self, x, query, mask=None):
        """softmax(query x^T / sqrt(d)) x summed over the sequence (Nets.py:72-90); returns (context, None): the reference's second
        value (the [B, T, T] weights) is not materialised by the fused kernel and its forward() discards it."""
        B, T, D = x.shape
        qkv = torch.stack((query, x, x), dim=2).view(B, T, 3, 1, D)
        ctx = ops.AttentionFn.apply(qkv, None, None, B, T, 1, D, 1.0 / math.sqrt(D))
        return ctx.view(B, T, D).sum(1), None

    def forward(self, x):
        x = x.float().contiguous()
        for layer in range(self.rnn.num_layers):
            x = torch.cat((self._direction(x, layer, False), self._direction(x, layer, True)), dim=2)
        query = self.dropout(x)
        attn_output, _ = self.attention_net(x, query)
        return _linear(self.out, attn_output)
