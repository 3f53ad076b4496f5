"""Drop-in for the reference's MNIST sandbox nets (`Nets.py`): `MLP` (:11-35) and `FC` (:37-45).

Not on the DeepMerge hot path ("API surface only"); the Linear layers run on the HIP library's generic
fp32 GEMM (their widths 250 / 10 are not MFMA-tileable), the leaky ReLUs are torch's elementwise kernels on the device.
`RNN` (:48-111, a 4-layer bidirectional GRU with debug prints) is not provided.
"""
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
    def __init__(self):
        super().__init__()
        raise NotImplementedError("Nets.RNN (bidirectional GRU MNIST sandbox) is outside the accelerated path")
