"""API-surface counterpart of the reference's `Networks.py` (class name, constructor signature, arity dispatch and
the L2-normalised head only).

Upstream `SpatiallyMmemorizedNetwork` (:17-174) hard-wires `torchvision.models.vgg16(pretrained=True)` (a weight download)
and a `Non_local_block` module that is not in the repository, and its 6- / 9-argument paths use attributes whose
definitions are commented out (`self.dropout`, `self.attention_net`, :139-141): it cannot run as shipped.  Here the
feature extractor is supplied by the caller (`base_net`: any nn.Module mapping images to [B, out_channels, h, w]); the
1x1 reduction conv runs as a GEMM of the HIP library, and `forward_once` keeps the reference's normalisation
x / (||x||_2 + 1e-6) (:100).
"""
import torch
from torch import nn

from . import ops


class SpatiallyMmemorizedNetwork(nn.Module):
    def __init__(self, base_net, pooling, in_channels, out_channels, reduced_size):
        super().__init__()
        if not isinstance(base_net, nn.Module):
            raise NotImplementedError(
                "the reference builds torchvision's pretrained VGG16 + a NONLocalBlock2D that is missing from its repository; "
                "pass the feature extractor as an nn.Module (images -> [B, out_channels, h, w])")
        self.features = nn.Sequential(base_net, nn.AdaptiveAvgPool2d((1, 1)))
        self.reduce_conv = None
        if reduced_size < out_channels:
            print('Feature size reduction: {} -> {}'.format(out_channels, reduced_size))
            self.reduce_conv = nn.Conv2d(out_channels, reduced_size, (1, 1))
        self.pooling = pooling
        self.eps = 1e-6

    def forward_once(self, x):
        x = self.features(x)
        x = x.squeeze(3).squeeze(2)
        if self.reduce_conv is not None:                      # 1x1 conv on a 1x1 map == Linear
            x = ops.LinearFn.apply(x.float().contiguous(), self.reduce_conv.weight, self.reduce_conv.bias, None, torch.float32)
        return x / (torch.norm(x, p=2, dim=1, keepdim=True) + self.eps).expand_as(x)

    def forward_twice(self, x1, x2):
        return self.forward_once(x1), self.forward_once(x2)

    def forward_thrice1(self, x1, x2, x3):
        return self.forward_once(x1), self.forward_once(x2), self.forward_once(x3)

    def forward(self, *args):
        n = len(args)
        if n == 1:
            return self.forward_once(args[0])
        elif n == 2:
            return self.forward_twice(args[0], args[1])
        elif n == 3:
            return self.forward_thrice1(args[0], args[1], args[2])
        elif n in (6, 9):
            raise NotImplementedError("the reference's 6- / 9-argument paths use self.dropout / self.attention_net, which it never defines")
        else:
            raise ValueError('Invalid input arguments! You got {} arguments.'.format(n))
