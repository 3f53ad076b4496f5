"""Oracle (test infrastructure): the Adam update the reference's trainer applies.

Reference: Train_SMT.py:192-193 -- torch.optim.Adam(params, lr=1e-4) with library defaults
(betas 0.9/0.999, eps 1e-8, weight_decay 0, amsgrad False), stepped once per batch (:298-300).
Restated from the published algorithm (Kingma & Ba) in the exact operation order of
torch.optim.Adam's single-tensor path; tests/golden pins it against torch.optim.Adam itself.
"""
import math

import torch


def adam_step(param: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int,
              lr: float = 1e-4, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> None:
    """In-place update of (param, m, v) for 1-based `step`."""
    m.mul_(beta1).add_(grad, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(grad, grad, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-(lr / bc1))


def multistep_lr(base_lr: float, epoch: int, milestones=(40, 80), gamma: float = 0.2) -> float:
    """MultiStepLR(milestones=[40,80], gamma=0.2) stepped once per epoch (Train_SMT.py:194, :351)."""
    return base_lr * gamma ** sum(1 for m in milestones if epoch >= m)
