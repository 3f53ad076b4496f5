"""MI355X-native drop-in for the reference module `Losses.py`.

`Loss` (the one Train_SMT.py uses, Losses.py:12-38) runs forward and gradient in one HIP kernel
(dm_contrastive_loss).  `MultiLoss` / `ClassLoss` (Losses.py:41-95; unused by the shipped trainer)
combine it with cross-entropy terms, which run forward + gradient in one HIP kernel too
(dm_cross_entropy; class-index or class-probability targets).
"""
from torch import nn

from . import ops


class Loss(nn.Module):
    def __init__(self, margin, lamda, belta):
        super().__init__()
        self.margin, self.lamda, self.belta = margin, lamda, belta   # lamda / belta unused upstream too

    def forward(self, positive, negative, flag, size_average=True):
        return ops.contrastive_loss(positive, negative, flag, float(self.margin))


class MultiLoss(nn.Module):
    def __init__(self, margin, lamda, belta):
        super().__init__()
        self.margin, self.lamda, self.belta = margin, lamda, belta

    def Loss_Class(self, inputs, targets):
        return ops.CrossEntropyFn.apply(inputs, targets)

    def forward(self, positive, negative, flag, left_logits, left_one_hot, right_logits, right_one_hot, size_average=True):
        c = ops.contrastive_loss(positive, negative, flag, float(self.margin))
        return 0.7 * c + 0.15 * self.Loss_Class(left_logits, left_one_hot) + 0.15 * self.Loss_Class(right_logits, right_one_hot)


class ClassLoss(nn.Module):
    def __init__(self, margin, lamda, belta):
        super().__init__()
        self.margin, self.lamda, self.belta = margin, lamda, belta

    def Loss_Class(self, inputs, targets):
        return ops.CrossEntropyFn.apply(inputs, targets)

    def forward(self, left_logits, left_one_hot, right_logits, right_one_hot, size_average=True):
        return 0.5 * self.Loss_Class(left_logits, left_one_hot) + 0.5 * self.Loss_Class(right_logits, right_one_hot)
