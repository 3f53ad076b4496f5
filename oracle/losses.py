"""Oracle (test infrastructure): contrastive / auxiliary losses of the reference.

Reference: Losses.py:12-38 (`Loss`), :41-69 (`MultiLoss`), :72-95 (`ClassLoss`).
"""
import torch
import torch.nn.functional as F


def contrastive_loss(positive: torch.Tensor, negative: torch.Tensor, flag: torch.Tensor, margin: float) -> torch.Tensor:
    """Losses.py:34-38.  d = squared L2 distance (no sqrt); flag==1 pulls together,
    flag==0 pushes beyond `margin`; mean over the batch.  `lamda`/`belta` are unused upstream."""
    d = (positive - negative).pow(2).sum(1)
    losses = flag * d + (1 - flag) * F.relu(margin - d)
    return losses.mean()


def multi_loss(positive, negative, flag, left_logits, left_target, right_logits, right_target, margin: float):
    """Losses.py:58-69: 0.7*contrastive + 0.15*CE(left) + 0.15*CE(right)."""
    c = contrastive_loss(positive, negative, flag, margin)
    return 0.7 * c + 0.15 * F.cross_entropy(left_logits, left_target) + 0.15 * F.cross_entropy(right_logits, right_target)


def class_loss(left_logits, left_target, right_logits, right_target):
    """Losses.py:87-95: 0.5*CE(left) + 0.5*CE(right)."""
    return 0.5 * F.cross_entropy(left_logits, left_target) + 0.5 * F.cross_entropy(right_logits, right_target)
