"""Algorithmic work of the pair-encoder step (the figures bench.py and DESIGN.md price against)."""
from typing import Sequence


def encoder_forward_flops(scales: Sequence[int], in_c: int, depth: Sequence[int], grid: int = 8, dim: int = 768,
                          n_designed: int = 19, out_dim: int = 100) -> float:
    """Forward FLOPs per encoder sample (SURVEY 8d / BASELINE.md section 3):
    sum_scales 2*64*(in_c*p^2)*C + sum_stages depth*(24*N*C^2 + 4*N^2*C) + 2*(19*C + 2*C^2) + 2*(S+1)*C*100."""
    S, C = len(scales), dim
    f = sum(2.0 * grid * grid * (in_c * int(s / grid) ** 2) * C for s in scales)
    for stage in range(3):
        n = S * (grid >> stage) ** 2
        f += depth[stage] * (24.0 * n * C * C + 4.0 * n * n * C)
    f += 2.0 * (n_designed * C + 2 * C * C)
    f += 2.0 * (S + 1) * C * out_dim
    return f


def pair_step_flops(scales, in_c, depth) -> float:
    """2 sides x 3 (forward + dgrad + wgrad) x forward FLOPs."""
    return 6.0 * encoder_forward_flops(scales, in_c, depth)
