"""Relative-geometry helpers with the reference's names (utils/geometry.py:19-23, 47-51)."""
import math

import torch


def angle_between_2d_vectors(ctr_vector: torch.Tensor, nbr_vector: torch.Tensor) -> torch.Tensor:
    """Signed angle from ``ctr_vector`` to ``nbr_vector``: atan2(cross, dot).

    The dot product is a *reduction* (reference :23), i.e. it starts from +0: for a zero ``nbr_vector`` (first
    motion vector of a window, a centre-line point on its polygon origin, the current step seen from itself)
    the products can be -0 and ``(-0) + (-0) = -0`` would turn atan2(0, .) from 0 into pi; ``0 + (-0) + (-0)``
    is +0.  Keep the reduction form."""
    cross = ctr_vector[..., 0] * nbr_vector[..., 1] - ctr_vector[..., 1] * nbr_vector[..., 0]
    dot = (ctr_vector[..., :2] * nbr_vector[..., :2]).sum(dim=-1)
    return torch.atan2(cross, dot)


def wrap_angle(angle: torch.Tensor, min_val: float = -math.pi, max_val: float = math.pi) -> torch.Tensor:
    """Floor-mod wrap into [min_val, max_val)."""
    return torch.remainder(angle + max_val, max_val - min_val) + min_val
