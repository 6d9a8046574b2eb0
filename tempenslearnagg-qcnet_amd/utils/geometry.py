"""Relative-geometry helpers with the reference's names (utils/geometry.py:19-23, 47-51)."""
import math

import torch


def angle_between_2d_vectors(ctr_vector: torch.Tensor, nbr_vector: torch.Tensor) -> torch.Tensor:
    """Signed angle from ``ctr_vector`` to ``nbr_vector``: atan2(cross, dot)."""
    cx, cy = ctr_vector[..., 0], ctr_vector[..., 1]
    nx, ny = nbr_vector[..., 0], nbr_vector[..., 1]
    return torch.atan2(cx * ny - cy * nx, cx * nx + cy * ny)


def wrap_angle(angle: torch.Tensor, min_val: float = -math.pi, max_val: float = math.pi) -> torch.Tensor:
    """Floor-mod wrap into [min_val, max_val)."""
    return torch.remainder(angle + max_val, max_val - min_val) + min_val
