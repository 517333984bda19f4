"""Reward helpers shared by the tasks (torch, batched).

`tolerance` restates dm_control.utils.rewards.tolerance for the only sigmoid the
reference uses ("gaussian", value_at_margin=0.1):
piano_with_shadow_hands.py:261-269,279-298,300-331."""

from __future__ import annotations

import math

import torch

_VALUE_AT_MARGIN = 0.1
_GAUSS_SCALE = math.sqrt(-2.0 * math.log(_VALUE_AT_MARGIN))


def tolerance(x: torch.Tensor, bounds=(0.0, 0.0), margin: float = 0.0) -> torch.Tensor:
    lower, upper = bounds
    if lower > upper:
        raise ValueError("Lower bound must be <= upper bound.")
    if margin < 0:
        raise ValueError("`margin` must be non-negative.")
    in_bounds = (lower <= x) & (x <= upper)
    if margin == 0:
        return in_bounds.to(x.dtype)
    d = torch.where(x < lower, lower - x, x - upper) / margin
    value = torch.exp(-0.5 * (d * _GAUSS_SCALE) ** 2)
    return torch.where(in_bounds, torch.ones_like(x), value)
