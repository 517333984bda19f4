"""Minimal dm_env stand-ins (dm_env is not installed): StepType, TimeStep, specs.

Field names and semantics follow dm_env so reference-style code
(`timestep.last()`, `spec.minimum`, ...) reads the same.  Every TimeStep field
is batched over environments (leading dim n_envs)."""

from __future__ import annotations

import enum
from typing import Any, NamedTuple

import numpy as np


class StepType(enum.IntEnum):
    FIRST = 0
    MID = 1
    LAST = 2


class TimeStep(NamedTuple):
    step_type: Any  # int array [E]
    reward: Any  # float array [E] or None (all-FIRST)
    discount: Any  # float array [E] or None (all-FIRST)
    observation: Any  # dict name -> array [E, ...]

    def first(self):
        return self.step_type == StepType.FIRST

    def mid(self):
        return self.step_type == StepType.MID

    def last(self):
        return self.step_type == StepType.LAST


class Array:
    def __init__(self, shape, dtype, name=None):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.name = name

    def validate(self, value):
        value = np.asarray(value)
        if value.shape[-len(self.shape):] != self.shape and self.shape != ():
            raise ValueError(f"Expected shape (..., {self.shape}) but found {value.shape}")
        return value

    def __repr__(self):
        return f"Array(shape={self.shape}, dtype={self.dtype}, name={self.name!r})"


class BoundedArray(Array):
    def __init__(self, shape, dtype, minimum, maximum, name=None):
        super().__init__(shape, dtype, name)
        self.minimum = np.broadcast_to(np.asarray(minimum, dtype), self.shape).copy()
        self.maximum = np.broadcast_to(np.asarray(maximum, dtype), self.shape).copy()
        if np.any(self.minimum > self.maximum):
            raise ValueError("All values in `minimum` must be <= `maximum`.")

    def validate(self, value):
        value = super().validate(value)
        if (value < self.minimum).any() or (value > self.maximum).any():
            raise ValueError("Values out of bounds.")
        return value

    def __repr__(self):
        return (f"BoundedArray(shape={self.shape}, dtype={self.dtype}, name={self.name!r}, "
                f"minimum={self.minimum}, maximum={self.maximum})")


def merge_specs(spec_list):
    """mujoco_utils.spec_utils.merge_specs: concatenates 1-D bounded specs."""
    dtype = spec_list[0].dtype
    n = sum(s.shape[0] for s in spec_list)
    mins = np.concatenate([s.minimum for s in spec_list])
    maxs = np.concatenate([s.maximum for s in spec_list])
    name = "\t".join(s.name for s in spec_list if s.name)
    return BoundedArray((n,), dtype, mins, maxs, name=name)
