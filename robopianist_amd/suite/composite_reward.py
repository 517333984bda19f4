"""Ordered sum of named reward terms (mirror of robopianist/suite/composite_reward.py).
Each term returns a tensor [n_envs]; `reward_terms` keeps the last per-term values."""

from typing import Callable, Dict


class CompositeReward:
    def __init__(self, **kwargs) -> None:
        self._reward_fns: Dict[str, Callable] = {}
        for name, reward_fn in kwargs.items():
            self.add(name, reward_fn)
        self._reward_terms: Dict[str, object] = {}

    def add(self, name: str, reward_fn: Callable) -> None:
        self._reward_fns[name] = reward_fn

    def remove(self, name: str) -> None:
        del self._reward_fns[name]

    def compute(self, physics):
        """Computes the terms in insertion order and returns their sum."""
        total = 0.0
        for name, reward_fn in self._reward_fns.items():
            rew = reward_fn(physics)
            total = total + rew
            self._reward_terms[name] = rew
        return total

    @property
    def reward_fns(self):
        return self._reward_fns

    @property
    def reward_terms(self):
        return self._reward_terms
