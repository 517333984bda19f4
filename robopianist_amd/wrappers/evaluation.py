"""MidiEvaluationWrapper: per-episode precision / recall / F1 of key presses and of the
sustain pedal against the MIDI goal (mirror of robopianist/wrappers/evaluation.py:40-177),
computed as batched device reductions instead of sklearn calls.  On the HIP task layer the
reduction runs inside the `rp_task_advance` launch (include/rp_task.h, `eval_*` buffers); the
torch code below is the definition and the fallback.

Per step and env: precision/recall/F1 of activation vs the goal row of that step with
sklearn's `average="binary", zero_division=1` convention, averaged over the episode."""

from __future__ import annotations

from collections import deque
from typing import Dict

import torch


def _prf(y_true: torch.Tensor, y_pred: torch.Tensor):
    """Binary precision/recall/F1 along the last dim, zero_division=1 (sklearn)."""
    tp = (y_true & y_pred).sum(-1).double()
    fp = (~y_true & y_pred).sum(-1).double()
    fn = (y_true & ~y_pred).sum(-1).double()
    one = torch.ones_like(tp)
    precision = torch.where(tp + fp > 0, tp / torch.clamp(tp + fp, min=1), one)
    recall = torch.where(tp + fn > 0, tp / torch.clamp(tp + fn, min=1), one)
    denom = precision + recall
    f1 = torch.where(denom > 0, 2 * precision * recall / torch.clamp(denom, min=1e-300), torch.zeros_like(tp))
    # sklearn: if there are no positives at all (tp+fp+fn == 0) f-score is zero_division=1
    f1 = torch.where(tp + fp + fn == 0, one, f1)
    return precision, recall, f1


class MidiEvaluationWrapper:
    def __init__(self, environment, deque_size: int = 1) -> None:
        self._environment = environment
        E = environment.n_envs
        dev = environment.physics.device
        self._sums = torch.zeros((E, 6), dtype=torch.float64, device=dev)
        self._count = torch.zeros(E, dtype=torch.float64, device=dev)
        # per env: ring of the last `deque_size` finished episodes (device side, no syncs)
        self._hist = torch.zeros((E, deque_size, 6), dtype=torch.float64, device=dev)
        self._n_finished = torch.zeros(E, dtype=torch.long, device=dev)
        self._deque_size = deque_size
        self._fused_with = None  # the FusedAdvance object our buffers are registered with
        self._use_fused = True

    def __getattr__(self, name):
        return getattr(self._environment, name)

    def reset(self):
        self._sums.zero_()
        self._count.zero_()
        return self._environment.reset()

    def _fused(self) -> bool:
        """True when the task's fused launch does the reduction (our buffers registered)."""
        task, phys = self._environment.task, self._environment.physics
        if not self._use_fused or not hasattr(task, "set_evaluation_buffers"):
            return False
        if task.fused_advance_for(phys) is None:
            if self._fused_with is not None:
                task.set_evaluation_buffers(None)
                self._fused_with = None
            return False
        if self._fused_with is None:
            task.set_evaluation_buffers((self._sums, self._count, self._hist, self._n_finished))
            self._fused_with = task
        return True

    def step(self, action):
        if self._fused():
            return self._environment.step(action)
        task = self._environment.task
        # goal row of the step about to be simulated = goal_state[:, 0]
        goal = task._goal_state[:, 0].clone()
        timestep = self._environment.step(action)
        stepped = ~timestep.first()
        keys_true = goal[:, :-1] > 0
        keys_pred = task.piano.activation
        sus_true = goal[:, -1:] > 0
        sus_pred = task.piano.sustain_activation
        vals = torch.stack(_prf(keys_true, keys_pred) + _prf(sus_true, sus_pred), dim=-1)
        self._sums += vals * stepped[:, None]
        self._count += stepped.double()
        last = timestep.last()
        mean = self._sums / torch.clamp(self._count, min=1)[:, None]
        slot = (self._n_finished % self._deque_size)
        cur = self._hist[torch.arange(self._hist.shape[0], device=slot.device), slot]
        self._hist[torch.arange(self._hist.shape[0], device=slot.device), slot] = torch.where(
            last[:, None], mean, cur)
        self._n_finished += last.long()
        self._sums.copy_(torch.where(last[:, None], torch.zeros_like(self._sums), self._sums))
        self._count.copy_(torch.where(last, torch.zeros_like(self._count), self._count))
        return timestep

    def get_musical_metrics(self) -> Dict[str, float]:
        """Mean over the last `deque_size` finished episodes of every env."""
        n = torch.clamp(self._n_finished, max=self._deque_size)
        if int(n.sum()) == 0:
            raise ValueError("No episode metrics available yet.")
        valid = torch.arange(self._deque_size, device=n.device)[None, :] < n[:, None]
        allv = (self._hist * valid[..., None]).sum((0, 1)) / valid.sum()
        names = ["precision", "recall", "f1", "sustain_precision", "sustain_recall", "sustain_f1"]
        return {k: float(v) for k, v in zip(names, allv)}
