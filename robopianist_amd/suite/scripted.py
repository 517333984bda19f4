"""Scripted action replay (the reference's example loop as an action source)."""

from __future__ import annotations


class ScriptedActions:
    """An action source for `Environment.step`: a recorded action table [T, n_action] replayed per env, env e taking row
    `index[e]` (int64, device) -- what the reference's example does on the host, one env at a time
    (/root/reference/examples/piano_with_shadow_hands_env.py:110-141: `for t: env.step(actions[t])`).  After every step
    the index is 0 for an env whose step returned FIRST (a reset consumes no row) and min(index + 1, T - 1) otherwise.
    On the HIP task path the pre-step launch reads the table and advances the index (no gather or index arithmetic
    between two steps); anywhere else `take` / `advance` do the same with torch ops."""

    def __init__(self, table, index):
        self.table, self.index = table, index

    def take(self):
        return self.table.index_select(0, self.index.clamp(0, int(self.table.shape[0]) - 1))

    def advance(self, first):
        self.index.add_(1).clamp_(max=int(self.table.shape[0]) - 1).masked_fill_(first, 0)
