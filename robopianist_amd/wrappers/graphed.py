"""GraphedStepWrapper: replays a whole env.step (action scaling, task hooks, the
1 + 2 n_substeps engine launches, observations, rewards, termination) from one captured
hipGraph.

An env step is ~250 small launches (21 engine kernels + ~230 elementwise torch kernels of
the task layer); enqueueing them one by one costs more host time than some of them run.
Everything in the step path is enqueue-only and updates its state in place (see
suite/environment.py, suite/tasks/*), so the sequence is captured once with
`torch.cuda.graph` (hipGraph on ROCm) and then replayed.

Semantics: identical to the wrapped env, with two documented differences --
  * the returned TimeStep tensors are static buffers that the next `step` overwrites
    (clone what you keep);
  * the first `warmup_steps` calls run eagerly (PyTorch needs every kernel loaded before a
    capture), the capture happens on the next call.
Not capturable (raises): n_envs == 1 (host-side dm_env reset rule), hand-position
randomisation and MIDI augmentations (host work per episode), key-trace recording to a host
buffer.
"""

from __future__ import annotations

import torch


class GraphedStepWrapper:
    def __init__(self, environment, warmup_steps: int = 2):
        self._environment = environment
        self._eager_left = int(warmup_steps)
        self._graph = None
        self._static_action = None
        self._static_ts = None
        base = environment
        while hasattr(base, "_environment"):
            base = base._environment
        self._base = base
        if base.n_envs == 1:
            raise ValueError("GraphedStepWrapper needs n_envs > 1 (the single-env reset rule reads the host)")
        if getattr(base.task, "_randomize_hand_positions", False):
            raise ValueError("hand-position randomisation draws from the host RNG and cannot be captured")
        if getattr(base.task, "needs_host_episode_setup", False):
            raise ValueError("MIDI augmentations regenerate goal tables on the host at episode starts "
                             "and cannot be captured")

    def __getattr__(self, name):
        return getattr(self._environment, name)

    @property
    def graph_captured(self) -> bool:
        return self._graph is not None

    def reset(self):
        return self._environment.reset()

    def _capture(self, action):
        phys = self._base.physics
        dev = phys.device
        self._static_action = torch.as_tensor(action, device=dev, dtype=phys.dtype).clone()
        graph = torch.cuda.CUDAGraph()
        eager_stream = phys._stream
        torch.cuda.synchronize(dev)
        try:
            with torch.cuda.graph(graph):
                # engine launches must land on the capturing stream
                phys.engine.set_stream(torch.cuda.current_stream(dev).cuda_stream)
                ts = self._environment.step(self._static_action)
        finally:
            phys.engine.set_stream(eager_stream.cuda_stream)
        self._graph, self._static_ts = graph, ts

    def step(self, action):
        if self._eager_left > 0:
            self._eager_left -= 1
            return self._environment.step(action)
        if self._graph is None:
            self._capture(action)  # records only; the replay below executes this step
        self._static_action.copy_(torch.as_tensor(action, device=self._static_action.device,
                                                  dtype=self._static_action.dtype))
        self._graph.replay()
        return self._static_ts

    def step_eager(self, action):
        """One step through the wrapped env without the graph (same state)."""
        return self._environment.step(action)
