"""Vectorised dm_env-style Environment.

Reproduces the hook order of dm_control's composer.Environment.step as the
reference relies on it (in-tree corroboration at
piano_with_shadow_hands.py:372-376 and self_actuated_piano.py:160-167):

    before_step -> n_sub x physics.step (+ after_substep) -> after_step
    -> observation update -> reward -> discount / termination

for all envs at once.  dm_env's protocol is kept per env: the first step()
after a LAST (or before any reset) resets that env and returns FIRST without
simulating it (the engine's RP_ACTIVE mask).
"""

from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from robopianist_amd.suite import specs
from robopianist_amd.suite.specs import StepType, TimeStep
from robopianist_amd.suite.scripted import ScriptedActions


class Environment:
    def __init__(self, task, n_envs: int = 1, random_state=None, device_id: int = 0,
                 precision: int = 64, physics=None, record_key_trace: bool = False,
                 copy_outputs: bool = True, legacy_step: bool = True):
        self._task = task
        self._n_envs = int(n_envs)
        if isinstance(random_state, np.random.RandomState):
            self._random_state = random_state
        else:
            self._random_state = np.random.RandomState(random_state)
        if physics is None:
            from robopianist_amd.suite.physics import TorchPhysics
            physics = TorchPhysics(task.scene, self._n_envs, device_id=device_id, precision=precision)
        self._physics = physics
        # composer.Environment(legacy_step=...) (robopianist/suite/__init__.py:91): False = every physics.step() is
        # mj_step (mj_step1; mj_step2) -- same state trajectory; sites, contacts and velocity sensors are then the
        # ones of the state before the last integration (include/rp_engine.h: rp_set_legacy_step)
        self._legacy_step = bool(legacy_step)
        if not self._legacy_step:
            if not hasattr(getattr(physics, "engine", None), "set_legacy_step"):
                raise ValueError("legacy_step=False needs the HIP engine (this physics object has no step-order switch)")
            physics.engine.set_legacy_step(False)
        self._n_sub_steps = task.physics_steps_per_control_step
        task.bind(physics, self._n_envs, self._random_state)
        self._needs_reset = torch.ones(self._n_envs, dtype=torch.bool, device=physics.device)
        self._record_key_trace = record_key_trace
        self._copy_outputs = bool(copy_outputs)
        self._key_trace = None
        if record_key_trace:
            self._key_trace = torch.zeros((self._n_envs, self._n_sub_steps, 4), dtype=torch.int32,
                                          device=physics.device)

    # -- accessors ---------------------------------------------------------------
    @property
    def task(self):
        return self._task

    @property
    def physics(self):
        return self._physics

    @property
    def random_state(self):
        return self._random_state

    @property
    def n_envs(self):
        return self._n_envs

    @property
    def key_trace(self):
        """[n_envs, n_substeps, 4] int32 activation bit masks of the last step."""
        return self._key_trace

    def control_timestep(self):
        return self._task.control_timestep

    def action_spec(self):
        return self._task.action_spec(self._physics)

    def observation_spec(self):
        return self._task.observation_spec()

    # -- protocol -----------------------------------------------------------------
    def reset(self) -> TimeStep:
        phys = self._physics
        phys.set_active(torch.ones(self._n_envs, dtype=torch.bool, device=phys.device))
        phys.reset(None)
        self._task.initialize_episode(phys, None)
        phys.forward()  # physics.forward() after initialize_episode, as composer does
        self._task.piano._update_key_state(phys)
        self._needs_reset.zero_()
        obs = self._task.get_observation(self._physics)
        st = torch.full((self._n_envs,), int(StepType.FIRST), dtype=torch.int32,
                        device=self._physics.device)
        # (own copies, like step(): a FIRST observation kept by the caller must survive the next step)
        return self._fresh(TimeStep(st, None, None, obs))

    def request_reset(self, mask) -> None:
        """Extension: flags the envs selected by the boolean device tensor `mask` for a reset at
        the next step() (they return FIRST there and are not simulated), exactly as if their episode
        had just ended.  Device-side, no synchronisation."""
        self._needs_reset.logical_or_(mask.to(device=self._needs_reset.device, dtype=torch.bool))

    def step_canonical(self, action, bounds, clip=False) -> TimeStep:
        """step() for an action in [-1, 1] with the mapping onto the spec's bounds (`bounds` = (lo, hi - lo) device
        tensors) folded into the pre-step launch: what CanonicalSpecWrapper.step does in front of step()."""
        return self.step(action, _canonical=(bounds, clip))

    def step(self, action, _canonical=None) -> TimeStep:
        """Fully asynchronous for n_envs > 1: resets are applied through device-side
        masks, nothing is read back to the host.  On the HIP engine with the standard task set an env step is
        three C calls: rp_task_prestep, rp_step_masked, rp_task_advance."""
        phys, task = self._physics, self._task
        resetting = self._needs_reset
        if self._n_envs == 1 and bool(resetting.all()):
            if isinstance(action, ScriptedActions):
                action.index.zero_()
            return self.reset()  # dm_env: step after LAST == reset (reward None)
        fused = task.fused_advance_for(phys) if hasattr(task, "fused_advance_for") else None
        pre = task.fused_prestep_for(phys) if (fused is not None and hasattr(task, "fused_prestep_for")) else None
        scripted = action if isinstance(action, ScriptedActions) else None
        if pre is not None and not getattr(task, "needs_host_episode_setup", False):
            if scripted is None:
                action = torch.as_tensor(action, device=phys.device, dtype=phys.dtype).reshape(self._n_envs, -1)
                if not action.is_contiguous():
                    action = action.contiguous()
            mask = pre.run(action, resetting, *(_canonical if _canonical is not None else (None, False)))
            phys.step_masked(self._n_sub_steps, self._key_trace, mask)
            st, reward, discount, obs = task.fused_advance(phys, self._needs_reset)
            return self._fresh(TimeStep(st, reward, discount, obs))
        if scripted is not None:
            # (torch paths: the table's rows gathered here, the index advanced with the step types of THIS step)
            action = scripted.take()
            scripted.advance(resetting)
        if _canonical is not None:
            (lo, rng), clip = _canonical
            action = torch.as_tensor(action, device=phys.device, dtype=phys.dtype)
            if clip:
                action = torch.clamp(action, -1.0, 1.0)
            action = lo + (action + 1.0) * 0.5 * rng
        active = ~resetting
        # the action of an env that is being reset is discarded (dm_env): it must not leak into ctrl /
        # the sustain latch, which the FIRST observation reports as 0 after reset()
        action = torch.as_tensor(action, device=phys.device, dtype=phys.dtype).reshape(self._n_envs, -1)
        action = action.masked_fill(resetting[:, None], 0.0)   # (a NaN times zero would survive a product)
        if fused is not None:
            # HIP task layer (include/rp_task.h): the episode reset of the flagged envs, the
            # key state, after_step, observables, rewards, termination and the step types
            # are one launch after the physics
            if getattr(task, "needs_host_episode_setup", False):
                task.prepare_episodes(phys, resetting)  # MIDI augmentations (host; one read-back)
                fused = task.fused_advance_for(phys)    # (the bank may have been reallocated)
            phys.reset(resetting)
            phys.set_active(resetting)
            phys.forward()
            phys.set_active(active)
            task.before_step(phys, action)
            phys.step(self._n_sub_steps, self._key_trace)
            st, reward, discount, obs = task.fused_advance(phys, self._needs_reset)
            return self._fresh(TimeStep(st, reward, discount, obs))
        # envs that finished (or were never reset) start a new episode and are not simulated
        phys.reset(resetting)
        task.initialize_episode(phys, resetting)
        phys.set_active(resetting)
        phys.forward()
        phys.set_active(active)
        task.before_step(phys, action)
        phys.step(self._n_sub_steps, self._key_trace)
        task.after_substeps(phys)
        task.after_step(phys, active)
        obs = task.get_observation(phys)
        reward = task.get_reward(phys)
        # composer.Environment.step reads the discount BEFORE should_terminate_episode: the LAST
        # TimeStep of a wrong-press termination carries discount 1, only task._discount drops to 0
        # (which is what the reference's test asserts, piano_with_shadow_hands_test.py:228-242)
        discount = task.get_discount(phys).clone()
        terminate = task.should_terminate_episode(phys) & active
        # physics divergence terminates the episode (dm_control PhysicsError semantics); so does a
        # capacity overflow of the engine (dropped contacts / cross terms = wrong physics from here on)
        bad = (phys.warn & int(getattr(task, "fatal_warn_mask", 1))).bool() & active
        if hasattr(task, "count_fatal"):
            task.count_fatal(phys.warn, active)
        terminate = terminate | bad
        if hasattr(task, "count_overflow"):
            task.count_overflow(phys.warn, terminate)
        reward = torch.where(bad, torch.zeros_like(reward), reward)
        discount = torch.where(bad, torch.zeros_like(discount), discount)
        st = torch.where(terminate, int(StepType.LAST), int(StepType.MID)).to(torch.int32)
        st = torch.where(resetting, torch.full_like(st, int(StepType.FIRST)), st)
        reward = torch.where(resetting, torch.zeros_like(reward), reward)
        discount = torch.where(resetting, torch.ones_like(discount), discount)
        self._needs_reset.copy_(terminate)  # in place: the step is hipGraph-capturable
        return self._fresh(TimeStep(st, reward, discount, obs))

    # -- checkpoint / resume -----------------------------------------------------------
    def state_dict(self):
        """Snapshot from which `load_state_dict` continues the rollout bit for bit: engine
        state, task / piano episode state, the pending-reset flags and the host RandomState."""
        return {"physics": self._physics.state_dict(), "task": self._task.state_dict(),
                "needs_reset": self._needs_reset.detach().clone(),
                "random_state": self._random_state.get_state()}

    def load_state_dict(self, sd):
        self._task.load_state_dict(sd["task"])
        self._physics.load_state_dict(sd["physics"])
        self._needs_reset.copy_(sd["needs_reset"].to(self._needs_reset.device))
        self._random_state.set_state(sd["random_state"])

    def _fresh(self, ts: TimeStep) -> TimeStep:
        """dm_env hands out arrays the caller may keep; the task state behind them is
        updated in place, so the TimeStep gets its own copies (`copy_outputs=False` hands
        out the live buffers instead)."""
        if not self._copy_outputs:
            return ts
        # (one multi-tensor copy instead of a device memcpy per field: ~9 launches per step between two steps'
        # kernels, when nothing else runs on the GPU)
        keys = list(ts.observation)
        src = [ts.step_type] + [v for v in (ts.reward, ts.discount) if v is not None] + [ts.observation[k] for k in keys]
        dst = [torch.empty_like(v, memory_format=torch.contiguous_format) for v in src]
        groups = {}
        for d, v in zip(dst, src):   # (the multi-tensor kernel takes one dtype per call)
            groups.setdefault(v.dtype, ([], []))[0].append(d); groups[v.dtype][1].append(v)
        for ds, vs in groups.values():
            if len(ds) > 1: torch._foreach_copy_(ds, vs)
            else: ds[0].copy_(vs[0])
        it = iter(dst)
        st = next(it)
        rw = None if ts.reward is None else next(it)
        dc = None if ts.discount is None else next(it)
        return TimeStep(st, rw, dc, {k: next(it) for k in keys})
