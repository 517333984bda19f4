"""Vectorised `PianoWithOneShadowHand` task.

Same constructor arguments, hooks, observables and reward terms as
robopianist/suite/tasks/piano_with_one_shadow_hand.py (line references below), batched
over n_envs on the same engine (the scene is built with one hand only, so the kernels see
a single 26-link tree).  Differences to the two-hand task, all taken from the reference:

  * action = [hand actuators (22), sustain]                             (:178-192, :194-197)
  * rewards: key press, sustain, energy (fixed coefficient), fingering;  no forearm / OT term
    (:113-121)
  * the fingering observable has 5 entries and only lists the notes whose fingering belongs
    to this hand (fingers 0-4 right, 5-9 left); a note without fingering (-1) counts as the
    right hand's last finger, as Python's negative indexing does in the reference (:292-311)
  * observables: the hand's `joints_pos` and `position` (root body xpos), piano state,
    sustain state, goal, fingering                                       (:313-352)

On the HIP engine the hooks run as the fused task kernels in their one-hand mode
(`rp_task_reward_args.hand_filter`); the torch methods below are the definition / cross-check.
"""

from __future__ import annotations

from typing import Optional, Sequence, Union

import numpy as np
import torch

from robopianist_amd.model.shadow_hand import HandSide
from robopianist_amd.music import midi_file
from robopianist_amd.suite import composite_reward, specs
from robopianist_amd.suite.rewards import tolerance
from robopianist_amd.suite.tasks import piano_with_shadow_hands as two_hands

# piano_with_one_shadow_hand.py:33-36
_FINGER_CLOSE_ENOUGH_TO_KEY = 0.01
_KEY_CLOSE_ENOUGH_TO_PRESSED = 0.05
_ENERGY_PENALTY_COEF = 5e-3


def _side_name(hand_side: Union[HandSide, str]) -> str:
    if isinstance(hand_side, HandSide):
        return "left" if hand_side == HandSide.LEFT else "right"
    if hand_side in ("left", "right"):
        return hand_side
    raise ValueError(f"Invalid hand side: {hand_side!r}.")


class PianoWithOneShadowHand(two_hands.PianoWithShadowHands):
    def __init__(
        self,
        midi: midi_file.MidiFile,
        hand_side: Union[HandSide, str],
        n_steps_lookahead: int = 1,
        n_seconds_lookahead: Optional[float] = None,
        trim_silence: bool = False,
        wrong_press_termination: bool = False,
        initial_buffer_time: float = 0.0,
        disable_fingering_reward: bool = False,
        disable_colorization: bool = False,
        augmentations: Optional[Sequence] = None,
        **kwargs,
    ) -> None:
        self._hand_side = _side_name(hand_side)
        super().__init__(
            midi=midi, n_steps_lookahead=n_steps_lookahead, n_seconds_lookahead=n_seconds_lookahead,
            trim_silence=trim_silence, wrong_press_termination=wrong_press_termination,
            initial_buffer_time=initial_buffer_time, disable_fingering_reward=disable_fingering_reward,
            disable_forearm_reward=True, disable_colorization=disable_colorization,
            augmentations=augmentations, energy_penalty_coef=_ENERGY_PENALTY_COEF,
            _hands=(self._hand_side,), _root_sites=True, **kwargs)
        # :96 — unlike the two-hand task, a MIDI without fingering does not disable the term
        self._disable_fingering_reward = disable_fingering_reward
        self._hand = self.left_hand if self._hand_side == "left" else self.right_hand
        self._set_rewards()

    # -- construction ------------------------------------------------------------------
    def _set_rewards(self) -> None:
        """:113-121."""
        self._reward_fn = composite_reward.CompositeReward(
            key_press_reward=self._compute_key_press_reward,
            sustain_reward=self._compute_sustain_reward,
            energy_reward=self._compute_energy_reward,
        )
        if not self._disable_fingering_reward:
            self._reward_fn.add("fingering_reward", self._compute_fingering_reward)

    def _bind_hands(self):
        dev = self._physics_device
        self._act = torch.as_tensor(self._hand.actuators, device=dev, dtype=torch.long)
        self._jnt = torch.as_tensor(self._hand.joints, device=dev, dtype=torch.long)
        self._tip_sites = list(self._hand.fingertip_sites)
        self._bind_key_geometry()

    def _bind_task_state(self):
        super()._bind_task_state()
        self._fingering_state = torch.zeros((self._E, 5), device=self._physics_device, dtype=self._dtype)

    def _fused_rewards_for(self, physics):
        """The fused HIP task kernels (include/rp_task.h) in their one-hand mode."""
        if not self._use_fused_rewards or not getattr(physics, "device", None) or physics.device.type != "cuda":
            return None
        want = ("key_press_reward", "sustain_reward", "energy_reward") + (
            ("fingering_reward",) if not self._disable_fingering_reward else ())
        if tuple(self._reward_fn.reward_fns) != want:
            return None  # customised reward set: torch path
        if self._fused_rewards is None:
            from robopianist_amd import task_kernels
            self._fused_rewards = task_kernels.FusedRewards(
                physics, n_envs=self._E, key_qadr=[int(j) for j in self.piano.joints],
                key_anchor=self._key_anchor, key_half=self._key_half, hand_act=list(self._hand.actuators),
                tip_site=[physics._site_modelid[int(s)] for s in self._tip_sites], rfa=[], lfa=[],
                use_fingering=not self._disable_fingering_reward, use_forearm=False,
                energy_coef=_ENERGY_PENALTY_COEF, key_close=_KEY_CLOSE_ENOUGH_TO_PRESSED,
                finger_close=_FINGER_CLOSE_ENOUGH_TO_KEY,
                hand_filter=1 if self._hand_side == "right" else 2)
        return self._fused_rewards

    @property
    def hand_side(self) -> str:
        return self._hand_side

    def fused_prestep_for(self, physics):
        """rp_task_prestep with this hand's actuators (include/rp_task.h)."""
        if self.fused_advance_for(physics) is None:
            return None
        if type(self).before_step is not PianoWithOneShadowHand.before_step:   # (a subclass's own hook is not bypassed)
            return None
        if getattr(self, "_fused_prestep", None) is None:
            from robopianist_amd import task_kernels
            hand_act = [int(x) for x in torch.as_tensor(self._act).reshape(-1).tolist()]
            self._fused_prestep = task_kernels.FusedPrestep(
                physics, n_envs=self._E, n_action=len(hand_act) + 1, hand_act=hand_act,
                sustain_state=self.piano._sustain_state)
        return self._fused_prestep

    # -- hooks ---------------------------------------------------------------------------
    def before_step(self, physics, action) -> None:
        """:194-197."""
        action = torch.as_tensor(action, device=self._physics_device, dtype=self._dtype)
        action = action.reshape(self._E, -1)
        self.piano.apply_sustain(action[:, -1])
        physics.ctrl[:, self._act] = action[:, :-1]

    def action_spec(self, physics=None) -> specs.BoundedArray:
        """:178-192."""
        hand_spec = self._hand.action_spec()
        sustain_spec = specs.BoundedArray((1,), hand_spec.dtype, [0.0], [1.0], name="sustain")
        return specs.merge_specs([hand_spec, sustain_spec])

    # -- observations ----------------------------------------------------------------------
    def _update_fingering_state(self) -> None:
        """:292-311.  `_finger_next[e, key]` = index into this hand's fingertips of the note
        on `key` at the current step, -1 if the key is not in the goal or belongs to the
        other hand."""
        slen = self._song_len[self._song_id]
        live = self._t_idx < slen
        idx = torch.clamp(self._t_idx, max=self._finger_bank.shape[1] - 1)
        f = self._finger_bank[self._song_id, idx]
        goal_now = self._goal_bank[self._song_id, idx][:, :88] > 0
        if self._hand_side == "right":
            mine = goal_now & (f < 5)
            local = torch.where(f < 0, torch.full_like(f, 4), f)  # fingertip_sites[-1]
        else:
            mine = goal_now & (f >= 5)
            local = f - 5
        local = torch.where(mine, local, torch.full_like(f, -1))
        self._finger_next.copy_(torch.where(live[:, None], local, self._finger_next))
        fs = torch.zeros((self._E, 5), device=f.device, dtype=self._dtype)
        has = mine & live[:, None]
        fs.scatter_add_(1, torch.where(has, local, torch.zeros_like(local)), has.to(self._dtype))
        fs = (fs > 0).to(self._dtype)
        self._fingering_state.copy_(torch.where(live[:, None], fs, self._fingering_state))

    def get_observation(self, physics):
        """Enabled observables (:313-352) in the reference's update order."""
        self._update_goal_state()
        self._update_fingering_state()
        return self._observation_dict(physics)

    def _observation_dict(self, physics):
        name = self._hand.name
        obs = {
            f"{name}/joints_pos": physics.qpos[:, self._jnt],
            f"{name}/position": physics.site_xpos([self._hand.root_site_id])[:, 0],
            "piano/state": self.piano.normalized_state,
            "piano/sustain_state": self.piano.sustain_state,
            "goal": self._goal_state.reshape(self._E, -1),
        }
        if not self._disable_fingering_reward:
            obs["fingering"] = self._fingering_state
        self._add_optional_observables(physics, obs)
        return obs

    def observation_spec(self):
        L, d, name = self._n_steps_lookahead, np.float64, self._hand.name
        out = {
            f"{name}/joints_pos": specs.Array((len(self._hand.joints),), d),
            f"{name}/position": specs.Array((3,), d),
            "piano/state": specs.Array((88,), d),
            "piano/sustain_state": specs.Array((1,), d),
            "goal": specs.Array(((L + 1) * 89,), d),
        }
        if not self._disable_fingering_reward:
            out["fingering"] = specs.Array((5,), d)
        self._add_optional_specs(out)
        return out

    def steps_left(self):
        """The (disabled by default) `steps_left` observable (:346-350)."""
        slen = self._song_len[self._song_id].to(self._dtype)
        return (slen - self._t_idx.to(self._dtype)) / slen

    # -- rewards -----------------------------------------------------------------------------
    def _compute_energy_reward(self, physics):
        """:214-217."""
        power = physics.act_force[:, self._act].abs() * physics.act_vel[:, self._act].abs()
        return -_ENERGY_PENALTY_COEF * power.sum(1)

    def _compute_fingering_reward(self, physics):
        """:237-271 — mean over this hand's (key, finger) pairs of the step, 0 if none."""
        f = self._finger_current
        has = f >= 0
        fid = torch.clamp(f, min=0)
        tips = physics.site_xpos(self._tip_sites)  # [E, 5, 3]
        tgt = self._key_targets(physics)
        tip_for_key = torch.gather(tips, 1, fid[..., None].expand(-1, -1, 3))
        dist = torch.linalg.norm(tgt - tip_for_key, dim=-1)
        rews = tolerance(dist, bounds=(0, _FINGER_CLOSE_ENOUGH_TO_KEY),
                         margin=_FINGER_CLOSE_ENOUGH_TO_KEY * 10)
        n = has.sum(1)
        mean = (rews * has).sum(1) / torch.clamp(n, min=1)
        return torch.where(n > 0, mean, torch.zeros_like(mean))
