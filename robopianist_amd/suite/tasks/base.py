"""Vectorised base piano tasks.

Mirror of robopianist/suite/tasks/base.py (PianoOnlyTask :42-90, PianoTask
:93-197) and of the entity-side state logic the tasks rely on
(Piano._update_key_state, models/piano/piano.py:178-192; ShadowHand accessors,
models/hands/shadow_hand.py:313-387).  Everything is batched over `n_envs` and
lives in torch tensors on the physics device.
"""

from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from robopianist_amd.model import scene as scene_lib
from robopianist_amd.model import shadow_hand as hand_model
from robopianist_amd.suite import specs

_PHYSICS_TIMESTEP = scene_lib.PHYSICS_TIMESTEP  # base.py:28
_CONTROL_TIMESTEP = scene_lib.CONTROL_TIMESTEP  # base.py:31

# piano.py:31-32
_KEY_THRESHOLD = 0.00872665  # 0.5 degrees.
_SUSTAIN_THRESHOLD = 0.5


class Piano:
    """Batched key / sustain state (models/piano/piano.py:164-224)."""

    def __init__(self, scene_info, add_actuators: bool):
        m = scene_info.model
        self._add_actuators = add_actuators
        self.joints = scene_info.key_joint_ids
        self.key_geom_ids = scene_info.key_geom_ids
        self.actuators = scene_info.key_actuator_ids
        self._qpos_range_np = m.jnt_range[self.joints].copy()
        self.n_keys = len(self.joints)
        self.size = scene_info.piano_size
        if add_actuators:
            self._ctrl_midpoint_np = np.mean(m.actuator_ctrlrange[self.actuators], axis=1)

    def bind(self, n_envs, device, dtype):
        self._E, self._device, self._dtype = n_envs, device, dtype
        f = dict(device=device, dtype=dtype)
        self._qpos_range = torch.as_tensor(self._qpos_range_np, **f)
        self._jidx = torch.as_tensor(self.joints, device=device, dtype=torch.long)
        if self._add_actuators:
            self._ctrl_midpoint = torch.as_tensor(self._ctrl_midpoint_np, **f)
            self._aidx = torch.as_tensor(self.actuators, device=device, dtype=torch.long)
        self._initialize_state()

    def _initialize_state(self, mask=None):
        E, n = self._E, self.n_keys
        f = dict(device=self._device, dtype=self._dtype)
        # State tensors are allocated once and only ever updated in place, so that a whole
        # env.step can be captured in (and replayed from) a hipGraph.
        if not hasattr(self, "_state"):
            self._state = torch.zeros((E, n), **f)
            self._sustain_state = torch.zeros((E, 1), **f)
            self._activation = torch.zeros((E, n), device=self._device, dtype=torch.bool)
            self._sustain_activation = torch.zeros((E, 1), device=self._device, dtype=torch.bool)
            self._normalized_state = torch.zeros((E, n), **f)
        elif mask is None:
            for t in (self._state, self._sustain_state, self._activation,
                      self._sustain_activation, self._normalized_state):
                t.zero_()
        else:
            for t in (self._state, self._sustain_state, self._activation,
                      self._sustain_activation, self._normalized_state):
                t.masked_fill_(mask[:, None], 0)

    def initialize_episode(self, physics, mask=None):
        self._initialize_state(mask)
        self._update_key_state(physics)

    def _update_key_state(self, physics):
        """piano.py:178-192."""
        if self._add_actuators:
            torch.ge(physics.ctrl[:, self._aidx], self._ctrl_midpoint, out=self._activation)
        else:
            joints_pos = physics.qpos[:, self._jidx]
            torch.clamp(joints_pos, min=self._qpos_range[:, 0], max=self._qpos_range[:, 1], out=self._state)
            torch.div(self._state, self._qpos_range[:, 1], out=self._normalized_state)
            torch.le(torch.abs(self._state - self._qpos_range[:, 1]), _KEY_THRESHOLD, out=self._activation)
        torch.ge(self._sustain_state, _SUSTAIN_THRESHOLD, out=self._sustain_activation)

    _STATE = ("_state", "_sustain_state", "_activation", "_sustain_activation", "_normalized_state")

    def state_dict(self):
        return {k: getattr(self, k).detach().clone() for k in self._STATE}

    def load_state_dict(self, sd):
        for k in self._STATE:
            getattr(self, k).copy_(sd[k].to(getattr(self, k).device))

    def apply_sustain(self, sustain):
        self._sustain_state.copy_(sustain.reshape(self._E, 1))

    @property
    def activation(self):
        return self._activation

    @property
    def sustain_activation(self):
        return self._sustain_activation

    @property
    def state(self):
        return self._state

    @property
    def normalized_state(self):
        return self._normalized_state

    @property
    def sustain_state(self):
        return self._sustain_state

    def is_key_black(self, key_id: int) -> bool:
        return bool([0, 1, 0, 0, 1, 0, 1, 0, 0, 1, 0, 1][key_id % 12])


class Hand:
    """Python-order views of one hand (shadow_hand.py:313-387)."""

    def __init__(self, info: scene_lib.HandInfo, model):
        self._info = info
        self._m = model
        self.name = info.name
        self.hand_side = info.side
        self.joints = info.joint_ids
        self.actuators = info.actuator_ids
        self.fingertip_sites = info.fingertip_site_ids
        self.n_forearm_dofs = info.n_forearm_dofs
        self.root_body_id = info.root_body_id
        self.root_site_id = info.root_site_id
        self.forearm_geom_ids = info.forearm_geom_ids
        self._action_spec = None

    def action_spec(self, physics=None) -> specs.BoundedArray:
        """spec_utils.create_action_spec: bounds = actuator ctrlrange."""
        if self._action_spec is None:
            cr = self._m.actuator_ctrlrange[self.actuators]
            names = "\t".join(self._m.names["actuator"][a] for a in self.actuators)
            self._action_spec = specs.BoundedArray((len(self.actuators),), np.float64,
                                                   cr[:, 0], cr[:, 1], name=names)
        return self._action_spec


class PianoOnlyTask:
    """Piano task with no hands (base.py:42-90)."""

    def __init__(self, add_piano_actuators: bool = False,
                 change_color_on_activation: bool = False,
                 physics_timestep: float = _PHYSICS_TIMESTEP,
                 control_timestep: float = _CONTROL_TIMESTEP, _hands: Sequence[str] = (),
                 **scene_kwargs):
        del change_color_on_activation  # cosmetic (piano.py:194-206), headless engine
        self.physics_timestep = physics_timestep
        self.control_timestep = control_timestep
        n = control_timestep / physics_timestep
        if abs(n - round(n)) > 1e-6:
            raise ValueError("Control timestep must be an integer multiple of the physics timestep.")
        self.physics_steps_per_control_step = int(round(n))
        self.scene = scene_lib.build_scene(
            hands=_hands, add_piano_actuators=add_piano_actuators,
            physics_timestep=physics_timestep, **scene_kwargs)
        self._piano = Piano(self.scene, add_piano_actuators)

    @property
    def piano(self) -> Piano:
        return self._piano

    def bind(self, physics, n_envs, random_state):
        self._physics_device = physics.device
        self._E = n_envs
        self._dtype = physics.dtype
        self._random_state = random_state
        self._piano.bind(n_envs, physics.device, physics.dtype)

    def get_reward(self, physics):
        return torch.zeros(self._E, device=self._physics_device, dtype=self._dtype)

    def get_discount(self, physics):
        return torch.ones(self._E, device=self._physics_device, dtype=self._dtype)


class PianoTask(PianoOnlyTask):
    """Base class for tasks with two Shadow Hands (base.py:93-197)."""

    def __init__(self, gravity_compensation: bool = False,
                 change_color_on_activation: bool = False,
                 primitive_fingertip_collisions: bool = False,
                 reduced_action_space: bool = False, attachment_yaw: float = 0.0,
                 forearm_dofs: Sequence[str] = hand_model.DEFAULT_FOREARM_DOFS,
                 physics_timestep: float = _PHYSICS_TIMESTEP,
                 control_timestep: float = _CONTROL_TIMESTEP,
                 disable_hand_collisions: bool = False, _hands=("right", "left"),
                 _root_sites: bool = False, mesh_colliders: int = 0, standin_wrist_clearance: bool = True,
                 cylinder_colliders: bool = False, impratio: Optional[float] = None):
        super().__init__(
            add_piano_actuators=False, change_color_on_activation=change_color_on_activation,
            physics_timestep=physics_timestep, control_timestep=control_timestep, _hands=_hands,
            gravity_compensation=gravity_compensation,
            primitive_fingertip_collisions=primitive_fingertip_collisions,
            reduced_action_space=reduced_action_space, attachment_yaw=attachment_yaw,
            forearm_dofs=forearm_dofs, disable_hand_collisions=disable_hand_collisions,
            root_sites=_root_sites, mesh_colliders=mesh_colliders,   # (extensions: model/scene.py)
            standin_wrist_clearance=standin_wrist_clearance, cylinder_colliders=cylinder_colliders, impratio=impratio)
        m = self.scene.model
        self._right_hand = Hand(self.scene.hands["right"], m) if "right" in self.scene.hands else None
        self._left_hand = Hand(self.scene.hands["left"], m) if "left" in self.scene.hands else None

    @property
    def left_hand(self) -> Optional[Hand]:
        return self._left_hand

    @property
    def right_hand(self) -> Optional[Hand]:
        return self._right_hand
