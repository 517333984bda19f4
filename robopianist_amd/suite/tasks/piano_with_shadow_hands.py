"""Vectorised `PianoWithShadowHands` task.

Same constructor arguments, hook order, observation names, reward terms and
termination rule as robopianist/suite/tasks/piano_with_shadow_hands.py (line
references below), batched over n_envs.  `midi` may also be a list of MidiFiles:
env e then plays song e % len(midi) (heterogeneous goal bank, BASELINE config #5).

`augmentations` (suite/variations.py) switch the goal bank to one slot per env: at every
episode start of env e the variations are applied to that env's initial MIDI on the host
(reference: _maybe_change_midi, :151-157), its goal / fingering tables are rebuilt and
uploaded into slot e.  This needs the set of resetting envs on the host, i.e. one small
device->host read per control step, and ~0.2 ms of host work per episode start.
`augmentation_prefetch=True` (an extension) removes the read: every env owns two bank slots, the host
keeps the idle one filled with the tables of the env's next episode, the fused launch switches slots
when an episode starts and raises a flag the host polls asynchronously
(include/rp_task.h `next_ready` / `consumed`).  Draws then happen ahead of time, in refill order.
"""

from __future__ import annotations

from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from robopianist_amd.music import midi_file
from robopianist_amd.suite import composite_reward, specs
from robopianist_amd.suite.rewards import tolerance
from robopianist_amd.suite.tasks import base

# piano_with_shadow_hands.py:35-46
_FINGER_CLOSE_ENOUGH_TO_KEY = 0.01
_KEY_CLOSE_ENOUGH_TO_PRESSED = 0.05
_ENERGY_PENALTY_COEF = 5e-3
_POSITION_OFFSET = 0.05


class PianoWithShadowHands(base.PianoTask):
    def __init__(
        self,
        midi: Union[midi_file.MidiFile, Sequence[midi_file.MidiFile]],
        n_steps_lookahead: int = 1,
        n_seconds_lookahead: Optional[float] = None,
        trim_silence: bool = False,
        wrong_press_termination: bool = False,
        initial_buffer_time: float = 0.0,
        disable_fingering_reward: bool = False,
        disable_forearm_reward: bool = False,
        disable_colorization: bool = False,
        disable_hand_collisions: bool = False,
        augmentations=None,
        energy_penalty_coef: float = _ENERGY_PENALTY_COEF,
        randomize_hand_positions: bool = False,
        augmentation_prefetch: bool = False,
        overflow_termination: bool = False,
        **kwargs,
    ) -> None:
        super().__init__(disable_hand_collisions=disable_hand_collisions, **kwargs)
        # extension (not in the reference): keep the tables of every env's NEXT episode ready in a
        # second bank slot, so that augmentations need no device->host read per step
        self._prefetch = bool(augmentation_prefetch) and augmentations is not None
        del disable_colorization  # cosmetic (:451-474)
        self._augmentations = list(augmentations) if augmentations is not None else None
        midis = list(midi) if isinstance(midi, (list, tuple)) else [midi]
        if trim_silence:
            midis = [m.trim_silence() for m in midis]
        self._midis = midis
        self._midi = midis[0]
        self._initial_midis = list(midis)  # :104 `_initial_midi`
        self._n_steps_lookahead = n_steps_lookahead
        if n_seconds_lookahead is not None:
            self._n_steps_lookahead = int(np.ceil(n_seconds_lookahead / self.control_timestep))
        self._initial_buffer_time = initial_buffer_time
        self._disable_fingering_reward = disable_fingering_reward or not all(
            m.has_fingering() for m in midis)
        self._disable_forearm_reward = disable_forearm_reward
        self._wrong_press_termination = wrong_press_termination
        self._disable_hand_collisions = disable_hand_collisions
        self._energy_penalty_coef = energy_penalty_coef
        self._randomize_hand_positions = randomize_hand_positions
        # The reference ends an episode only at the end of the MIDI or on a wrong press (:212-220), and so does this
        # task by default: an engine capacity overflow (more than 64 contacts / 640 contact Jacobian entries / 12
        # touched keys / 57 cross-coupled rows in one env: the shallowest contacts / the excess are dropped) raises
        # the env's warn flag and is COUNTED (`overflow_episodes`; < 1e-5 of the env-steps of a uniformly random
        # policy since round 4), the episode goes on.  Extension, `overflow_termination=True` (the default of rounds
        # 2-3, when a third of a percent of such env-steps overflowed): the overflow ends the episode like a diverged
        # state does -- reward 0, discount 0.
        from robopianist_amd import engine as _eng
        self.overflow_warn_mask = _eng.WARN_CONTACT_FULL | _eng.WARN_KEYSLOT_FULL | _eng.WARN_DENSE_FULL | _eng.WARN_WORK_FULL | _eng.WARN_SPLIT_FULL
        self.fatal_warn_mask = _eng.WARN_BADSTATE | (self.overflow_warn_mask if overflow_termination else 0)
        self._fatal_count = None
        self._overflow_count = None
        self._use_fused_rewards = True   # set False to force the torch reward functions
        self._fused_rewards = None
        self._use_fused_advance = True   # set False to force the torch task hooks
        self._fused_advance = None; self._fused_prestep = None
        self._reset_trajectory()
        self._set_rewards()

    # -- construction ------------------------------------------------------------
    def _set_rewards(self) -> None:
        """:130-144 (insertion order matters: CompositeReward sums in order)."""
        self._reward_fn = composite_reward.CompositeReward(
            key_press_reward=self._compute_key_press_reward,
            sustain_reward=self._compute_sustain_reward,
            energy_reward=self._compute_energy_reward,
        )
        if not self._disable_fingering_reward:
            self._reward_fn.add("fingering_reward", self._compute_fingering_reward)
        else:
            self._reward_fn.add("ot_fingering_reward", self._compute_ot_fingering_reward)
        if not self._disable_forearm_reward:
            self._reward_fn.add("forearm_reward", self._compute_forearm_reward)

    def _reset_trajectory(self) -> None:
        """:159-165 — one NoteTrajectory per song; dense [song, T, .] goal bank."""
        trajs = []
        for m in self._midis:
            t = midi_file.NoteTrajectory.from_midi(m, self.control_timestep)
            t.add_initial_buffer_time(self._initial_buffer_time)
            trajs.append(t)
        self._trajs = trajs
        self._notes = trajs[0].notes
        self._sustains = trajs[0].sustains
        tmax = max(len(t) for t in trajs)
        ns = len(trajs)
        goal = np.zeros((ns, tmax, 89), np.float32)
        finger = np.full((ns, tmax, 88), -1, np.int64)
        self._song_tables = []
        for s, t in enumerate(trajs):
            g, f = t.to_goal_tables()
            goal[s, :len(t)] = g
            finger[s, :len(t)] = f
            self._song_tables.append((g, f))
        self._goal_bank_np, self._finger_bank_np = goal, finger
        self._song_len_np = np.array([len(t) for t in trajs], np.int64)

    def _tables_for(self, midi):
        """Goal tables of one (augmented) MIDI (:159-165)."""
        fast = midi_file.NoteTrajectory.goal_tables_from_arrays(
            midi.note_arrays(), self.control_timestep, self._initial_buffer_time)
        if fast is not None:
            return fast
        t = midi_file.NoteTrajectory.from_midi(midi, self.control_timestep)
        t.add_initial_buffer_time(self._initial_buffer_time)
        return t.to_goal_tables()

    def _maybe_change_midi(self, mask=None) -> None:
        """:151-157 for the envs selected by `mask` (None = all): the variations are applied
        to the env's initial MIDI with the task's RandomState, in env order."""
        if self._augmentations is None:
            return
        if self._prefetch:
            if mask is not None:
                raise RuntimeError("augmentation_prefetch runs on the fused HIP task path only")
            self._prefetch_full_reset()
            return
        E = self._E
        envs = range(E) if mask is None else np.flatnonzero(mask.detach().cpu().numpy())
        if len(envs) == 0:
            return
        self._fill_slots(np.asarray(envs, np.int64), envs)

    def _draw_midis(self, envs):
        """Applies the variations to the initial MIDI of each env in `envs` (task RandomState, in
        that order)."""
        ns = len(self._initial_midis)
        out = []
        for e in envs:
            midi = self._initial_midis[e % ns]
            for var in self._augmentations:
                midi = var(initial_value=midi, random_state=self._random_state)
            self._env_midi[e] = midi
            out.append(midi)
        return out

    def _device_rasterizer(self):
        """rp_task_rasterize over the task's initial songs, or None (CPU double, switched off, or
        a song the kernel does not cover: notes off the 88 keys / velocities > 127)."""
        if not getattr(self, "_use_device_rasterizer", True) or self._physics_device.type != "cuda":
            return None
        if not hasattr(self, "_rasterizer"):
            from robopianist_amd import task_kernels
            arrays = [m.note_arrays() for m in self._initial_midis]
            ok = all(len(a.pitch) and a.pitch.min() >= 21 and a.pitch.max() <= 108 and a.velocity.max() <= 127
                     for a in arrays)
            self._rasterizer = task_kernels.Rasterizer(
                self._physics_device, self._dtype, arrays, self.control_timestep,
                self._initial_buffer_time) if ok else None
        return self._rasterizer

    def _fill_slots(self, slots, envs) -> None:
        """New augmented songs for `envs` into the bank slots `slots`: the draws happen on the
        host; the tables are rasterised on the device when the drawn MIDI is a stretch / transpose
        chain over the env's initial song (include/rp_task.h rp_task_rasterize), on the host
        otherwise (e.g. MidiSelect)."""
        midis = self._draw_midis(envs)
        ns = len(self._initial_midis)
        ras = self._device_rasterizer()
        on_dev, on_host = [], []
        for i, (e, midi) in enumerate(zip(envs, midis)):
            initial = self._initial_midis[e % ns]
            extra = midi._ops[len(initial._ops):]
            if (ras is not None and midi._base is initial._base and midi._ops[:len(initial._ops)] == initial._ops
                    and len(extra) <= ras.MAX_OPS):
                on_dev.append((i, e % ns, extra))
            else:
                on_host.append(i)
        if on_dev:
            fps = 1 / self.control_timestep
            nbuf = int(round(self._initial_buffer_time / self.control_timestep))
            need = 0
            for _, si, extra in on_dev:
                total = self._initial_midis[si].note_arrays().total_time
                for kind, val in extra:
                    if kind == "stretch" and val != 1.0:
                        total *= val
                need = max(need, int(total * fps + 1) + nbuf)
            if need > self._goal_bank.shape[1]:
                self._grow_bank(need + need // 4)
            ras.rasterize(self._goal_bank, self._finger_bank, self._song_len,
                          [int(slots[i]) for i, _, _ in on_dev], [si for _, si, _ in on_dev],
                          [list(x) for _, _, x in on_dev])
        if on_host:
            rows = []
            for i in on_host:
                initial = self._initial_midis[envs[i] % ns]
                rows.append(self._song_tables[envs[i] % ns] if midis[i] is initial else self._tables_for(midis[i]))
            self._upload_tables(np.asarray([slots[i] for i in on_host], np.int64), rows)

    def _upload_tables(self, slots, rows) -> None:
        """Writes `rows` into the bank slots `slots` (grows the bank if a song is longer)."""
        lens = [len(g) for g, _ in rows]
        need = max(lens)
        if need > self._goal_bank.shape[1]:
            self._grow_bank(need + need // 4)
        cap = self._goal_bank.shape[1]
        goal = np.zeros((len(rows), cap, 89), np.float32)
        finger = np.full((len(rows), cap, 88), -1, np.int64)
        for i, (g, f) in enumerate(rows):
            goal[i, :len(g)] = g
            finger[i, :len(g)] = f
        dev = self._physics_device
        idx = torch.as_tensor(np.asarray(slots, np.int64), device=dev)
        self._goal_bank.index_copy_(0, idx, torch.as_tensor(goal, device=dev).to(self._dtype))
        self._finger_bank.index_copy_(0, idx, torch.as_tensor(finger, device=dev))
        self._song_len.index_copy_(0, idx, torch.as_tensor(np.asarray(lens, np.int64), device=dev))

    # -- prefetch mode ---------------------------------------------------------------------
    def _prefetch_full_reset(self) -> None:
        """Environment.reset(): current tables into slot 2e, next episode's into 2e+1."""
        E = self._E
        envs = np.arange(E, dtype=np.int64)
        self._fill_slots(2 * envs, envs)
        self._fill_slots(2 * envs + 1, envs)
        self._song_id.copy_(torch.as_tensor(2 * envs, device=self._physics_device))
        self._host_parity = np.zeros(E, np.int64)
        self._next_ready.fill_(1)
        self._consumed.zero_()
        self._poll_event = None
        self.prefetch_refills = 0

    def _prefetch_poll(self) -> None:
        """Non-blocking: if the last snapshot of the `consumed` flags has arrived, refill the
        freed slots, then request the next snapshot."""
        ev = self._poll_event
        if ev is not None and not ev.query():
            return
        if ev is not None:
            envs = np.flatnonzero(self._consumed_host.numpy())
            if len(envs):
                self._host_parity[envs] ^= 1                       # those envs switched slots
                free = 2 * envs + (self._host_parity[envs] ^ 1)
                self._fill_slots(free, envs)
                idx = torch.as_tensor(envs, device=self._physics_device)
                self._consumed.index_fill_(0, idx, 0)              # (stream-ordered after the upload)
                self._next_ready.index_fill_(0, idx, 1)
                self.prefetch_refills += len(envs)
        self._consumed_host.copy_(self._consumed, non_blocking=True)
        self._poll_event = torch.cuda.Event()
        self._poll_event.record(torch.cuda.current_stream(self._physics_device))

    def _grow_bank(self, cap: int) -> None:
        """Reallocates the per-env bank with `cap` rows (the fused launch arguments hold
        raw pointers, so they are rebuilt on next use)."""
        old_g, old_f = self._goal_bank, self._finger_bank
        n, t = old_g.shape[0], old_g.shape[1]
        self._goal_bank = torch.zeros((n, cap, 89), device=old_g.device, dtype=old_g.dtype)
        self._finger_bank = torch.full((n, cap, 88), -1, device=old_f.device, dtype=old_f.dtype)
        self._goal_bank[:, :t] = old_g
        self._finger_bank[:, :t] = old_f
        self._fused_advance = None; self._fused_prestep = None

    def prepare_episodes(self, physics, mask) -> None:
        """Host-side part of initialize_episode that the fused device path cannot do:
        the MIDI augmentations of the envs that start an episode in this step."""
        del physics
        if self._augmentations is None:
            return
        if self._prefetch:
            self._prefetch_poll()
        else:
            self._maybe_change_midi(mask)

    @property
    def needs_host_episode_setup(self) -> bool:
        return self._augmentations is not None

    def bind(self, physics, n_envs, random_state):
        super().bind(physics, n_envs, random_state)
        self._fused_advance = None; self._fused_prestep = None   # (both hold raw pointers into the bound physics)
        self._bind_goal_bank()
        self._bind_hands()
        self._bind_task_state()
        self._fatal_count = torch.zeros(n_envs, device=physics.device, dtype=torch.long)
        self._overflow_count = torch.zeros(n_envs, device=physics.device, dtype=torch.long)
        self._physics = physics
        if any(n.split("/")[1] in ("joints_torque", "fingertip_force") for n in getattr(self, "_extra_observables", ())):
            physics.enable_acc_sensors()
        if self._prefetch:
            # both slots of every env are filled before the first step, whether or not the caller
            # starts with an explicit reset()
            self._prefetch_full_reset()

    def count_fatal(self, warn, active) -> None:
        """Torch path: counts the episodes ended by an engine warn flag (see fatal_warn_mask)."""
        self._fatal_count.add_(((warn & self.fatal_warn_mask) != 0) & active)

    def count_overflow(self, warn, terminate) -> None:
        """Torch path: counts the episodes that end with a capacity-overflow flag raised."""
        self._overflow_count.add_(((warn & self.overflow_warn_mask) != 0) & terminate)

    def overflow_episodes(self) -> int:
        """Episodes that ended (for any reason) after an engine capacity overflow, all envs."""
        n = int(self._overflow_count.sum().item()) if self._overflow_count is not None else 0
        if self._fused_advance is not None:
            n += int(self._fused_advance.warn_count.sum().item())
        return n

    def overflow_terminations(self) -> int:
        """Episodes ended by a warn flag so far (bad state or capacity overflow), all envs."""
        n = int(self._fatal_count.sum().item()) if self._fatal_count is not None else 0
        if self._fused_advance is not None:
            n += int(self._fused_advance.fatal_count.sum().item())
        return n

    def _bind_goal_bank(self):
        dev, E = self._physics_device, self._E
        self._goal_bank = torch.as_tensor(self._goal_bank_np, device=dev, dtype=self._dtype)
        self._finger_bank = torch.as_tensor(self._finger_bank_np, device=dev)
        self._song_len = torch.as_tensor(self._song_len_np, device=dev)
        self._song_id = torch.arange(E, device=dev) % len(self._midis)
        if self._augmentations is not None:
            # one bank slot per env, initialised with the env's un-augmented song
            self._goal_bank = self._goal_bank[self._song_id].contiguous()
            self._finger_bank = self._finger_bank[self._song_id].contiguous()
            self._song_len = self._song_len[self._song_id].contiguous()
            self._song_id = torch.arange(E, device=dev)
            self._env_midi = [self._initial_midis[e % len(self._initial_midis)] for e in range(E)]
            self._fused_advance = None; self._fused_prestep = None
            if self._prefetch:
                if dev.type != "cuda":
                    raise ValueError("augmentation_prefetch needs the HIP task path (a GPU)")
                self._goal_bank = self._goal_bank.repeat_interleave(2, dim=0).contiguous()
                self._finger_bank = self._finger_bank.repeat_interleave(2, dim=0).contiguous()
                self._song_len = self._song_len.repeat_interleave(2).contiguous()
                self._song_id = 2 * torch.arange(E, device=dev)
                self._next_ready = torch.zeros(E, dtype=torch.uint8, device=dev)
                self._consumed = torch.zeros(E, dtype=torch.uint8, device=dev)
                self._consumed_host = torch.zeros(E, dtype=torch.uint8).pin_memory()
                self._host_parity = np.zeros(E, np.int64)
                self._poll_event = None
                self.prefetch_refills = 0

    def _bind_hands(self):
        dev, m = self._physics_device, self.scene.model
        self._rh_act = torch.as_tensor(self.right_hand.actuators, device=dev, dtype=torch.long)
        self._lh_act = torch.as_tensor(self.left_hand.actuators, device=dev, dtype=torch.long)
        self._rh_jnt = torch.as_tensor(self.right_hand.joints, device=dev, dtype=torch.long)
        self._lh_jnt = torch.as_tensor(self.left_hand.joints, device=dev, dtype=torch.long)
        # fingertip sites in fingering-id order: 0-4 right (th..lf), 5-9 left
        self._tip_sites = list(self.right_hand.fingertip_sites) + list(self.left_hand.fingertip_sites)
        self._bind_key_geometry()
        self._rfa = torch.as_tensor(self.right_hand.forearm_geom_ids, device=dev, dtype=torch.int32)
        self._lfa = torch.as_tensor(self.left_hand.forearm_geom_ids, device=dev, dtype=torch.int32)

    def _bind_key_geometry(self):
        """Key hinge positions and box half sizes (fingering target, :312-313)."""
        dev, m = self._physics_device, self.scene.model
        kg = self.piano.key_geom_ids
        kb = m.geom_bodyid[kg]
        self._key_anchor = torch.as_tensor(m.body_pos[kb] + m.jnt_pos[self.piano.joints],
                                           device=dev, dtype=self._dtype)
        self._key_half = torch.as_tensor(m.geom_size[kg], device=dev, dtype=self._dtype)

    def _bind_task_state(self):
        dev, E = self._physics_device, self._E
        self._reset_quantities_at_episode_init()
        L = self._n_steps_lookahead
        self._goal_state = torch.zeros((E, L + 1, 89), device=dev, dtype=self._dtype)
        self._goal_current = torch.zeros((E, 89), device=dev, dtype=self._dtype)
        self._finger_next = torch.full((E, 88), -1, device=dev, dtype=torch.long)
        self._finger_current = torch.full((E, 88), -1, device=dev, dtype=torch.long)
        self._fingering_state = torch.zeros((E, 10), device=dev, dtype=self._dtype)
        self._failure_termination = torch.zeros(E, device=dev, dtype=torch.bool)

    def _reset_quantities_at_episode_init(self, mask=None) -> None:
        """:146-149."""
        dev, E = self._physics_device, self._E
        # (state is allocated once and updated in place: the step must be hipGraph-capturable)
        if not hasattr(self, "_t_idx"):
            self._t_idx = torch.zeros(E, device=dev, dtype=torch.long)
            self._should_terminate = torch.zeros(E, device=dev, dtype=torch.bool)
            self._discount = torch.ones(E, device=dev, dtype=self._dtype)
        elif mask is None:
            self._t_idx.zero_(); self._should_terminate.zero_(); self._discount.fill_(1.0)
        else:
            self._t_idx.masked_fill_(mask, 0)
            self._should_terminate.masked_fill_(mask, False)
            self._discount.masked_fill_(mask, 1.0)

    # -- checkpoint / resume ------------------------------------------------------------
    _STATE = ("_t_idx", "_should_terminate", "_discount", "_goal_state", "_goal_current", "_finger_next",
              "_finger_current", "_fingering_state", "_failure_termination")

    def state_dict(self):
        """Episode state of every env (the reference keeps it in Python ints / arrays:
        piano_with_shadow_hands.py:146-149, piano.py:166-171), incl. the per-env goal bank when
        MIDI augmentations are on."""
        sd = {k: getattr(self, k).detach().clone() for k in self._STATE}
        sd["piano"] = self.piano.state_dict()
        if self._augmentations is not None:
            for k in ("_goal_bank", "_finger_bank", "_song_len"):
                sd[k] = getattr(self, k).detach().clone()
        if self._prefetch:
            torch.cuda.synchronize(self._physics_device)
            for k in ("_song_id", "_next_ready", "_consumed"):
                sd[k] = getattr(self, k).detach().clone()
            sd["_host_parity"] = self._host_parity.copy()
        if hasattr(self, "_tree_offset"):
            sd["_tree_offset"] = self._tree_offset.detach().clone()
        return sd

    def load_state_dict(self, sd):
        dev = self._physics_device
        for k in self._STATE:
            getattr(self, k).copy_(sd[k].to(dev))
        self.piano.load_state_dict(sd["piano"])
        if self._augmentations is not None:
            # the bank may have a different capacity: swap it in (the fused launch holds raw
            # pointers and is rebuilt on next use)
            self._goal_bank = sd["_goal_bank"].to(dev).clone()
            self._finger_bank = sd["_finger_bank"].to(dev).clone()
            self._song_len.copy_(sd["_song_len"].to(dev))
            self._fused_advance = None; self._fused_prestep = None
        if self._prefetch:
            for k in ("_song_id", "_next_ready", "_consumed"):
                getattr(self, k).copy_(sd[k].to(dev))
            self._host_parity = np.array(sd["_host_parity"], np.int64)
            self._poll_event = None  # a pending snapshot belongs to the old state
        if "_tree_offset" in sd:
            self._tree_offset = sd["_tree_offset"].to(dev).clone()

    # -- composer-style hooks --------------------------------------------------------
    def initialize_episode(self, physics, mask=None) -> None:
        """:167-174 for the envs selected by `mask` (None = all)."""
        self._maybe_change_midi(mask)
        self._reset_quantities_at_episode_init(mask)
        self._randomize_initial_hand_positions(physics, mask)
        self.piano.initialize_episode(physics, mask)

    def before_step(self, physics, action) -> None:
        """:176-186 — action layout [right(22), left(22), sustain]."""
        action = torch.as_tensor(action, device=self._physics_device, dtype=self._dtype)
        action = action.reshape(self._E, -1)
        hands = action[:, :-1]
        n_r = len(self.right_hand.actuators)
        ctrl = physics.ctrl  # zero-copy view of the engine's ctrl array
        if getattr(self, "_hand_act", None) is None or self._hand_act.device != ctrl.device:
            self._hand_act = torch.cat([torch.as_tensor(self._rh_act, device=ctrl.device).reshape(-1).long(),
                                        torch.as_tensor(self._lh_act, device=ctrl.device).reshape(-1).long()])
            assert self._hand_act.numel() == hands.shape[1] and n_r == torch.as_tensor(self._rh_act).numel()
        ctrl.index_copy_(1, self._hand_act, hands)   # (right hand's actuators, then the left's: one launch)
        self.piano.apply_sustain(action[:, -1])

    def after_substeps(self, physics) -> None:
        """Piano.after_substep (piano.py:154-162): only the state after the last
        substep is observable by rewards; the per-substep trace is the engine's
        `key_trace`."""
        self.piano._update_key_state(physics)

    def after_step(self, physics, active=None) -> None:
        """:188-204."""
        inc = torch.ones_like(self._t_idx) if active is None else active.to(torch.long)
        self._t_idx.add_(inc)
        slen = self._song_len[self._song_id]
        torch.eq(self._t_idx, slen, out=self._should_terminate)  # (t_idx - 1) == len - 1
        self._goal_current.copy_(self._goal_state[:, 0])
        self._finger_current.copy_(self._finger_next)
        off = self._goal_current[:, :-1] == 0
        torch.any(self.piano.activation & off, dim=1, out=self._failure_termination)

    def get_reward(self, physics):
        """CompositeReward.compute (composite_reward.py:46-56).  On the HIP engine the
        standard terms are evaluated by one fused launch (include/rp_task.h); the torch
        functions above them remain the definition and are used whenever the reward set
        has been customised (or on the CPU test double)."""
        fused = self._fused_rewards_for(physics)
        if fused is None:
            return self._reward_fn.compute(physics)
        total, terms = fused.compute(
            goal_current=self._goal_current, key_norm_state=self.piano.normalized_state,
            key_activation=self.piano.activation, sustain_activation=self.piano.sustain_activation,
            finger_current=self._finger_current)
        for i in range(len(terms)):
            name = self._term_name(i)
            if name in self._reward_fn.reward_fns:
                self._reward_fn.reward_terms[name] = terms[i]
        return total

    def _fused_rewards_for(self, physics):
        if not self._use_fused_rewards or not getattr(physics, "device", None) or physics.device.type != "cuda":
            return None
        names = tuple(self._reward_fn.reward_fns)
        std = ("key_press_reward", "sustain_reward", "energy_reward")
        fing = "fingering_reward" if not self._disable_fingering_reward else "ot_fingering_reward"
        if fing == "ot_fingering_reward" and not self._ot_term_on_device():
            return None
        want = std + (fing,) + (("forearm_reward",) if not self._disable_forearm_reward else ())
        if names != want or self.piano._add_actuators:
            return None  # customised reward set: torch path
        if self._fused_rewards is None:
            from robopianist_amd import task_kernels
            tips = [physics._site_modelid[int(s)] for s in self._tip_sites]
            self._fused_rewards = task_kernels.FusedRewards(
                physics, n_envs=self._E, key_qadr=[int(j) for j in self.piano.joints],  # 1-dof joints: qpos address == joint id
                key_anchor=self._key_anchor, key_half=self._key_half,
                hand_act=list(self.right_hand.actuators) + list(self.left_hand.actuators), tip_site=tips,
                rfa=self.right_hand.forearm_geom_ids, lfa=self.left_hand.forearm_geom_ids,
                use_fingering=1 if not self._disable_fingering_reward else 2,  # 2: the OT assignment term
                use_forearm=not self._disable_forearm_reward,
                energy_coef=self._energy_penalty_coef, key_close=_KEY_CLOSE_ENOUGH_TO_PRESSED,
                finger_close=_FINGER_CLOSE_ENOUGH_TO_KEY)
        return self._fused_rewards

    def _ot_term_on_device(self) -> bool:
        """The OT fingering term (:333-369) as a wave-per-env assignment kernel (two-hand task)."""
        return getattr(self, "_use_device_ot", True) and self.right_hand is not None and self.left_hand is not None

    def _term_name(self, i):
        from robopianist_amd import task_kernels
        name = task_kernels.TERM_NAMES[i]
        return "ot_fingering_reward" if (name == "fingering_reward" and self._disable_fingering_reward) else name

    def fused_advance_for(self, physics):
        """The one-launch replacement of after_substeps .. TimeStep assembly
        (include/rp_task.h: rp_task_advance), or None when the configuration is not
        covered (custom reward set, OT fingering, hand-position randomisation, CPU double)."""
        if not self._use_fused_advance or self._randomize_hand_positions:
            return None
        rewards = self._fused_rewards_for(physics)
        if rewards is None:
            return None
        if self._fused_advance is None:
            from robopianist_amd import task_kernels
            from robopianist_amd.suite.tasks import base as _base
            self._fused_advance = task_kernels.FusedAdvance(
                rewards, n_lookahead=self._n_steps_lookahead, goal_bank=self._goal_bank,
                finger_bank=self._finger_bank, song_len=self._song_len, song_id=self._song_id,
                wrong_press_termination=self._wrong_press_termination,
                key_threshold=_base._KEY_THRESHOLD, sustain_threshold=_base._SUSTAIN_THRESHOLD,
                key_qrange=self.piano._qpos_range, warn_fatal_mask=self.fatal_warn_mask,
                warn_count_mask=self.overflow_warn_mask)
            if getattr(self, "_eval_buffers", None) is not None:
                self._fused_advance.set_evaluation_buffers(*self._eval_buffers)
            if self._prefetch:
                self._fused_advance.set_prefetch_buffers(self._next_ready, self._consumed)
        return self._fused_advance

    def fused_prestep_for(self, physics):
        """The one-launch replacement of everything between env.step(action) and physics.step() (include/rp_task.h:
        rp_task_prestep), wherever the fused advance applies."""
        if self.fused_advance_for(physics) is None:
            return None
        # a subclass with its own before_step (action noise, extra ctrl writes, another sustain rule) must not be
        # bypassed by the launch that restates the STOCK hook: it gets the before_step path of Environment.step
        if type(self).before_step is not PianoWithShadowHands.before_step:
            return None
        if getattr(self, "_fused_prestep", None) is None:
            from robopianist_amd import task_kernels
            hand_act = [int(x) for x in torch.as_tensor(self._rh_act).reshape(-1).tolist()] + \
                       [int(x) for x in torch.as_tensor(self._lh_act).reshape(-1).tolist()]
            self._fused_prestep = task_kernels.FusedPrestep(
                physics, n_envs=self._E, n_action=len(hand_act) + 1, hand_act=hand_act,
                sustain_state=self.piano._sustain_state)
        return self._fused_prestep

    def set_evaluation_buffers(self, buffers) -> None:
        """(sums, count, hist, n_finished) of a MidiEvaluationWrapper, or None: the fused launch
        then performs the wrapper's reduction (include/rp_task.h `eval_*`)."""
        self._eval_buffers = buffers
        if self._fused_advance is not None:
            self._fused_advance.set_evaluation_buffers(*(buffers or (None, None, None, None)))

    def fused_advance(self, physics, needs_reset):
        """Runs rp_task_advance; returns (step_type, reward, discount, observation)."""
        from robopianist_amd import task_kernels
        fa = self._fused_advance
        pn = self.piano
        st, total, disc, terms = fa.advance(
            needs_reset=needs_reset, key_state=pn._state, key_norm_state=pn._normalized_state,
            key_activation=pn._activation, sustain_state=pn._sustain_state,
            sustain_activation=pn._sustain_activation, t_idx=self._t_idx,
            should_terminate=self._should_terminate, failure_termination=self._failure_termination,
            discount_state=self._discount, goal_state=self._goal_state, goal_current=self._goal_current,
            finger_next=self._finger_next, finger_current=self._finger_current,
            fingering_state=self._fingering_state)
        for i in range(len(terms)):
            name = self._term_name(i)
            if name in self._reward_fn.reward_fns:
                self._reward_fn.reward_terms[name] = terms[i]
        return st, total, disc, self._observation_dict(physics)

    def get_discount(self, physics=None):
        return self._discount

    def should_terminate_episode(self, physics=None):
        """:213-220."""
        term = self._should_terminate.clone()
        if self._wrong_press_termination:
            fail = self._failure_termination & ~term
            self._discount.masked_fill_(fail, 0.0)
            term = term | self._failure_termination
        return term

    def action_spec(self, physics=None) -> specs.BoundedArray:
        """:226-237."""
        hands_spec = specs.merge_specs([self.right_hand.action_spec(), self.left_hand.action_spec()])
        sustain_spec = specs.BoundedArray((1,), hands_spec.dtype, [0.0], [1.0], name="sustain")
        return specs.merge_specs([hands_spec, sustain_spec])

    @property
    def midi(self):
        return self._midi

    @property
    def reward_fn(self):
        return self._reward_fn

    # -- observations ---------------------------------------------------------------------
    def _update_goal_state(self) -> None:
        """:371-389 — skipped for envs whose t_idx ran past the end (kept stale, as
        the reference returns early)."""
        slen = self._song_len[self._song_id]
        live = self._t_idx < slen
        L = self._n_steps_lookahead
        steps = self._t_idx[:, None] + torch.arange(L + 1, device=self._t_idx.device)[None, :]
        valid = steps < slen[:, None]
        idx = torch.clamp(steps, max=self._goal_bank.shape[1] - 1)
        g = self._goal_bank[self._song_id[:, None], idx]
        g = torch.where(valid[..., None], g, torch.zeros_like(g))
        self._goal_state.copy_(torch.where(live[:, None, None], g, self._goal_state))

    def _update_fingering_state(self) -> None:
        """:391-412."""
        slen = self._song_len[self._song_id]
        live = self._t_idx < slen
        idx = torch.clamp(self._t_idx, max=self._finger_bank.shape[1] - 1)
        f = self._finger_bank[self._song_id, idx]  # [E, 88], -1 = key not in goal
        goal_now = self._goal_bank[self._song_id, idx][:, :88] > 0
        f = torch.where(goal_now, f, torch.full_like(f, -1))
        self._finger_next.copy_(torch.where(live[:, None], f, self._finger_next))
        # observable [right 5, left 5]; a note without fingering (-1) counts as
        # right-hand finger index -1 (python indexing), as in the reference (:401-412)
        fs = torch.zeros((self._E, 10), device=f.device, dtype=self._dtype)
        has = goal_now & live[:, None]
        fid = torch.where(f < 0, torch.full_like(f, 4), f)
        fs.scatter_add_(1, torch.where(has, fid, torch.zeros_like(fid)),
                        has.to(self._dtype))
        fs = (fs > 0).to(self._dtype)
        self._fingering_state.copy_(torch.where(live[:, None], fs, self._fingering_state))

    def get_observation(self, physics):
        """Enabled observables (:414-449), evaluated once per control step."""
        self._update_goal_state()
        self._update_fingering_state()
        return self._observation_dict(physics)

    def _observation_dict(self, physics):
        obs = {
            f"{self.right_hand.name}/joints_pos": physics.qpos[:, self._rh_jnt],
            f"{self.left_hand.name}/joints_pos": physics.qpos[:, self._lh_jnt],
            "piano/state": self.piano.normalized_state,
            "piano/sustain_state": self.piano.sustain_state,
            "goal": self._goal_state.reshape(self._E, -1),
        }
        if not self._disable_fingering_reward:
            obs["fingering"] = self._fingering_state
        self._add_optional_observables(physics, obs)
        return obs

    # -- optional observables ------------------------------------------------------------------
    # The reference's entities define more observables than the task enables
    # (models/hands/base.py:75-114, shadow_hand.py:390-432, models/piano/piano.py:286-336); a user
    # switches them on with `entity.observables.<name>.enabled = True`.  Here:
    # `task.enable_observable("rh_shadow_hand/joints_vel")`.
    _HAND_OBSERVABLES = ("joints_vel", "joints_pos_cos_sin", "actuators_force", "actuators_velocity",
                         "actuators_power", "fingertip_positions", "joints_torque", "fingertip_force")
    _PIANO_OBSERVABLES = ("joints_pos", "activation", "sustain_activation")

    def _present_hands(self):
        return [h for h in (self.right_hand, self.left_hand) if h is not None]

    def available_observables(self):
        names = [f"{h.name}/{o}" for h in self._present_hands() for o in self._HAND_OBSERVABLES]
        return names + [f"piano/{o}" for o in self._PIANO_OBSERVABLES]

    def enable_observable(self, name: str, enabled: bool = True) -> None:
        if name not in self.available_observables():
            raise KeyError(f"Unknown observable {name!r}; optional observables: {self.available_observables()}")
        extra = getattr(self, "_extra_observables", [])
        if enabled and name not in extra:
            extra = extra + [name]
        if not enabled:
            extra = [n for n in extra if n != name]
        self._extra_observables = extra
        # the acceleration-stage sensors are switched on here, not at the first read (which would allocate and
        # synchronise in the middle of a step, and return readings the sensor stage never filled)
        if enabled and name.split("/")[1] in ("joints_torque", "fingertip_force") and getattr(self, "_physics", None) is not None:
            self._physics.enable_acc_sensors()

    def _optional_observable(self, physics, name):
        owner, what = name.split("/")
        if owner == "piano":
            if what == "joints_pos":
                return physics.qpos[:, self.piano._jidx]
            act = self.piano.activation if what == "activation" else self.piano.sustain_activation
            return act.to(self._dtype)
        hand = next(h for h in self._present_hands() if h.name == owner)
        dev = self._physics_device
        jnt = torch.as_tensor(hand.joints, device=dev, dtype=torch.long)
        act = torch.as_tensor(hand.actuators, device=dev, dtype=torch.long)
        if what == "joints_vel":
            return physics.qvel[:, jnt]
        if what == "joints_pos_cos_sin":
            q = physics.qpos[:, jnt]
            return torch.cat([torch.cos(q), torch.sin(q)], dim=1)
        if what == "actuators_force":
            return physics.act_force[:, act]
        if what == "actuators_velocity":
            return physics.act_vel[:, act]
        if what == "actuators_power":
            return physics.act_force[:, act].abs() * physics.act_vel[:, act].abs()
        if what == "joints_torque":
            # torque sensors at each joint's body origin projected on the joint axis
            # (hands/base.py:101-109); the engine's sensor stage delivers the projection
            physics.enable_acc_sensors()
            return physics.sens_torque[:, jnt]
        if what == "fingertip_force":
            # touch sensors at the fingertips (shadow_hand.py:425-432)
            return physics.site_touch(list(hand.fingertip_sites))
        return physics.site_xpos(list(hand.fingertip_sites)).reshape(self._E, -1)  # fingertip_positions

    def _add_optional_observables(self, physics, obs) -> None:
        for name in getattr(self, "_extra_observables", ()):
            obs[name] = self._optional_observable(physics, name)

    def observation_spec(self):
        L = self._n_steps_lookahead
        d = np.float64
        out = {
            f"{self.right_hand.name}/joints_pos": specs.Array((len(self.right_hand.joints),), d),
            f"{self.left_hand.name}/joints_pos": specs.Array((len(self.left_hand.joints),), d),
            "piano/state": specs.Array((88,), d),
            "piano/sustain_state": specs.Array((1,), d),
            "goal": specs.Array(((L + 1) * 89,), d),
        }
        if not self._disable_fingering_reward:
            out["fingering"] = specs.Array((10,), d)
        self._add_optional_specs(out)
        return out

    def _add_optional_specs(self, out) -> None:
        for name in getattr(self, "_extra_observables", ()):
            owner, what = name.split("/")
            if owner == "piano":
                n = 88 if what != "sustain_activation" else 1
            else:
                hand = next(h for h in self._present_hands() if h.name == owner)
                n = {"joints_vel": len(hand.joints), "joints_pos_cos_sin": 2 * len(hand.joints),
                     "fingertip_positions": 15, "joints_torque": len(hand.joints),
                     "fingertip_force": 5}.get(what, len(hand.actuators))
            out[name] = specs.Array((n,), np.float64)

    # -- rewards ------------------------------------------------------------------------------
    def _compute_forearm_reward(self, physics):
        """:251-259 — 0.5 unless the two forearms' geoms are in contact."""
        cg = physics.contact_geoms  # [E, C, 2]
        a, b = cg[..., 0], cg[..., 1]
        in_r = lambda x: (x[..., None] == self._rfa).any(-1)
        in_l = lambda x: (x[..., None] == self._lfa).any(-1)
        hit = ((in_r(a) & in_l(b)) | (in_l(a) & in_r(b))).any(dim=1)
        return torch.where(hit, 0.0, 0.5).to(self._dtype)

    def _compute_sustain_reward(self, physics):
        """:261-269."""
        return tolerance(self._goal_current[:, -1] - self.piano.sustain_activation[:, 0].to(self._dtype),
                         bounds=(0, _KEY_CLOSE_ENOUGH_TO_PRESSED),
                         margin=_KEY_CLOSE_ENOUGH_TO_PRESSED * 10)

    def _compute_energy_reward(self, physics):
        """:271-277 with actuators_power = |actuatorfrc| * |actuatorvel|
        (shadow_hand.py:407-416)."""
        power = physics.act_force.abs() * physics.act_vel.abs()
        rew = -self._energy_penalty_coef * (power[:, self._rh_act].sum(1) + power[:, self._lh_act].sum(1))
        return rew

    def _compute_key_press_reward(self, physics):
        """:279-298."""
        goal = self._goal_current[:, :-1]
        on = goal > 0
        actual = self.piano.state / self.piano._qpos_range[:, 1]
        rews = tolerance(goal - actual, bounds=(0, _KEY_CLOSE_ENOUGH_TO_PRESSED),
                         margin=_KEY_CLOSE_ENOUGH_TO_PRESSED * 10)
        n_on = on.sum(1)
        mean_on = (rews * on).sum(1) / torch.clamp(n_on, min=1)
        rew = torch.where(n_on > 0, 0.5 * mean_on, torch.zeros_like(mean_on))
        false_pos = (self.piano.activation & ~on).any(1)
        return rew + 0.5 * (1 - false_pos.to(self._dtype))

    def _key_targets(self, physics):
        """World position the fingertip should reach for every key: key geom centre
        + (0.35 size_x, 0, 0.5 size_z)  (:311-313)."""
        q = physics.qpos[:, self.piano._jidx]
        hx = self._key_half[:, 0]
        cx = self._key_anchor[:, 0] + hx * torch.cos(q)
        cz = self._key_anchor[:, 2] - hx * torch.sin(q)
        cy = self._key_anchor[:, 1].expand_as(cx)
        return torch.stack([cx + 0.35 * hx, cy, cz + 0.5 * self._key_half[:, 2]], dim=-1)

    def _compute_fingering_reward(self, physics):
        """:300-331."""
        f = self._finger_current  # [E, 88]; finger id of each goal key or -1
        has = self._goal_current[:, :-1] > 0
        fid = torch.where(f < 0, torch.full_like(f, 4), f)  # -1 -> right little finger
        tips = physics.site_xpos(self._tip_sites)  # [E, 10, 3]
        tgt = self._key_targets(physics)  # [E, 88, 3]
        tip_for_key = torch.gather(tips, 1, fid[..., None].expand(-1, -1, 3))
        dist = torch.linalg.norm(tgt - tip_for_key, dim=-1)
        rews = tolerance(dist, bounds=(0, _FINGER_CLOSE_ENOUGH_TO_KEY),
                         margin=_FINGER_CLOSE_ENOUGH_TO_KEY * 10)
        n = has.sum(1)
        mean = (rews * has).sum(1) / torch.clamp(n, min=1)
        return torch.where(n > 0, mean, torch.zeros_like(mean))

    def _compute_ot_fingering_reward(self, physics):
        """:333-369 — optimal assignment of the 10 fingertips (left first) to the keys to
        press.  This is the DEFINITION (scipy's solver per env on the host, as the reference); on the
        HIP engine the term runs inside the fused task kernels (include/rp_task.h, use_fingering = 2)
        and this function is the cross-check of tests/test_gpu_env.py."""
        from scipy.optimize import linear_sum_assignment
        tips = torch.cat([physics.site_xpos(list(self.left_hand.fingertip_sites)),
                          physics.site_xpos(list(self.right_hand.fingertip_sites))], dim=1)
        tips = tips.detach().cpu().numpy().astype(np.float64)
        tgt = self._key_targets(physics).detach().cpu().numpy().astype(np.float64)
        goal = (self._goal_current[:, :-1] > 0).detach().cpu().numpy()
        out = np.ones(self._E)
        s = np.sqrt(-2 * np.log(0.1))
        for e in range(self._E):
            keys = np.flatnonzero(goal[e])
            if keys.size == 0:
                continue
            dist = np.linalg.norm(tips[e][:, None, :] - tgt[e][None, keys, :], axis=-1)
            r, c = linear_sum_assignment(dist)
            d = dist[r, c]
            m = _FINGER_CLOSE_ENOUGH_TO_KEY
            rew = np.where(d <= m, 1.0, np.exp(-0.5 * (((d - m) / (m * 10)) * s) ** 2))
            out[e] = rew.mean()
        return torch.as_tensor(out, device=self._physics_device, dtype=self._dtype)

    # -- misc -------------------------------------------------------------------------------------
    def _randomize_initial_hand_positions(self, physics, mask=None) -> None:
        """:491-499 — one offset per env, applied to both hands along world y."""
        if not self._randomize_hand_positions:
            return
        E = self._E
        if not hasattr(self, "_tree_offset"):
            self._tree_offset = torch.zeros((E, 2, 3), device=self._physics_device, dtype=self._dtype)
        off = self._random_state.uniform(low=-_POSITION_OFFSET, high=_POSITION_OFFSET, size=E)
        off = torch.as_tensor(off, device=self._physics_device, dtype=self._dtype)
        sel = torch.ones(E, dtype=torch.bool, device=off.device) if mask is None else mask
        # shift_pose is cumulative in the reference (it edits body_pos in place)
        self._tree_offset[:, :, 1] = torch.where(sel[:, None], self._tree_offset[:, :, 1] + off[:, None],
                                                 self._tree_offset[:, :, 1])
        physics.set_tree_offset(self._tree_offset)
