"""ctypes binding of include/rp_task.h (fused task-layer kernels in librp_engine.so).

Used by suite/tasks/piano_with_shadow_hands.py when the physics is the HIP engine: the
five reward terms of every env are evaluated by one launch on torch's current stream
instead of ~130 elementwise torch kernels.  The torch implementation stays as the
definition of the semantics (CPU tests with FakePhysics, GPU cross-check test)."""

from __future__ import annotations

import ctypes

import torch

from robopianist_amd.suite.scripted import ScriptedActions  # noqa: F401  (the pre-step launch's table source)

from robopianist_amd import engine

EXPORTED_SYMBOLS = ("rp_task_rewards", "rp_task_advance", "rp_task_prestep", "rp_task_rasterize", "rp_task_last_error")
TERM_NAMES = ("key_press_reward", "sustain_reward", "energy_reward", "fingering_reward", "forearm_reward")


class RewardArgs(ctypes.Structure):
    _fields_ = [
        ("n_envs", ctypes.c_int), ("precision", ctypes.c_int),
        ("nv", ctypes.c_int), ("nu", ctypes.c_int), ("n_sites", ctypes.c_int), ("n_contacts", ctypes.c_int),
        ("use_fingering", ctypes.c_int), ("use_forearm", ctypes.c_int),
        ("energy_coef", ctypes.c_double), ("key_close", ctypes.c_double), ("finger_close", ctypes.c_double),
        ("qpos", ctypes.c_void_p), ("act_force", ctypes.c_void_p), ("act_vel", ctypes.c_void_p),
        ("site_xpos", ctypes.c_void_p), ("contact_geoms", ctypes.c_void_p),
        ("goal_current", ctypes.c_void_p), ("key_norm_state", ctypes.c_void_p),
        ("key_activation", ctypes.c_void_p), ("sustain_activation", ctypes.c_void_p),
        ("finger_current", ctypes.c_void_p),
        ("key_qadr", ctypes.c_void_p), ("key_anchor", ctypes.c_void_p), ("key_half", ctypes.c_void_p),
        ("hand_act", ctypes.c_void_p), ("n_hand_act", ctypes.c_int),
        ("tip_site", ctypes.c_void_p),
        ("rfa", ctypes.c_void_p), ("n_rfa", ctypes.c_int),
        ("lfa", ctypes.c_void_p), ("n_lfa", ctypes.c_int),
        ("terms", ctypes.c_void_p), ("total", ctypes.c_void_p),
        ("hand_filter", ctypes.c_int),
    ]


class AdvanceArgs(ctypes.Structure):
    _fields_ = [
        ("rw", RewardArgs),
        ("n_lookahead", ctypes.c_int), ("n_songs", ctypes.c_int), ("bank_len", ctypes.c_int),
        ("wrong_press_termination", ctypes.c_int),
        ("key_threshold", ctypes.c_double), ("sustain_threshold", ctypes.c_double),
        ("warn", ctypes.c_void_p), ("key_qrange", ctypes.c_void_p),
        ("goal_bank", ctypes.c_void_p), ("finger_bank", ctypes.c_void_p),
        ("song_len", ctypes.c_void_p), ("song_id", ctypes.c_void_p),
        ("key_state", ctypes.c_void_p), ("sustain_state", ctypes.c_void_p),
        ("t_idx", ctypes.c_void_p), ("should_terminate", ctypes.c_void_p), ("failure_termination", ctypes.c_void_p),
        ("discount_state", ctypes.c_void_p), ("goal_state", ctypes.c_void_p), ("finger_next", ctypes.c_void_p),
        ("fingering_state", ctypes.c_void_p), ("needs_reset", ctypes.c_void_p),
        ("discount", ctypes.c_void_p), ("step_type", ctypes.c_void_p),
        ("eval_sums", ctypes.c_void_p), ("eval_count", ctypes.c_void_p), ("eval_hist", ctypes.c_void_p),
        ("eval_nfinished", ctypes.c_void_p), ("eval_deque", ctypes.c_int),
        ("next_ready", ctypes.c_void_p), ("consumed", ctypes.c_void_p),
        ("warn_fatal_mask", ctypes.c_int), ("fatal_count", ctypes.c_void_p),
        ("warn_count_mask", ctypes.c_int), ("warn_count", ctypes.c_void_p),
        ("traj_record", ctypes.c_void_p),
    ]


class PrestepArgs(ctypes.Structure):
    _fields_ = [
        ("n_envs", ctypes.c_int), ("precision", ctypes.c_int), ("n_action", ctypes.c_int), ("nu", ctypes.c_int),
        ("action", ctypes.c_void_p), ("act_lo", ctypes.c_void_p), ("act_range", ctypes.c_void_p),
        ("clip", ctypes.c_int),
        ("needs_reset", ctypes.c_void_p), ("hand_act", ctypes.c_void_p),
        ("ctrl", ctypes.c_void_p), ("sustain_state", ctypes.c_void_p),
        ("active", ctypes.c_void_p), ("reset_mask", ctypes.c_void_p),
        ("action_table", ctypes.c_void_p), ("action_index", ctypes.c_void_p), ("action_table_len", ctypes.c_int),
    ]


class RasterArgs(ctypes.Structure):
    _fields_ = [
        ("n_jobs", ctypes.c_int), ("precision", ctypes.c_int),
        ("n_songs", ctypes.c_int), ("bank_len", ctypes.c_int), ("max_ops", ctypes.c_int), ("n_buffer", ctypes.c_int),
        ("fps", ctypes.c_double),
        ("note_ofs", ctypes.c_void_p), ("note_start", ctypes.c_void_p), ("note_end", ctypes.c_void_p),
        ("note_pitch", ctypes.c_void_p), ("note_velocity", ctypes.c_void_p), ("note_part", ctypes.c_void_p),
        ("cc_ofs", ctypes.c_void_p), ("cc_time", ctypes.c_void_p), ("cc_value", ctypes.c_void_p),
        ("total_time", ctypes.c_void_p),
        ("job_slot", ctypes.c_void_p), ("job_song", ctypes.c_void_p), ("op_kind", ctypes.c_void_p),
        ("op_value", ctypes.c_void_p),
        ("goal_bank", ctypes.c_void_p), ("finger_bank", ctypes.c_void_p), ("song_len", ctypes.c_void_p),
        ("status", ctypes.c_void_p),
    ]


def _lib():
    L = engine.load_library()
    if not getattr(L, "_rp_task_ready", False):
        L.rp_task_rewards.argtypes = [ctypes.POINTER(RewardArgs), ctypes.c_void_p]
        L.rp_task_rewards.restype = ctypes.c_int
        L.rp_task_advance.argtypes = [ctypes.POINTER(AdvanceArgs), ctypes.c_void_p]
        L.rp_task_advance.restype = ctypes.c_int
        L.rp_task_prestep.argtypes = [ctypes.POINTER(PrestepArgs), ctypes.c_void_p]
        L.rp_task_prestep.restype = ctypes.c_int
        L.rp_task_rasterize.argtypes = [ctypes.POINTER(RasterArgs), ctypes.c_void_p]
        L.rp_task_rasterize.restype = ctypes.c_int
        L.rp_task_last_error.restype = ctypes.c_char_p
        L._rp_task_ready = True
    return L


def _chk(t: torch.Tensor, dtype, shape):
    if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous() or not t.is_cuda:
        raise engine.EngineError(f"fused task kernel: bad array {tuple(t.shape)} {t.dtype} (want {shape} {dtype})")
    return t.data_ptr()


class FusedRewards:
    """All reward terms of PianoWithShadowHands for every env in one launch."""

    def __init__(self, physics, *, n_envs, key_qadr, key_anchor, key_half, hand_act, tip_site, rfa, lfa,
                 use_fingering, use_forearm, energy_coef, key_close, finger_close, hand_filter=0):
        self._L = _lib()
        dev, dt = physics.device, physics.dtype
        self._phys, self._E, self._dt = physics, int(n_envs), dt
        i32 = lambda v: torch.as_tensor([int(x) for x in v], dtype=torch.int32, device=dev).contiguous()
        # device-resident constants (kept alive by this object)
        self._key_qadr, self._hand_act, self._tip_site = i32(key_qadr), i32(hand_act), i32(tip_site)
        self._rfa, self._lfa = i32(rfa), i32(lfa)
        self._key_anchor = key_anchor.to(device=dev, dtype=dt).contiguous()
        self._key_half = key_half.to(device=dev, dtype=dt).contiguous()
        self.terms = torch.zeros((5, self._E), device=dev, dtype=dt)
        self.total = torch.zeros((self._E,), device=dev, dtype=dt)
        a = RewardArgs()
        a.n_envs, a.precision = self._E, 64 if dt == torch.float64 else 32
        a.nv, a.nu = int(physics.qpos.shape[1]), int(physics.act_force.shape[1])
        a.n_sites, a.n_contacts = int(physics.site_xpos_eng.shape[1]), int(physics.contact_geoms.shape[1])
        # use_fingering: False / True, or 2 = the optimal-transport term (include/rp_task.h)
        a.use_fingering, a.use_forearm = int(use_fingering), int(bool(use_forearm))
        a.energy_coef, a.key_close, a.finger_close = float(energy_coef), float(key_close), float(finger_close)
        E = self._E
        a.qpos = _chk(physics.qpos, dt, (E, a.nv))
        a.act_force = _chk(physics.act_force, dt, (E, a.nu))
        a.act_vel = _chk(physics.act_vel, dt, (E, a.nu))
        a.site_xpos = _chk(physics.site_xpos_eng, dt, (E, a.n_sites, 3))
        a.contact_geoms = _chk(physics.contact_geoms, torch.int32, (E, a.n_contacts, 2))
        a.key_qadr, a.key_anchor, a.key_half = self._key_qadr.data_ptr(), self._key_anchor.data_ptr(), self._key_half.data_ptr()
        a.hand_act, a.n_hand_act = self._hand_act.data_ptr(), int(self._hand_act.numel())
        a.tip_site = self._tip_site.data_ptr()
        a.rfa, a.n_rfa = self._rfa.data_ptr(), int(self._rfa.numel())
        a.lfa, a.n_lfa = self._lfa.data_ptr(), int(self._lfa.numel())
        a.terms, a.total = self.terms.data_ptr(), self.total.data_ptr()
        a.hand_filter = int(hand_filter)
        if len(tip_site) != (5 if a.hand_filter else 10):
            raise engine.EngineError("fused task kernel: tip_site must list 10 fingertips (5 for one hand)")
        self._args = a

    def compute(self, *, goal_current, key_norm_state, key_activation, sustain_activation, finger_current):
        """Enqueues the launch on torch's current stream; returns (total [E], terms [5][E]).
        The state tensors are the task's persistent (in-place updated) buffers."""
        a, E = self._args, self._E
        a.goal_current = _chk(goal_current, self._dt, (E, 89))
        a.key_norm_state = _chk(key_norm_state, self._dt, (E, 88))
        a.key_activation = _chk(key_activation, torch.bool, (E, 88))
        a.sustain_activation = _chk(sustain_activation, torch.bool, (E, 1))
        a.finger_current = _chk(finger_current, torch.int64, (E, 88))
        with torch.cuda.device(self._phys.device):  # the launch must see the stream's device as current
            stream = torch.cuda.current_stream(self._phys.device).cuda_stream
            rc = self._L.rp_task_rewards(ctypes.byref(a), ctypes.c_void_p(stream))
        if rc != 0:
            raise engine.EngineError(self._L.rp_task_last_error().decode())
        return self.total, self.terms


class FusedAdvance:
    """rp_task_advance: key state, after_step, goal/fingering observables, rewards,
    termination, discount and step types of every env in one launch.  All state tensors
    are the task's persistent buffers and are updated in place."""

    def __init__(self, rewards: FusedRewards, *, n_lookahead, goal_bank, finger_bank, song_len, song_id,
                 wrong_press_termination, key_threshold, sustain_threshold, key_qrange, warn_fatal_mask=1,
                 warn_count_mask=0):
        self._L = _lib()
        self._rw = rewards
        E, dt, dev = rewards._E, rewards._dt, rewards._phys.device
        self._E, self._dt = E, dt
        self._goal_bank = goal_bank.to(device=dev, dtype=dt).contiguous()
        self._finger_bank = finger_bank.to(device=dev, dtype=torch.int64).contiguous()
        self._song_len = song_len.to(device=dev, dtype=torch.int64).contiguous()
        self._song_id = song_id.to(device=dev, dtype=torch.int64).contiguous()
        self._key_qrange = key_qrange.to(device=dev, dtype=dt).contiguous()
        self.discount = torch.zeros((E,), device=dev, dtype=dt)
        self.step_type = torch.zeros((E,), device=dev, dtype=torch.int32)
        p = AdvanceArgs()
        p.n_lookahead = int(n_lookahead)
        p.n_songs, p.bank_len = int(self._goal_bank.shape[0]), int(self._goal_bank.shape[1])
        p.wrong_press_termination = int(bool(wrong_press_termination))
        p.key_threshold, p.sustain_threshold = float(key_threshold), float(sustain_threshold)
        p.warn = _chk(rewards._phys.warn, torch.int32, (E,))
        p.key_qrange = self._key_qrange.data_ptr()
        p.goal_bank, p.finger_bank = self._goal_bank.data_ptr(), self._finger_bank.data_ptr()
        p.song_len, p.song_id = self._song_len.data_ptr(), self._song_id.data_ptr()
        p.discount, p.step_type = self.discount.data_ptr(), self.step_type.data_ptr()
        p.warn_fatal_mask = int(warn_fatal_mask)
        self.fatal_count = torch.zeros((E,), device=dev, dtype=torch.int64)
        p.fatal_count = self.fatal_count.data_ptr()
        p.warn_count_mask = int(warn_count_mask)
        self.warn_count = torch.zeros((E,), device=dev, dtype=torch.int64)
        p.warn_count = self.warn_count.data_ptr()
        self._p = p
        self._L_lookahead = int(n_lookahead)

    def enable_trajectory_record(self, n_buffers: int = 2):
        """The multi-GPU gather's per-env record (include/rp_task.h `traj_record`; distributed.pack_trajectory_record's
        layout) written by every launch from now on, into `n_buffers` preallocated buffers in turn (two: the
        asynchronous all-gather of step t may still read its buffer while step t + 1 writes the other).
        `trajectory_record` is the buffer the last launch filled."""
        E, dt, dev = self._E, self._dt, self._rw._phys.device
        width = int(self._rw._args.nv) + 3 + (2 if dt == torch.float64 else 3)
        self._traj = [torch.zeros((E, width), device=dev, dtype=dt) for _ in range(max(1, int(n_buffers)))]
        self._traj_i = -1

    @property
    def trajectory_record(self):
        return None if getattr(self, "_traj", None) is None or self._traj_i < 0 else self._traj[self._traj_i % len(self._traj)]

    def set_prefetch_buffers(self, next_ready, consumed):
        """Double-buffered goal bank (include/rp_task.h `next_ready` / `consumed`); None: off."""
        p, E = self._p, self._E
        if next_ready is None:
            p.next_ready = p.consumed = None
            self._prefetch = None
            return
        if int(self._goal_bank.shape[0]) != 2 * E:
            raise engine.EngineError("prefetch mode needs two bank slots per env")
        p.next_ready = _chk(next_ready, torch.uint8, (E,))
        p.consumed = _chk(consumed, torch.uint8, (E,))
        self._prefetch = (next_ready, consumed)

    def set_evaluation_buffers(self, sums, count, hist, n_finished):
        """Turns on the MidiEvaluationWrapper reduction inside the launch (None: off)."""
        p, E = self._p, self._E
        if sums is None:
            p.eval_sums = p.eval_count = p.eval_hist = p.eval_nfinished = None
            p.eval_deque = 0
            self._eval = None
            return
        D = int(hist.shape[1])
        p.eval_sums = _chk(sums, torch.float64, (E, 6))
        p.eval_count = _chk(count, torch.float64, (E,))
        p.eval_hist = _chk(hist, torch.float64, (E, D, 6))
        p.eval_nfinished = _chk(n_finished, torch.int64, (E,))
        p.eval_deque = D
        self._eval = (sums, count, hist, n_finished)  # keep alive

    def advance(self, *, needs_reset, key_state, key_norm_state, key_activation, sustain_state, sustain_activation,
                t_idx, should_terminate, failure_termination, discount_state, goal_state, goal_current,
                finger_next, finger_current, fingering_state):
        E, dt, p, L = self._E, self._dt, self._p, self._L_lookahead
        a = self._rw._args
        a.goal_current = _chk(goal_current, dt, (E, 89))
        a.key_norm_state = _chk(key_norm_state, dt, (E, 88))
        a.key_activation = _chk(key_activation, torch.bool, (E, 88))
        a.sustain_activation = _chk(sustain_activation, torch.bool, (E, 1))
        a.finger_current = _chk(finger_current, torch.int64, (E, 88))
        p.rw = a
        p.key_state = _chk(key_state, dt, (E, 88))
        p.sustain_state = _chk(sustain_state, dt, (E, 1))
        p.t_idx = _chk(t_idx, torch.int64, (E,))
        p.should_terminate = _chk(should_terminate, torch.bool, (E,))
        p.failure_termination = _chk(failure_termination, torch.bool, (E,))
        p.discount_state = _chk(discount_state, dt, (E,))
        p.goal_state = _chk(goal_state, dt, (E, L + 1, 89))
        p.finger_next = _chk(finger_next, torch.int64, (E, 88))
        p.fingering_state = _chk(fingering_state, dt, (E, 5 if a.hand_filter else 10))
        p.needs_reset = _chk(needs_reset, torch.bool, (E,))
        if getattr(self, "_traj", None) is not None:
            # A captured graph bakes ONE record pointer in: replays would all write the buffer of the captured call while
            # this counter stands still, and the "step t's gather may still read while step t + 1 writes" guarantee of
            # two buffers would be void.  Under capture only the single-buffer mode is allowed; its
            # contract is explicit: the consumer finishes with the record before the next replay.
            if len(self._traj) > 1 and torch.cuda.is_current_stream_capturing():
                raise engine.EngineError(
                    "trajectory record with alternating buffers inside a stream capture: a replay always writes the "
                    "captured buffer.  Call enable_trajectory_record(n_buffers=1) and wait for the gather of step t "
                    "before replaying step t + 1")
            self._traj_i += 1
            p.traj_record = self._traj[self._traj_i % len(self._traj)].data_ptr()
        else:
            p.traj_record = None
        with torch.cuda.device(self._rw._phys.device):
            stream = torch.cuda.current_stream(self._rw._phys.device).cuda_stream
            rc = self._L.rp_task_advance(ctypes.byref(p), ctypes.c_void_p(stream))
        if rc != 0:
            raise engine.EngineError(self._L.rp_task_last_error().decode())
        return self.step_type, self._rw.total, self.discount, self._rw.terms


class FusedPrestep:
    """rp_task_prestep: canonical action -> spec bounds, reset bookkeeping (active / reset masks), hand actions ->
    actuator ctrl, sustain -> the piano's latch: one launch before the physics (include/rp_task.h)."""

    def __init__(self, physics, *, n_envs, n_action, hand_act, sustain_state):
        self._L = _lib()
        dev, dt = physics.device, physics.dtype
        self._phys, self._E, self._dt = physics, int(n_envs), dt
        self._hand_act = torch.as_tensor([int(x) for x in hand_act], dtype=torch.int32, device=dev).contiguous()
        if int(self._hand_act.numel()) != int(n_action) - 1:
            raise engine.EngineError("fused pre-step kernel: one actuator per hand action expected")
        self.reset_mask = torch.zeros((self._E,), device=dev, dtype=torch.uint8)
        a = PrestepArgs()
        a.n_envs, a.precision = self._E, 64 if dt == torch.float64 else 32
        a.n_action, a.nu = int(n_action), int(physics.ctrl.shape[1])
        a.hand_act = self._hand_act.data_ptr()
        a.ctrl = _chk(physics.ctrl, dt, (self._E, a.nu))
        a.sustain_state = _chk(sustain_state, dt, (self._E, 1))
        a.active = _chk(physics.active_mask, torch.int32, (self._E,))
        a.reset_mask = self.reset_mask.data_ptr()
        self._keep = (sustain_state,)
        self._args = a
        self._bounds = None

    def run(self, action, needs_reset, bounds=None, clip=False):
        """action [E, n_action] (device, the engine's dtype); bounds = (lo, hi - lo) device tensors for a canonical
        action in [-1, 1], None for an action in the spec's units.  Returns the reset mask for step_masked."""
        a, E = self._args, self._E
        if isinstance(action, ScriptedActions):
            # (include/rp_task.h: the launch takes every env's own row of the table and advances the index itself)
            a.action = None
            a.action_table = _chk(action.table, self._dt, (int(action.table.shape[0]), a.n_action))
            a.action_index = _chk(action.index, torch.int64, (E,))
            a.action_table_len = int(action.table.shape[0])
        else:
            a.action = _chk(action, self._dt, (E, a.n_action))
            a.action_table = a.action_index = None
            a.action_table_len = 0
        a.needs_reset = _chk(needs_reset, torch.bool, (E,))
        if bounds is None:
            a.act_lo = a.act_range = None
        else:
            a.act_lo = _chk(bounds[0], self._dt, (a.n_action,)); a.act_range = _chk(bounds[1], self._dt, (a.n_action,))
        a.clip = int(bool(clip))
        self._bounds = bounds
        with torch.cuda.device(self._phys.device):
            stream = torch.cuda.current_stream(self._phys.device).cuda_stream
            rc = self._L.rp_task_prestep(ctypes.byref(a), ctypes.c_void_p(stream))
        if rc != 0:
            raise engine.EngineError(self._L.rp_task_last_error().decode())
        return self.reset_mask


class Rasterizer:
    """rp_task_rasterize: goal / fingering tables of stretched / transposed songs built on the
    device.  `songs` are music.midi_file.NoteArrays of the base songs (pitches on the piano,
    velocities <= 127: the caller checks); a job = (bank slot, base song, ordered ops)."""

    MAX_OPS = 8
    KIND = {"stretch": 1, "transpose": 2}

    def __init__(self, device, dtype, songs, control_timestep, initial_buffer_time):
        import numpy as np
        self._L = _lib()
        self._dev, self._dt = device, dtype
        order = [np.argsort(a.start, kind="stable") for a in songs]
        cat = lambda xs, dt: torch.as_tensor(np.concatenate(xs) if xs else np.zeros(0), dtype=dt, device=device).contiguous()
        ofs = lambda ns: torch.as_tensor(np.concatenate([[0], np.cumsum(ns)]), dtype=torch.int64, device=device)
        self._note_ofs = ofs([len(a.start) for a in songs])
        self._start = cat([a.start[o] for a, o in zip(songs, order)], torch.float64)
        self._end = cat([a.end[o] for a, o in zip(songs, order)], torch.float64)
        self._pitch = cat([a.pitch[o] for a, o in zip(songs, order)], torch.int32)
        self._vel = cat([a.velocity[o] for a, o in zip(songs, order)], torch.int32)
        self._part = cat([a.part[o] for a, o in zip(songs, order)], torch.int32)
        sus = [a.cc_num == 64 for a in songs]
        self._cc_ofs = ofs([int(m.sum()) for m in sus])
        self._cc_time = cat([a.cc_time[m] for a, m in zip(songs, sus)], torch.float64)
        self._cc_val = cat([a.cc_val[m] for a, m in zip(songs, sus)], torch.int32)
        self._total = torch.as_tensor([a.total_time for a in songs], dtype=torch.float64, device=device)
        p = RasterArgs()
        p.precision = 64 if dtype == torch.float64 else 32
        p.n_songs, p.max_ops = len(songs), self.MAX_OPS
        p.fps = 1 / control_timestep
        p.n_buffer = int(round(initial_buffer_time / control_timestep))
        p.note_ofs, p.note_start, p.note_end = self._note_ofs.data_ptr(), self._start.data_ptr(), self._end.data_ptr()
        p.note_pitch, p.note_velocity, p.note_part = self._pitch.data_ptr(), self._vel.data_ptr(), self._part.data_ptr()
        p.cc_ofs, p.cc_time, p.cc_value = self._cc_ofs.data_ptr(), self._cc_time.data_ptr(), self._cc_val.data_ptr()
        p.total_time = self._total.data_ptr()
        self._p = p
        self._keep = None

    def rasterize(self, goal_bank, finger_bank, song_len, slots, songs, ops):
        """Enqueues the launch on torch's current stream; returns the per-job status tensor
        (0 = done, 1 = does not fit in the bank's rows).  ops[j] = [("stretch", f) | ("transpose", k), ...]."""
        import numpy as np
        n = len(slots)
        kind = np.zeros((n, self.MAX_OPS), np.int32)
        val = np.zeros((n, self.MAX_OPS), np.float64)
        for j, seq in enumerate(ops):
            for o, (k, v) in enumerate(seq):
                kind[j, o] = self.KIND[k]; val[j, o] = v
        dev = self._dev
        t_slot = torch.as_tensor(np.asarray(slots, np.int64), device=dev)
        t_song = torch.as_tensor(np.asarray(songs, np.int32), device=dev)
        t_kind, t_val = torch.as_tensor(kind, device=dev), torch.as_tensor(val, device=dev)
        status = torch.full((n,), -1, dtype=torch.int32, device=dev)
        p = self._p
        p.n_jobs, p.bank_len = n, int(goal_bank.shape[1])
        p.job_slot, p.job_song = t_slot.data_ptr(), t_song.data_ptr()
        p.op_kind, p.op_value = t_kind.data_ptr(), t_val.data_ptr()
        S = int(goal_bank.shape[0])
        p.goal_bank = _chk(goal_bank, self._dt, (S, p.bank_len, 89))
        p.finger_bank = _chk(finger_bank, torch.int64, (S, p.bank_len, 88))
        p.song_len = _chk(song_len, torch.int64, (S,))
        p.status = status.data_ptr()
        with torch.cuda.device(dev):
            rc = self._L.rp_task_rasterize(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise engine.EngineError(self._L.rp_task_last_error().decode())
        self._keep = (t_slot, t_song, t_kind, t_val, status)  # alive until the launch has run
        return status
