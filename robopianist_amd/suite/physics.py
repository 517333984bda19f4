"""Torch-side view of the HIP engine used by the vectorised tasks.

Plays the role of `physics.bind(...)` in the reference (call sites: SURVEY.md
§8b).  Every array is a zero-copy torch tensor aliasing engine memory
(`rp_field_ptr`), and the engine enqueues on torch's current stream
(`rp_set_stream`), so task code, rewards and the step kernel are ordered on one
HIP stream with no host synchronisation and no staging copies.
"""

from __future__ import annotations

import os

import torch

from robopianist_amd import engine as eng


class TorchPhysics:
    def __init__(self, scene_info, n_envs: int, device_id: int = 0, precision: int = 64):
        if not torch.cuda.is_available():
            raise eng.EngineError(
                "No HIP device visible: the batched engine has no CPU fallback.")
        self.scene = scene_info
        self.model = scene_info.model
        self.n_envs = n_envs
        self.device = torch.device("cuda", device_id)
        self.dtype = torch.float32 if precision == 32 else torch.float64
        self.engine = eng.BatchedPhysics(self.model, scene_info.key_joint_ids, n_envs,
                                         device_id=device_id, precision=precision)
        with torch.cuda.device(self.device):
            self._stream = torch.cuda.current_stream(self.device)
            self.engine.set_stream(self._stream.cuda_stream)
        e = self.engine
        self.qpos = e.view(eng.QPOS)
        self.qvel = e.view(eng.QVEL)
        self.act_force = e.view(eng.ACT_FORCE)
        self.act_vel = e.view(eng.ACT_VELOCITY)
        self.site_xpos_eng = e.view(eng.SITE_XPOS)
        self.contact_geoms = e.view(eng.CONTACT_GEOMS)
        self.time = e.view(eng.TIME)
        self.warn = e.view(eng.WARN_FLAGS)
        self._ctrl = e.view(eng.CTRL)
        self._qfrc_applied = e.view(eng.QFRC_APPLIED)
        self._tree_offset = e.view(eng.TREE_OFFSET) if e.ntree else None
        self._active = e.view(eng.ACTIVE)
        self._active.fill_(1)
        # The env layer changes qpos / qvel only through reset() and load_state_dict(), both
        # followed by forward(): rp_step may skip its leading position stage for untouched envs.
        # Code that writes `physics.qpos` / `qvel` through the views must call forward() itself.
        e.set_lazy_position_stage(True)
        # heterogeneous batches (random policies, envs at their own episode times): heaviest envs first
        e.set_cost_ordered_launch(os.environ.get("RP_COST_ORDER", "1") != "0")
        # two halves of the batch on two streams (one half's launch tail overlaps the other half's next
        # kernel) when the engine measures that to be faster: heterogeneous batches
        e.set_stream_slices(int(os.environ.get("RP_STREAM_SLICES", "0")))
        from robopianist_amd.model import engine_tables
        t = engine_tables.build_engine_tables(self.model, scene_info.key_joint_ids)
        self._site_modelid = {int(s): i for i, s in enumerate(t["eng_site_modelid"])}
        self._site_index_cache = {}
        self.timestep = float(self.model.opt_timestep)

    # -- acceleration-stage sensors ------------------------------------------------
    def enable_acc_sensors(self):
        """Turns on the `torque` / `touch` sensors (one extra engine launch per step) and maps their
        arrays: `sens_torque` [E, nv] (joint-axis projection at every hand joint = joints_torque),
        `sens_touch` [E, n_engine_sites] (fingertip_force at the fingertip sites)."""
        if getattr(self, "sens_torque", None) is None:
            self.engine.set_acc_sensors(True)
            self.sens_torque = self.engine.view(eng.SENSOR_TORQUE)
            self.sens_touch = self.engine.view(eng.SENSOR_TOUCH)

    def site_touch(self, model_site_ids):
        """Touch sensor readings of the given model sites, [E, len(ids)]."""
        self.enable_acc_sensors()
        idx = torch.as_tensor([self._site_modelid[int(s)] for s in model_site_ids], dtype=torch.long, device=self.device)
        return self.sens_touch.index_select(1, idx)

    # -- reads -----------------------------------------------------------------
    def refresh(self):
        """Views alias engine memory; nothing to copy."""

    def site_xpos(self, model_site_ids):
        key = tuple(int(s) for s in model_site_ids)
        idx = self._site_index_cache.get(key)
        if idx is None:  # index tensors live on the device: no host->device copy per step
            idx = torch.as_tensor([self._site_modelid[s] for s in key], dtype=torch.long, device=self.device)
            self._site_index_cache[key] = idx
        return self.site_xpos_eng.index_select(1, idx)

    # -- writes ----------------------------------------------------------------
    @property
    def ctrl(self):
        return self._ctrl

    def set_ctrl(self, ctrl: torch.Tensor):
        self._ctrl.copy_(ctrl)

    def set_qfrc_applied(self, f):
        self._qfrc_applied.copy_(torch.as_tensor(f, dtype=self.dtype, device=self.device)
                                 .expand(self.n_envs, self.model.nv))

    def set_tree_offset(self, off):
        self._tree_offset.copy_(torch.as_tensor(off, dtype=self.dtype, device=self.device))

    @property
    def active_mask(self):
        """The engine's RP_ACTIVE array (int32 [n_envs], zero-copy)."""
        return self._active

    def set_active(self, mask: torch.Tensor):
        self._active.copy_(mask)   # (bool -> int32 in the copy kernel)

    # -- checkpoint / resume -----------------------------------------------------
    _STATE_FIELDS = ("qpos", "qvel", "qacc_warmstart", "ctrl", "qfrc_applied", "time", "tree_offset")

    def _state_views(self):
        if not hasattr(self, "_warm"):
            self._warm = self.engine.view(eng.QACC_WARMSTART)
        v = {"qpos": self.qpos, "qvel": self.qvel, "qacc_warmstart": self._warm, "ctrl": self._ctrl,
             "qfrc_applied": self._qfrc_applied, "time": self.time}
        if self._tree_offset is not None:
            v["tree_offset"] = self._tree_offset
        return v

    def state_dict(self):
        """Everything mj_step reads: qpos, qvel, qacc_warmstart, ctrl, qfrc_applied, time (and the
        per-env hand offsets).  Derived arrays (contacts, site positions, actuator velocities) are
        recomputed by `load_state_dict` with physics.forward()."""
        return {k: v.detach().clone() for k, v in self._state_views().items()}

    def load_state_dict(self, sd):
        views = self._state_views()
        for k, v in views.items():
            v.copy_(sd[k].to(device=self.device, dtype=v.dtype))
        self._active.fill_(1)
        self.engine.forward()

    # -- stepping --------------------------------------------------------------
    def reset(self, mask=None):
        if mask is not None:
            mask = mask.to(torch.uint8).contiguous()
            self._reset_mask = mask  # keep alive until the kernel has run
        self.engine.reset(mask)

    def forward(self):
        self.engine.forward()

    def step(self, n_substeps: int, key_trace=None):
        self.engine.step(n_substeps, key_trace)

    def step_masked(self, n_substeps: int, key_trace, reset_mask):
        """One C call: reset + forward of the flagged envs, step of the active ones (rp_step_masked)."""
        self._reset_mask = reset_mask  # keep alive until the kernels have run
        self.engine.step_masked(n_substeps, key_trace, reset_mask)
