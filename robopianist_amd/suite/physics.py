"""Torch-side view of the HIP engine used by the vectorised tasks.

Plays the role of `physics.bind(...)` in the reference (call sites: SURVEY.md
§8b): batched device tensors for qpos / sensors / site positions, refreshed
after every `step()`, plus writers for ctrl / qfrc_applied / body offsets.
Device tensors are handed to the C ABI by raw pointer (no host round trip).
"""

from __future__ import annotations

import numpy as np
import torch

from robopianist_amd import engine as eng


class TorchPhysics:
    def __init__(self, scene_info, n_envs: int, device_id: int = 0, precision: int = 32):
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
        E, m = n_envs, self.model
        f = dict(dtype=self.dtype, device=self.device)
        self.qpos = torch.zeros((E, m.nv), **f)
        self.qvel = torch.zeros((E, m.nv), **f)
        self.act_force = torch.zeros((E, m.nu), **f)
        self.act_vel = torch.zeros((E, m.nu), **f)
        self.site_xpos_eng = torch.zeros((E, self.engine.nsite, 3), **f)
        self.contact_geoms = torch.full((E, eng.MAX_CONTACTS, 2), -1, dtype=torch.int32,
                                        device=self.device)
        self.time = torch.zeros((E,), **f)
        self.warn = torch.zeros((E,), dtype=torch.int32, device=self.device)
        self._ctrl = torch.zeros((E, m.nu), **f)
        self._active = torch.ones((E,), dtype=torch.int32, device=self.device)
        # engine site order -> model site ids
        from robopianist_amd.model import engine_tables
        t = engine_tables.build_engine_tables(m, scene_info.key_joint_ids)
        self._site_modelid = {int(s): i for i, s in enumerate(t["eng_site_modelid"])}
        self.timestep = float(m.opt_timestep)

    # -- reads -----------------------------------------------------------------
    def refresh(self):
        e = self.engine
        e.get(eng.QPOS, self.qpos)
        e.get(eng.QVEL, self.qvel)
        e.get(eng.ACT_FORCE, self.act_force)
        e.get(eng.ACT_VELOCITY, self.act_vel)
        if e.nsite:
            e.get(eng.SITE_XPOS, self.site_xpos_eng)
        e.get(eng.CONTACT_GEOMS, self.contact_geoms)
        e.get(eng.TIME, self.time)
        e.get(eng.WARN_FLAGS, self.warn)

    def site_xpos(self, model_site_ids):
        idx = [self._site_modelid[int(s)] for s in model_site_ids]
        return self.site_xpos_eng[:, idx, :]

    # -- writes ----------------------------------------------------------------
    def set_ctrl(self, ctrl: torch.Tensor):
        self._ctrl.copy_(ctrl)
        torch.cuda.current_stream(self.device).synchronize()
        self.engine.set(eng.CTRL, self._ctrl)

    @property
    def ctrl(self):
        return self._ctrl

    def set_qfrc_applied(self, f):
        f = torch.as_tensor(f, dtype=self.dtype, device=self.device).expand(self.n_envs, self.model.nv).contiguous()
        torch.cuda.current_stream(self.device).synchronize()
        self.engine.set(eng.QFRC_APPLIED, f)
        self.engine.sync()

    def set_tree_offset(self, off):
        off = torch.as_tensor(off, dtype=self.dtype, device=self.device).contiguous()
        torch.cuda.current_stream(self.device).synchronize()
        self.engine.set(eng.TREE_OFFSET, off)
        self.engine.sync()

    def set_active(self, mask: torch.Tensor):
        self._active.copy_(mask.to(torch.int32))
        torch.cuda.current_stream(self.device).synchronize()
        self.engine.set(eng.ACTIVE, self._active)

    # -- stepping --------------------------------------------------------------
    def reset(self, mask=None):
        self.engine.sync()
        if mask is not None:
            mask = mask.detach().to("cpu").numpy().astype(np.uint8)
        self.engine.reset(mask)

    def forward(self):
        self.engine.forward()

    def step(self, n_substeps: int, key_trace=None):
        self.engine.step(n_substeps, key_trace)
