"""Test double for robopianist_amd.suite.physics.TorchPhysics (CPU, no dynamics).

Lets the host-side task logic (goal/fingering tables, hook order, termination,
rewards as functions of state) be tested without a GPU.  It is NOT a physics
fallback: qpos only changes when a test writes it."""
import numpy as np
import torch


class FakePhysics:
    def __init__(self, scene_info, n_envs):
        m = scene_info.model
        self.scene, self.model, self.n_envs = scene_info, m, n_envs
        self.device = torch.device("cpu")
        self.dtype = torch.float64
        E = n_envs
        self.qpos = torch.zeros((E, m.nv), dtype=self.dtype)
        self.qvel = torch.zeros((E, m.nv), dtype=self.dtype)
        self.act_force = torch.zeros((E, m.nu), dtype=self.dtype)
        self.act_vel = torch.zeros((E, m.nu), dtype=self.dtype)
        self.contact_geoms = torch.full((E, 64, 2), -1, dtype=torch.int32)
        self.warn = torch.zeros(E, dtype=torch.int32)
        self.time = torch.zeros(E, dtype=self.dtype)
        self._ctrl = torch.zeros((E, m.nu), dtype=self.dtype)
        self._sites = torch.zeros((E, m.nsite, 3), dtype=self.dtype)
        self.active = torch.ones(E, dtype=torch.bool)
        self.n_steps = 0
        self.timestep = float(m.opt_timestep)

    @property
    def ctrl(self):
        return self._ctrl

    def set_ctrl(self, c):
        self._ctrl = c.clone()

    def set_active(self, mask):
        self.active = mask.bool().clone()

    def set_qfrc_applied(self, f):
        pass

    def set_tree_offset(self, off):
        self.tree_offset = off.clone()

    def site_xpos(self, ids):
        return self._sites[:, [int(i) for i in ids], :]

    def reset(self, mask=None):
        sel = slice(None) if mask is None else mask.bool()
        self.qpos[sel] = 0
        self.qvel[sel] = 0
        self._ctrl[sel] = 0
        self.time[sel] = 0

    def forward(self):
        pass

    def refresh(self):
        pass

    def step(self, n, key_trace=None):
        self.n_steps += n
        self.time = self.time + self.active.to(self.dtype) * n * self.timestep

    def state_dict(self):
        return {"qpos": self.qpos.clone(), "qvel": self.qvel.clone(), "ctrl": self._ctrl.clone(),
                "time": self.time.clone()}

    def load_state_dict(self, sd):
        self.qpos.copy_(sd["qpos"]); self.qvel.copy_(sd["qvel"]); self._ctrl.copy_(sd["ctrl"])
        self.time.copy_(sd["time"])
