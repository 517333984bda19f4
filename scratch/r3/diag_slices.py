import warnings; warnings.simplefilter('ignore')
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from robopianist_amd import engine
from robopianist_amd.model import scene
import test_gpu_parity as tgp
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
m = si.model
E = 1100
ctrl = tgp._replay_ctrl(si)
rng = np.random.default_rng(1)
gain = 1 + 0.1 * rng.standard_normal((E, 1))
ref = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=E, precision=64)
ref.set_acc_sensors(True)
modes = []
for slices, order in ((2, False), (4, True), (0, True), (1, False)):
    p = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=E, precision=64)
    p.set_stream_slices(slices); p.set_cost_ordered_launch(order); p.set_acc_sensors(True)
    modes.append(p)
for t in range(24):
    c = ctrl[10 * (t + 20)][None, :] * gain
    mask = None
    if t == 8:
        mask = np.zeros(E, np.uint8); mask[::7] = 1
    for p in [ref] + modes:
        p.set(engine.CTRL, c)
        if mask is not None: p.reset(mask)
        if t == 12:
            act = np.ones(E, np.int32); act[5::11] = 0
            p.view(engine.ACTIVE).copy_(tgp.torch_i32(act))
        if t == 13: p.view(engine.ACTIVE).fill_(1)
        p.step(10)
    for i, p in enumerate(modes):
        dq = np.abs(ref.qpos - p.qpos).max(1)
        bad = np.flatnonzero(dq > 0)
        if len(bad):
            h = ref.get(engine.DEBUG_HANDOVER_HDR)
            print("t", t, "mode", i, "nbad", len(bad), "first", bad[:12], "max", dq.max(), "light flags of bad", h[bad[:12], 6], "warn", p.warn_flags[bad[:6]], ref.warn_flags[bad[:6]])
print("done")
