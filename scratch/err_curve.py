import warnings; warnings.simplefilter('ignore')
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from robopianist_amd import engine
from robopianist_amd.model import scene
from oracle.rp_oracle import Oracle
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
m = si.model
a = np.load("tests/golden/twinkle_twinkle_actions.npy").astype(np.float64)[:, :-1]
lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
ctrl = np.repeat(lo + (np.clip(a, -1, 1) + 1.0) * 0.5 * (hi - lo), 10, axis=0)
phys = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=1, precision=64)
orc = Oracle(m, phys.blob)
for i in range(1500):
    c = ctrl[i]
    phys.set(engine.CTRL, c[None,:]); orc.ctrl[:] = c
    phys.step(1); orc.step(1)
    q = phys.qpos.astype(np.float64)[0]
    rel = np.abs(q-orc.qpos)/np.maximum(np.abs(orc.qpos),1e-2)
    if (i+1) % 50 == 0:
        j = int(rel.argmax())
        print(i+1, "%.2e" % rel.max(), "dof", j, m.names["joint"][j] if "joint" in m.names else "", "ncon", orc.ncon)
