import warnings; warnings.simplefilter('ignore')
import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from robopianist_amd import engine
from robopianist_amd.model import scene
from oracle.rp_oracle import Oracle
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
m = si.model
rng = np.random.default_rng(0)
lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
ctrl = lo + rng.uniform(0.2, 0.8, m.nu) * (hi - lo)
for prec in (64, 32):
    phys = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=8, precision=prec)
    orc = Oracle(m, phys.blob)
    phys.set(engine.CTRL, ctrl[None, :]); orc.ctrl[:] = ctrl
    errs=[]
    for s in range(10):
        phys.step(1); orc.step(1)
        q = phys.qpos.astype(np.float64)
        errs.append(np.abs(q[0]-orc.qpos).max())
    print(prec, "err per substep", ["%.1e"%e for e in errs], "iters", phys.get(engine.SOLVER_ITER)[0] & 255, "warn", phys.warn_flags.max())
