import warnings; warnings.simplefilter('ignore')
import sys, os; sys.path.insert(0,'/root/repo'); os.environ["RP_SKIP_SELF_CHECK"]="1"
import numpy as np
from robopianist_amd import engine
from robopianist_amd.model import scene
from oracle.rp_oracle import Oracle
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
m = si.model
rng = np.random.default_rng(0)
lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
ctrl = lo + rng.uniform(0.2, 0.8, m.nu) * (hi - lo)
phys = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=2, precision=64)
orc = Oracle(m, phys.blob)
phys.set(engine.CTRL, ctrl[None, :]); orc.ctrl[:] = ctrl
for s in range(3):
    phys.step(1); orc.step(1)
    qa = phys.get(engine.QACC_WARMSTART)[0]; oq = orc.qacc_warmstart
    e = np.abs(qa - oq)
    it = phys.get(engine.SOLVER_ITER)[0]
    print("substep", s+1, "iters gpu", it & 255, "dense", (it>>8)&255, "keys", (it>>16)&255, "ncon", phys.get(engine.NCON)[0], orc.ncon, "max qacc err %.2e at dof %d (%s)" % (e.max(), e.argmax(), m.names["joint"][int(e.argmax())]))
    bad = np.flatnonzero(e > 1e-6 * (1 + np.abs(oq)))
    print("   dofs off:", len(bad), bad[:20], " oracle iters", getattr(orc, "niter", None))
