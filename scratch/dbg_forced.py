import warnings; warnings.simplefilter('ignore')
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from robopianist_amd import engine
from robopianist_amd.model import scene
from oracle.rp_oracle import Oracle
from test_gpu_parity import _ctrl_sequence
prec = int(sys.argv[1]) if len(sys.argv)>1 else 64
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
m = si.model
phys = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=2, precision=prec)
orc = Oracle(m, phys.blob)
ctrl = _ctrl_sequence(m, 300, 1)
worst = 0
for s in range(300):
    # teacher forcing: GPU starts every step from the oracle's state
    phys.set(engine.QPOS, orc.qpos[None,:]); phys.set(engine.QVEL, orc.qvel[None,:])
    phys.set(engine.QACC_WARMSTART, orc.qacc_warmstart[None,:])
    phys.set(engine.CTRL, ctrl[s][None,:]); orc.ctrl[:] = ctrl[s]
    v0 = orc.qvel.copy()
    phys.step(1); orc.step(1)
    dv = np.abs(phys.qvel[0]-orc.qvel); dq = np.abs(phys.qpos[0]-orc.qpos)
    rel = dv.max()/max(np.abs(orc.qvel-v0).max(),1e-12)
    nc = phys.get(engine.NCON)[0]
    if rel > worst or s%50==0 or nc!=orc.ncon:
        print(s, 'dq %.2e dv %.2e rel_dv %.2e'%(dq.max(), dv.max(), rel), 'argmax', dv.argmax(), 'ncon', nc, orc.ncon, 'it', phys.get(engine.SOLVER_ITER)[0], orc.solver_iter)
    worst = max(worst, rel)
