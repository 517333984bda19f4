import warnings; warnings.simplefilter('ignore')
import sys; sys.path.insert(0,'.')
import numpy as np
from robopianist_amd import engine
from robopianist_amd.model import scene
from oracle.rp_oracle import Oracle
sys.path.insert(0,'tests')
from test_gpu_parity import _ctrl_sequence
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
m = si.model
phys = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=2, precision=64)
orc = Oracle(m, phys.blob)
ctrl = _ctrl_sequence(m, 300, 1)
prev = 0
for s in range(300):
    phys.set(engine.CTRL, ctrl[s][None,:]); orc.ctrl[:] = ctrl[s]
    phys.step(1); orc.step(1)
    q = phys.qpos[0]; e = np.abs(q-orc.qpos); 
    nc = phys.get(engine.NCON)[0]; it = phys.get(engine.SOLVER_ITER)[0]
    if e.max() > 3*prev+1e-13 or nc != orc.ncon or s%25==0:
        cg = phys.get(engine.CONTACT_GEOMS)[0][:nc]
        oc = orc.contact.reshape(-1,16)[:, 13:15].astype(int)
        print(s, 'err %.3e'%e.max(), 'argmax', e.argmax(), 'ncon gpu/orc', nc, orc.ncon, 'iter', it, orc.solver_iter, 'warn', phys.warn_flags[0])
        if nc != orc.ncon or s%25==0:
            print('   gpu pairs', sorted(map(tuple,cg.tolist())))
            print('   orc pairs', sorted(map(tuple,oc.tolist())))
    prev = max(e.max(), prev)
