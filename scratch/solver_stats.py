import warnings; warnings.simplefilter('ignore')
import sys; sys.path.insert(0,'.')
import numpy as np
from robopianist_amd import engine
from robopianist_amd.model import scene
from bench import load_actions
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
m = si.model
phys = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=1, precision=64)
ctrl,_ = load_actions(m)
rows=[]
for t in range(158):
    phys.set(engine.CTRL, ctrl[t][None,:])
    for s in range(10):
        phys.step(1)
        v=int(phys.get(engine.SOLVER_ITER)[0]); nc=int(phys.get(engine.NCON)[0])
        rows.append((v&255,(v>>8)&255,(v>>16)&255,nc))
r=np.array(rows)
print('iters mean %.2f max %d hist'%(r[:,0].mean(), r[:,0].max()), np.bincount(r[:,0]))
print('dense rows mean %.2f max %d hist'%(r[:,1].mean(), r[:,1].max()), np.bincount(r[:,1]))
print('keys mean %.2f max %d'%(r[:,2].mean(), r[:,2].max()), np.bincount(r[:,2]))
print('ncon mean %.2f max %d'%(r[:,3].mean(), r[:,3].max()), np.bincount(r[:,3]))
np.save('gpurun_out/solver_stats.npy', r)
