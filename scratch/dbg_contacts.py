import warnings; warnings.simplefilter('ignore')
import sys; sys.path.insert(0,'.')
import numpy as np
from robopianist_amd import engine
from robopianist_amd.model import scene
from bench import load_actions
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
m = si.model
phys = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=2, precision=32)
ctrl,_ = load_actions(m)
from collections import Counter
cnt = Counter(); nds=[]; its=[]; nks=[]
for t in range(158):
    phys.set(engine.CTRL, ctrl[t][None,:]); phys.step(10)
    st = phys.get(engine.SOLVER_ITER)[0]; nc = phys.get(engine.NCON)[0]
    nds.append((st>>8)&255); its.append(st&255); nks.append((st>>16)&255)
    cg = phys.get(engine.CONTACT_GEOMS)[0][:nc]
    for a,b in cg: cnt[(m.names['geom'][a].split('/')[-1], m.names['geom'][b].split('/')[-1])]+=1
print('dense rows: mean %.1f max %d; iters mean %.1f; keys mean %.1f'%(np.mean(nds), max(nds), np.mean(its), np.mean(nks)))
print('hist dense rows', Counter(nds))
for k,v in cnt.most_common(25): print(v,k)
