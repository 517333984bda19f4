import warnings; warnings.simplefilter('ignore')
import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from robopianist_amd import engine
from robopianist_amd.model import scene
from bench import load_actions
E=4096
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
m = si.model
ctrl,_ = load_actions(m)
def run(tag, phys, sync=False, n=60):
    phys.reset(); phys.sync(); phys.kernel_time()
    c = [np.ascontiguousarray(np.broadcast_to(ctrl[t].astype(np.float32),(E,m.nu))) for t in range(n)]
    dev = [torch.as_tensor(x, device='cuda') for x in c]
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for t in range(n):
        phys.set(engine.CTRL, dev[t]); phys.step(10)
        if sync: phys.sync()
    phys.sync(); dt=time.perf_counter()-t0
    k,_=phys.kernel_time()
    print(tag, 'kernel ms %.2f wall/step %.2f'%(k, 1e3*dt/n), 'qsum', float(phys.qpos.sum()))
p1 = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=E)
run('own stream async', p1)
run('own stream +sync', p1, sync=True)
run('own stream async', p1)
p2 = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=E)
p2.set_stream(torch.cuda.current_stream().cuda_stream)
run('torch default stream async', p2)
s = torch.cuda.Stream()
p3 = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=E)
p3.set_stream(s.cuda_stream)
run('torch side stream async', p3)
run('own stream async', p1)
