import warnings; warnings.simplefilter('ignore')
import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from robopianist_amd import engine, suite
from robopianist_amd.wrappers import CanonicalSpecWrapper
E=4096
env = suite.load("RoboPianist-debug-TwinkleTwinkleRousseau-v0", n_envs=E, task_kwargs=dict(trim_silence=True, gravity_compensation=True, primitive_fingertip_collisions=True, n_steps_lookahead=10))
phys = env.physics.engine; P = env.physics; m = env.task.scene.model
actions = np.load("tests/golden/twinkle_twinkle_actions.npy")
dev = P.device
lo = torch.as_tensor(m.actuator_ctrlrange[:,0], dtype=torch.float32, device=dev); hi = torch.as_tensor(m.actuator_ctrlrange[:,1], dtype=torch.float32, device=dev)
act = torch.as_tensor(actions, dtype=torch.float32, device=dev)
def run(mode, n=60):
    env.reset(); phys.kernel_time()
    if mode in ('active',): P.set_active(torch.ones(E, dtype=torch.bool, device=dev))
    else: phys.set(engine.ACTIVE, None)
    t0=time.perf_counter()
    for t in range(n):
        c = lo + (act[t,:-1]+1)*0.5*(hi-lo)
        P.set_ctrl(c.expand(E,-1).contiguous())
        phys.step(10)
        if mode=='refresh': P.refresh()
        if mode=='sleep': phys.sync(); time.sleep(0.002)
        if mode=='sync': phys.sync()
    phys.sync()
    dt=time.perf_counter()-t0
    k,_=phys.kernel_time()
    print(mode, 'kernel ms %.2f'%k, 'wall/step ms %.2f'%(1e3*dt/n))
for mode in ('plain','active','refresh','sync','sleep','plain'):
    run(mode)
