import warnings; warnings.simplefilter('ignore')
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from robopianist_amd import engine
from robopianist_amd.model import scene
from oracle.rp_oracle import Oracle
from bench import load_actions
from test_gpu_parity import key_press_sequence
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
m = si.model
ctrl,_ = load_actions(m)
kp = key_press_sequence(si, 1000)
for prec in (32, 64):
  for tol in (1e-8, 1e-7, 1e-6, 1e-5):
    # parity free-running key-press
    phys = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=1, precision=prec); phys.set_solver_tolerance(tol)
    orc = Oracle(m, phys.blob)
    worst=0
    for c in kp:
        phys.set(engine.CTRL, c[None,:]); orc.ctrl[:]=c; phys.step(1); orc.step(1)
        q=phys.qpos[0].astype(np.float64); worst=max(worst,(np.abs(q-orc.qpos)/np.maximum(np.abs(orc.qpos),1e-2)).max())
    # replay parity (158*10 steps) + speed
    E=4096
    p2 = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=E, precision=prec); p2.set_solver_tolerance(tol)
    o2 = Oracle(m, p2.blob)
    its=[]; worst2=0
    p2.kernel_time()
    for t in range(100):
        p2.set(engine.CTRL, ctrl[t][None,:]); p2.step(10)
        o2.ctrl[:]=ctrl[t]; o2.step(10)
        its.append(p2.get(engine.SOLVER_ITER)[0]&255)
        q=p2.qpos[0].astype(np.float64); worst2=max(worst2,(np.abs(q-o2.qpos)/np.maximum(np.abs(o2.qpos),1e-2)).max())
    k,_=p2.kernel_time()
    print('prec %d tol %.0e: keypress rel %.2e | replay rel@1000 %.2e iters %.2f kernel %.2f ms -> %.0f env-steps/s'%(prec,tol,worst,worst2,np.mean(its),k,E/(k*1e-3)))
