"""Throughput vs solver budget (MJX-style training settings): fp32/fp64 engine with the Newton
iteration count capped; deviation = max rel |dq| after 100 mj_steps of the key-press scenario
against the uncapped fp64 engine."""
import warnings; warnings.simplefilter('ignore')
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from robopianist_amd import suite, engine
from robopianist_amd.model import scene
from robopianist_amd.wrappers import CanonicalSpecWrapper
from test_gpu_parity import key_press_sequence

actions = np.load("tests/golden/twinkle_twinkle_actions.npy")
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
ctrl = key_press_sequence(si, 300)

def deviation(prec, cap, tol):
    ref = engine.BatchedPhysics(si.model, si.key_joint_ids, n_envs=2, precision=64)
    p = engine.BatchedPhysics(si.model, si.key_joint_ids, n_envs=2, precision=prec)
    if cap: p.set_solver_limits(cap, 0)
    if tol: p.set_solver_tolerance(tol, 0.0)
    worst = 0.0
    for c in ctrl:
        ref.set(engine.CTRL, c[None, :]); p.set(engine.CTRL, c[None, :])
        ref.step(1); p.step(1)
        q0 = ref.qpos[0].astype(np.float64); q1 = p.qpos[0].astype(np.float64)
        worst = max(worst, float((np.abs(q1 - q0) / np.maximum(np.abs(q0), 1e-2)).max()))
    return worst

E = 4096
for prec, cap, tol in ((64, 0, 0), (32, 0, 0), (32, 4, 0), (32, 3, 0), (32, 2, 0), (32, 1, 0), (32, 0, 1e-6), (64, 3, 0)):
    env = CanonicalSpecWrapper(suite.load("RoboPianist-debug-TwinkleTwinkleRousseau-v0", seed=1, n_envs=E, precision=prec,
        task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True,
                         primitive_fingertip_collisions=True, n_steps_lookahead=10)))
    if cap: env.physics.engine.set_solver_limits(cap, 0)
    if tol: env.physics.engine.set_solver_tolerance(tol, 0.0)
    env.reset()
    a = torch.as_tensor(actions, device=env.physics.device, dtype=env.physics.dtype)
    for t in range(20): env.step(a[t].expand(E, -1))
    torch.cuda.synchronize(); t0 = time.time()
    for t in range(20, 140): env.step(a[t].expand(E, -1))
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"fp{prec} newton cap {cap or 'none':>4} tol {tol or 'default':>7}: {E*120/dt:10,.0f} env-steps/s   "
          f"warn {int(env.physics.warn.max())}   key-press deviation vs fp64 uncapped over 300 mj_steps: {deviation(prec, cap, tol):.2e}")
    del env
