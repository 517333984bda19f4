"""Substep-resolution look at one env that diverges under random actions (see diverge_diag.py)."""
import warnings; warnings.simplefilter('ignore')
import sys; sys.path.insert(0,'.')
import numpy as np, torch
from robopianist_amd import suite, engine
from robopianist_amd.wrappers import CanonicalSpecWrapper
E = 8192
e, t_bad = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3493, 3)
env = CanonicalSpecWrapper(suite.load("RoboPianist-debug-TwinkleTwinkleRousseau-v0", seed=E, n_envs=E, precision=64,
    task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True, primitive_fingertip_collisions=True)))
env.reset()
g = torch.Generator(device='cuda'); g.manual_seed(E)
eng = env.physics.engine
for t in range(t_bad):
    a = torch.rand((E, 45), generator=g, device='cuda', dtype=torch.float64) * 2 - 1
    env.step(a)
a = torch.rand((E, 45), generator=g, device='cuda', dtype=torch.float64) * 2 - 1
env.task.before_step(env.physics, env._convert(a))
m = env.task.scene.model
names = m.names["joint"]
for s in range(10):
    q0 = env.physics.qpos[e].clone(); v0 = env.physics.qvel[e].clone()
    eng.step(1)
    w = int(env.physics.warn[e]); n = int(eng.get(engine.NCON)[e]); it = int(eng.get(engine.SOLVER_ITER)[e])
    v = env.physics.qvel[e]
    j = int(torch.nan_to_num(v.abs(), nan=1e30).argmax())
    cg = eng.get(engine.CONTACT_GEOMS)[e][:max(n, 0)]
    cd = eng.get(engine.CONTACT_DIST)[e][:max(n, 0)]
    print(f"substep {s}: warn {w} ncon {n} newton {it & 255} dense {(it >> 8) & 255} keys {(it >> 16) & 255} max|qvel| {float(v.abs().max()):.3g} at {names[j]} "
          f"(q {float(env.physics.qpos[e, j]):.4f}, range {m.jnt_range[j]}) min dist {cd.min() if n > 0 else 0:.4g}")
    if n > 0 and n <= 12:
        gn = m.names["geom"]
        print("     contacts:", [(gn[a_].split('/')[-1], gn[b_].split('/')[-1], '%.4f' % d) for (a_, b_), d in zip(cg, cd)])
