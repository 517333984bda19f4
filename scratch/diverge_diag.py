"""Which warn flags / contact counts precede a diverged (BADSTATE) env under random actions?"""
import warnings; warnings.simplefilter('ignore')
import sys; sys.path.insert(0,'.')
import numpy as np, torch
from robopianist_amd import suite, engine
from robopianist_amd.wrappers import CanonicalSpecWrapper
E, steps = 8192, 120
env = CanonicalSpecWrapper(suite.load("RoboPianist-debug-TwinkleTwinkleRousseau-v0", seed=E, n_envs=E, precision=64,
    task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True, primitive_fingertip_collisions=True)))
env.reset()
g = torch.Generator(device='cuda'); g.manual_seed(E)
eng = env.physics.engine
hist_w, hist_n, hist_it, hist_q = [], [], [], []
for t in range(steps):
    a = torch.rand((E, 45), generator=g, device='cuda', dtype=torch.float64) * 2 - 1
    env.step(a)
    hist_w.append(env.physics.warn.clone().cpu().numpy())
    hist_n.append(eng.get(engine.NCON).copy())
    hist_it.append(eng.get(engine.SOLVER_ITER).copy())
    hist_q.append(env.physics.qvel.abs().max(dim=1).values.cpu().numpy())
W = np.array(hist_w); N = np.array(hist_n); IT = np.array(hist_it); Q = np.array(hist_q)
bad_envs = np.unique(np.nonzero(W & 1)[1])
print("envs that ever went bad:", len(bad_envs), " envs ever CONTACT_FULL:", len(np.unique(np.nonzero(W & 2)[1])),
      " ever HESSIAN:", len(np.unique(np.nonzero(W & 4)[1])), " ever DENSE_FULL:", len(np.unique(np.nonzero(W & 32)[1])))
for e in bad_envs[:12]:
    t0 = int(np.nonzero(W[:, e] & 1)[0][0])
    lo = max(0, t0 - 5)
    print(f"env {e}: bad at step {t0}; flags {[int(x) for x in W[lo:t0+1, e]]} ncon {[int(x) for x in N[lo:t0+1, e]]} "
          f"newton {[int(x) & 255 for x in IT[lo:t0+1, e]]} dense rows {[(int(x) >> 8) & 255 for x in IT[lo:t0+1, e]]} max|qvel| {['%.1f' % x for x in Q[lo:t0+1, e]]}")
print("max ncon seen", N.max(), "max dense rows", ((IT >> 8) & 255).max(), "max newton", (IT & 255).max())
