import warnings; warnings.simplefilter('ignore')
import sys; sys.path.insert(0,'.')
import numpy as np, torch, time
from robopianist_amd import suite
from robopianist_amd.wrappers import CanonicalSpecWrapper
for E, prec, steps in ((1, 64, 40), (3, 32, 40), (8192, 64, 120), (4096, 32, 200)):
    env = CanonicalSpecWrapper(suite.load("RoboPianist-debug-TwinkleTwinkleRousseau-v0", seed=E, n_envs=E, precision=prec,
        task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True, primitive_fingertip_collisions=True, n_steps_lookahead=10)))
    ts = env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(E)
    dt = torch.float64 if prec == 64 else torch.float32
    t0 = time.time(); nlast = 0; rsum = 0.0; seen = {b: 0 for b in (1, 2, 4, 8, 16, 32)}
    for t in range(steps):
        a = torch.rand((E, 45), generator=g, device='cuda', dtype=dt) * 2 - 1
        ts = env.step(a)
        wv = env.physics.warn
        for b in seen: seen[b] += int(((wv & b) != 0).sum())
        if ts.reward is not None:
            nlast += int((ts.step_type == 2).sum()); rsum += float(ts.reward.sum())
    torch.cuda.synchronize()
    w = env.physics.warn
    q = env.physics.qpos
    bits = {b: int(((w & b) != 0).sum()) for b in (1, 2, 4, 8, 16, 32)}
    print("   warn bit counts", bits, "cumulative flags seen:", seen)
    print(f"E={E} prec={prec}: {steps} steps {time.time()-t0:.2f}s finite={bool(torch.isfinite(q).all())} warn bits OR={int(w.max()) if E else 0} "
          f"envs with warn={int((w!=0).sum())} LAST seen={nlast} mean reward/step={rsum/max(1,E*steps):.3f} max|q|={float(q.abs().max()):.3f}")
