import warnings; warnings.simplefilter('ignore')
import sys; sys.path.insert(0,'.')
import numpy as np, torch
from robopianist_amd import suite, engine
from collections import Counter
env = suite.load("RoboPianist-debug-CMajorScaleTwoHands-v0", seed=1, n_envs=512, task_kwargs=dict(primitive_fingertip_collisions=True))
spec = env.action_spec(); rng = np.random.RandomState(0)
env.reset(); c=Counter(); maxcon=0; maxd=0
for t in range(150):
    a = rng.uniform(spec.minimum, spec.maximum, size=(512,)+spec.shape)
    ts = env.step(a)
    w = env.physics.warn.cpu().numpy()
    for bit in (1,2,4,8,16,32): c[bit] += int(((w & bit)>0).sum())
    st = env.physics.engine.get(engine.SOLVER_ITER)
    maxcon = max(maxcon, int(env.physics.engine.get(engine.NCON).max())); maxd = max(maxd, int(((st>>8)&255).max()))
    env.physics.warn.zero_()
print('warn counts (env-steps flagged) of', 150*512, dict(c), 'max ncon', maxcon, 'max dense rows', maxd)
