import warnings; warnings.simplefilter('ignore')
import sys; sys.path.insert(0,'.')
import numpy as np, torch, time
from robopianist_amd import suite
from robopianist_amd.wrappers import CanonicalSpecWrapper
E, noise = 4096, float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
acts = np.load("tests/golden/twinkle_twinkle_actions.npy"); T = acts.shape[0]
env = CanonicalSpecWrapper(suite.load("RoboPianist-debug-TwinkleTwinkleRousseau-v0", seed=3, n_envs=E, precision=64,
    task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True, primitive_fingertip_collisions=not (len(sys.argv) > 2 and sys.argv[2] == 'hull'), n_steps_lookahead=10)))
env.reset()
a = torch.as_tensor(acts, device='cuda', dtype=torch.float64)
g = torch.Generator(device='cuda'); g.manual_seed(0)
seen = {b: 0 for b in (1, 2, 4, 8, 16, 32)}; nlast = 0; t0 = time.time(); rs = 0.0
steps = 8 * T
tix = torch.zeros(E, dtype=torch.long, device='cuda')
for t in range(steps):
    base = a[tix % T]                                   # every env follows the script at its own episode time
    act = (base + noise * torch.randn((E, 45), generator=g, device='cuda', dtype=torch.float64)).clamp(-1, 1)
    ts = env.step(act)
    tix = torch.where(ts.step_type == 0, torch.zeros_like(tix), tix + 1)
    w = env.physics.warn
    for b in seen: seen[b] += int(((w & b) != 0).sum())
    nlast += int((ts.step_type == 2).sum()); rs += float(ts.reward.sum())
torch.cuda.synchronize()
print(f"{steps} steps x {E} envs, action noise {noise}, {sys.argv[2] if len(sys.argv) > 2 else 'capsule'} fingertips, fused schedule in use at the end: {env.physics.engine.fused_substeps if hasattr(env.physics, 'engine') else None}: {time.time()-t0:.1f}s, episodes finished {nlast}, mean reward {rs/(steps*E):.4f}, "
      f"finite={bool(torch.isfinite(env.physics.qpos).all())}, env-steps with warn bits {seen}")
