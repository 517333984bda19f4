"""Long-run stability: scripted replay + N(0, 0.3) action noise, 4096 envs, many episodes."""
import warnings; warnings.simplefilter('ignore')
import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from robopianist_amd import suite
from robopianist_amd.wrappers import CanonicalSpecWrapper, MidiEvaluationWrapper
E, steps = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 20000
acts = np.load("tests/golden/twinkle_twinkle_actions.npy"); T = acts.shape[0]
env = MidiEvaluationWrapper(CanonicalSpecWrapper(suite.load("RoboPianist-debug-TwinkleTwinkleRousseau-v0", seed=3, n_envs=E,
    task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True, primitive_fingertip_collisions=not (len(sys.argv) > 2 and sys.argv[2] == "hull"), n_steps_lookahead=10))))
env.reset()
a = torch.as_tensor(acts, device='cuda', dtype=torch.float64)
g = torch.Generator(device='cuda'); g.manual_seed(0)
bad = torch.zeros((), dtype=torch.long, device='cuda'); cap = torch.zeros((), dtype=torch.long, device='cuda')
rsum = torch.zeros((), dtype=torch.float64, device='cuda')
t0 = time.time()
for t in range(steps):
    act = (a[t % T] + 0.3 * torch.randn((E, 45), generator=g, device='cuda', dtype=torch.float64)).clamp(-1, 1)
    ts = env.step(act)
    w = env.physics.warn
    bad += ((w & 1) != 0).sum(); cap += ((w & 2) != 0).sum(); rsum += ts.reward.sum()
    if t % 2000 == 1999:
        torch.cuda.synchronize()
        print(f"step {t+1}: {E*(t+1)/(time.time()-t0):,.0f} env-steps/s, bad-state env-steps {int(bad)}, capacity env-steps {int(cap)}, "
              f"mean reward {float(rsum)/(E*(t+1)):.4f}, mem {torch.cuda.memory_allocated()/1e6:.0f} MB", flush=True)
print("finite:", bool(torch.isfinite(env.physics.qpos).all()), env.get_musical_metrics())
