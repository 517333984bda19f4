"""Uniformly random actions every step (BASELINE config 3 policy), fp64: count diverged envs."""
import warnings; warnings.simplefilter('ignore')
import sys; sys.path.insert(0,'.')
import numpy as np, torch, time
from robopianist_amd import suite
from robopianist_amd.wrappers import CanonicalSpecWrapper
E, steps = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 250
env = CanonicalSpecWrapper(suite.load("RoboPianist-debug-TwinkleTwinkleRousseau-v0", seed=E, n_envs=E, precision=64,
    task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True, primitive_fingertip_collisions=True)))
env.reset()
g = torch.Generator(device='cuda'); g.manual_seed(7)
events = {b: 0 for b in (1, 2, 4, 8, 16, 32)}
prev = torch.zeros(E, dtype=torch.int32, device='cuda')
t0 = time.time()
for t in range(steps):
    a = torch.rand((E, 45), generator=g, device='cuda', dtype=torch.float64) * 2 - 1
    ts = env.step(a)
    w = env.physics.warn.clone()
    for b in events:
        events[b] += int((((w & b) != 0) & ((prev & b) == 0)).sum())   # flags are sticky per episode: count onsets
    prev = torch.where(ts.step_type == 0, torch.zeros_like(w), w)
torch.cuda.synchronize()
print(f"{E} envs x {steps} steps = {E*steps/1e6:.2f} M env-steps of uniformly random actions in {time.time()-t0:.1f} s; "
      f"episodes with flag onsets: bad state {events[1]}, contact/entry capacity {events[2]}, clamped pivot {events[4]}, "
      f"key slots {events[8]}, work list {events[16]}, dense block {events[32]}")
