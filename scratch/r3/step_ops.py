"""Which torch ops (and how many device launches) one env.step of the bench loop issues: torch profiler, 8 steps."""
import warnings; warnings.simplefilter('ignore')
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from robopianist_amd import suite
from robopianist_amd.wrappers import CanonicalSpecWrapper
E = 4096
acts = np.load("tests/golden/twinkle_twinkle_actions.npy"); T = acts.shape[0]
env = CanonicalSpecWrapper(suite.load("RoboPianist-debug-TwinkleTwinkleRousseau-v0", seed=3, n_envs=E, precision=64,
    task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True, primitive_fingertip_collisions=True, n_steps_lookahead=10)))
env.reset()
a = torch.as_tensor(acts, device='cuda', dtype=torch.float64)
idx = torch.zeros(E, dtype=torch.long, device='cuda')
def step():
    global idx
    ts = env.step(a.index_select(0, idx))
    first = ts.step_type == 0
    idx.copy_(torch.where(first, torch.zeros_like(idx), torch.clamp(idx + 1, max=T - 1)))
for _ in range(5): step()
torch.cuda.synchronize()
N = 8
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N): step()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type is not None]
import collections
cnt = collections.Counter(); dev = collections.Counter()
for e in prof.key_averages(group_by_stack_n=4):
    if e.device_time_total > 0 or e.count:
        pass
rows = []
for e in prof.key_averages(group_by_stack_n=6):
    if e.self_device_time_total > 0:
        st = [s for s in e.stack if 'robopianist_amd' in s or 'step_ops' in s or 'bench' in s]
        rows.append((e.count / N, e.self_device_time_total / N, e.key, (st[0] if st else '')[-90:]))
rows.sort(key=lambda r: -r[1])
tot = sum(r[1] for r in rows if 'rp_' not in r[2] and 'Memcpy' not in r[2])
for r in rows[:45]: print('%5.2f/step %8.1f us/step  %-42s %s' % r)
