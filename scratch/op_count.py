import warnings; warnings.simplefilter('ignore')
import sys, os; sys.path.insert(0,'.')
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity, record_function
from robopianist_amd import suite
from robopianist_amd.wrappers import CanonicalSpecWrapper
E=4096
base = suite.load("RoboPianist-debug-TwinkleTwinkleRousseau-v0", seed=1, n_envs=E, precision=64,
    task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True, primitive_fingertip_collisions=True, n_steps_lookahead=10))
env = CanonicalSpecWrapper(base)
task, phys = base.task, base.physics
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        with record_function("HOOK_" + (label or name)):
            return f(*a, **k)
    setattr(obj, name, g)
for n in ["initialize_episode","before_step","after_substeps","after_step","get_observation","get_reward","should_terminate_episode","get_discount"]:
    wrap(task, n)
for n in ["reset","set_active","forward","step"]:
    wrap(phys, n, "phys_"+n)
for n in list(task.reward_fn.reward_fns):
    f = task.reward_fn.reward_fns[n]
    def mk(f, n):
        def g(p):
            with record_function("REW_" + n):
                return f(p)
        return g
    task.reward_fn.reward_fns[n] = mk(f, n)
wrap(env, "_convert")
acts = np.load("tests/golden/twinkle_twinkle_actions.npy")
a = torch.as_tensor(acts, device=phys.device, dtype=torch.float64)
env.reset()
for t in range(5): env.step(a[t].expand(E,-1))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for t in range(5,15): env.step(a[t].expand(E,-1))
    torch.cuda.synchronize()
ev = prof.events()
hooks = [e for e in ev if e.name.startswith(("HOOK_","REW_"))]
kern = [e for e in ev if e.device_type is not None and str(e.device_type).endswith("CUDA")]
# attribute launches: count cpu-side "hipLaunchKernel"/aten ops within the hook time range
launches = [e for e in ev if e.name in ("hipLaunchKernel","hipExtModuleLaunchKernel","hipMemcpyAsync","hipExtLaunchKernel","hipModuleLaunchKernel")]
import collections
cnt = collections.Counter(); tot=len(launches)
for h in hooks:
    tr = h.time_range
    for l in launches:
        if l.time_range.start >= tr.start and l.time_range.end <= tr.end:
            cnt[h.name]+=1
for k,v in sorted(cnt.items(), key=lambda kv:-kv[1]): print("%-40s %6.1f launches/step" % (k, v/10))
print("total launches/step", tot/10)
