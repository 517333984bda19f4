"""Distribution of the per-env solver-stage cost (cycles) in a heterogeneous batch: how far the
solver launch can be from perfectly balanced.  GPU box only."""
import os, sys, warnings
warnings.simplefilter("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from robopianist_amd import engine
from robopianist_amd.wrappers import CanonicalSpecWrapper
E = 4096
for cfg in (2, 3):
    base = bench.build_env(cfg, E, 0, 0, 64)
    env = CanonicalSpecWrapper(base)
    env.reset()
    dev = base.physics.device
    A = env.action_spec().shape[0]
    acts = np.load(os.path.join(bench.ROOT, "tests/golden/twinkle_twinkle_actions.npy"))
    T = acts.shape[0]
    act_dev = torch.as_tensor(acts, dtype=torch.float64, device=dev)
    idx = torch.zeros(E, dtype=torch.long, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(1)
    if cfg == 2:
        phase = torch.arange(E, device=dev) % T
        for j in range(T):
            base.request_reset(phase == (T - 1 - j))
            ts = env.step(act_dev.index_select(0, idx))
            idx.copy_(torch.where(ts.step_type == 0, torch.zeros_like(idx), torch.clamp(idx + 1, max=T - 1)))
    for t in range(40):
        a = act_dev.index_select(0, idx) if cfg == 2 else torch.rand((E, A), generator=g, device=dev, dtype=torch.float64) * 2 - 1
        ts = env.step(a)
        if cfg == 2:
            idx.copy_(torch.where(ts.step_type == 0, torch.zeros_like(idx), torch.clamp(idx + 1, max=T - 1)))
    torch.cuda.synchronize()
    c = base.physics.engine.get(engine.ENV_COST).astype(np.float64) * 256 / 2.4e9 * 1e6  # us at 2.4 GHz
    it = base.physics.engine.get(engine.SOLVER_ITER)
    print(f"config {cfg}: solver-stage wave time per env [us]: mean {c.mean():.1f} median {np.median(c):.1f} p90 {np.percentile(c,90):.1f} "
          f"p99 {np.percentile(c,99):.1f} max {c.max():.1f}; balanced launch = {c.sum()/1024:.1f} us, longest env {c.max():.1f} us; "
          f"iters mean {(it&255).mean():.2f} max {(it&255).max()}, dense rows mean {((it>>8)&255).mean():.1f} max {((it>>8)&255).max()}")
    print("  histogram (us):", np.histogram(c, bins=[0,50,75,100,125,150,200,250,300,400,600,1000,5000])[0].tolist())

# clock calibration: lockstep replay -> every env identical: kernel time = 4 rounds of the same wave time
base = bench.build_env(2, E, 0, 0, 64)
env = CanonicalSpecWrapper(base); env.reset()
phys = base.physics.engine
phys.solver_kernel_time()
tot_c, n = 0.0, 0
for t in range(60):
    env.step(act_dev[t].expand(E, -1))
    if t >= 20:
        torch.cuda.synchronize()
        tot_c += float(base.physics.engine.get(engine.ENV_COST).astype(np.float64).mean()) * 256; n += 1
ms, nl = phys.solver_kernel_time()
print(f"lockstep calibration: mean wave cycles {tot_c/n:.0f} (last substep of each step), solver kernel avg {ms*1e3:.1f} us over {nl} launches "
      f"-> if 4 rounds: {tot_c/n*4/(ms*1e-3)/1e9:.2f} GHz shader clock")
