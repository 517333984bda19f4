"""Per-substep cost trace of a heterogeneous batch (staggered replay): how predictable is an env's
solver-stage cost from what is known before the launch?  Saves gpurun_out/r02/cost_trace.npz."""
import os, sys, warnings
warnings.simplefilter("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from robopianist_amd import engine
from robopianist_amd.wrappers import CanonicalSpecWrapper
E = 4096
base = bench.build_env(2, E, 0, 0, 64)
env = CanonicalSpecWrapper(base); env.reset()
dev = base.physics.device
acts = np.load(os.path.join(bench.ROOT, "tests/golden/twinkle_twinkle_actions.npy"))
T = acts.shape[0]
act_dev = torch.as_tensor(acts, dtype=torch.float64, device=dev)
idx = torch.zeros(E, dtype=torch.long, device=dev)
phase = torch.arange(E, device=dev) % T
for j in range(T):
    base.request_reset(phase == (T - 1 - j))
    ts = env.step(act_dev.index_select(0, idx))
    idx.copy_(torch.where(ts.step_type == 0, torch.zeros_like(idx), torch.clamp(idx + 1, max=T - 1)))
eng = base.physics.engine
cost, its, ncon, kms = [], [], [], []
for t in range(4):
    # one control step = before_step + 10 single substeps, read back after each
    a = act_dev.index_select(0, idx)
    base.task.before_step(base.physics, env._convert(a))
    for k in range(10):
        eng.solver_kernel_time()
        eng.step(1)
        torch.cuda.synchronize()
        ms, _ = eng.solver_kernel_time()
        cost.append(eng.get(engine.ENV_COST).copy()); its.append(eng.get(engine.SOLVER_ITER).copy()); ncon.append(eng.get(engine.NCON).copy())
        kms.append(ms)
    idx.copy_(torch.clamp(idx + 1, max=T - 1))
cost = np.array(cost, np.float64) * 256; its = np.array(its); ncon = np.array(ncon)
os.makedirs("gpurun_out/r02", exist_ok=True)
np.savez_compressed("gpurun_out/r02/cost_trace.npz", cost=cost, its=its, ncon=ncon, kms=np.array(kms))
c0, c1 = cost[:-1].ravel(), cost[1:].ravel()
print("substep-to-substep correlation of the solver wave cycles:", np.corrcoef(c0, c1)[0, 1])
print("kernel ms per launch:", np.round(kms[:12], 3), " sum(cost)/1024/2.37GHz us:", np.round(cost[:12].sum(1) / 1024 / 2.37e3, 1))
