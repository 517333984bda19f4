"""How do the dense (cross-coupled) rows of a solve split between the two hands?  (debug build: bits 24-31 of
RP_SOLVER_ITER = dirty rows among lanes 0..25 = the first hand's links)"""
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
v = eng.get(engine.SOLVER_ITER).astype(np.int64)
it, nd, nk, n0 = v & 255, (v >> 8) & 255, (v >> 16) & 255, (v >> 24) & 255
n1 = nd - n0   # (second hand's links + all key slots)
print("envs with a dense block: %.1f %%; nd mean %.1f (when > 0: %.1f)" % (100 * (nd > 0).mean(), nd.mean(), nd[nd > 0].mean()))
both = (n0 > 0) & (n1 > 0)
print("dense rows on both hands: %.1f %% of envs (of those with a block: %.1f %%)" % (100 * both.mean(), 100 * both.sum() / max(1, (nd > 0).sum())))
big, small = np.maximum(n0, n1), np.minimum(n0, n1)
print("when on both: larger side mean %.1f, smaller side mean %.1f; sequential pivots saved if the sides factor side by side: %.1f %% of all dense rows"
      % (big[both].mean(), small[both].mean(), 100 * small[both].sum() / max(1, nd.sum())))
print("histogram nd:", np.bincount(nd, minlength=50)[:50])
