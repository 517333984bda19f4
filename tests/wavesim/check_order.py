"""TEST INFRASTRUCTURE: cost-ordered launch on the CPU wave emulator: same bits as the plain launch order."""
import os, sys, warnings
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("RP_ENGINE_LIB", os.path.join(HERE, "_build", "librp_engine_wavesim.so"))
os.environ["RP_SKIP_SELF_CHECK"] = "1"
import numpy as np
import test_gpu_parity as tgp
from robopianist_amd import engine
from robopianist_amd.model import scene
warnings.simplefilter("ignore")
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
E = 24
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
a = engine.BatchedPhysics(si.model, si.key_joint_ids, n_envs=E, precision=64)
b = engine.BatchedPhysics(si.model, si.key_joint_ids, n_envs=E, precision=64, blob=a.blob)
b.set_cost_ordered_launch(True)
rng = np.random.default_rng(0)
lo, hi = si.model.actuator_ctrlrange[:, 0], si.model.actuator_ctrlrange[:, 1]
for t in range(nsteps):
    c = lo + rng.uniform(0.1, 0.9, (E, si.model.nu)) * (hi - lo)
    for p in (a, b):
        p.set(engine.CTRL, c); p.step(5)
    assert np.array_equal(a.qpos, b.qpos) and np.array_equal(a.qvel, b.qvel), t
print("cost-ordered launch bit-identical over", nsteps, "steps; max contacts", int(a.get(engine.NCON).max()))
