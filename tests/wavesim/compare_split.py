"""TEST INFRASTRUCTURE (emulator / GPU): the split position stage (front / pooled narrow phase / back) against the one-kernel stage,
bit for bit, free-running on contact-rich control sequences.  Usage: compare_split.py [nsteps] [scene: cap|hull] [seq]"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["RP_SKIP_SELF_CHECK"] = "1"
import numpy as np
import test_gpu_parity as tgp
from robopianist_amd import engine
from robopianist_amd.model import scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
hull = len(sys.argv) > 2 and sys.argv[2] == "hull"
seq = sys.argv[3] if len(sys.argv) > 3 else "wrist"
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=not hull)
m = si.model
if seq == "wrist": ctrl = tgp.wrist_press_sequence(si, n)
elif seq == "replay": ctrl = tgp._replay_ctrl(si)[int(sys.argv[4]) if len(sys.argv) > 4 else 300:][:n]
else: ctrl = tgp.ctrl_sequence(m, n, 1)
ph = [engine.BatchedPhysics(m, si.key_joint_ids, n_envs=3, precision=64) for _ in range(2)]
ph[0].set_split_position_stage(True); ph[1].set_split_position_stage(False)
for p in ph: p.set_stream_slices(1); p.set_fused_substeps(False)
maxcon = 0
for t, c in enumerate(ctrl):
    for p in ph:
        p.set(engine.CTRL, np.tile(c[None, :], (3, 1)))
        p.step(2)
    q0, q1 = ph[0].qpos, ph[1].qpos
    v0, v1 = ph[0].qvel, ph[1].qvel
    n0, n1 = ph[0].get(engine.NCON), ph[1].get(engine.NCON)
    g0, g1 = ph[0].get(engine.CONTACT_GEOMS), ph[1].get(engine.CONTACT_GEOMS)
    d0, d1 = ph[0].get(engine.CONTACT_DIST), ph[1].get(engine.CONTACT_DIST)
    maxcon = max(maxcon, int(n0.max()))
    ok = (q0 == q1).all() and (v0 == v1).all() and (n0 == n1).all() and (g0 == g1).all() and (d0 == d1).all()
    if not ok:
        print("MISMATCH at step", t, "ncon", n0, n1, "max|dq|", np.abs(q0 - q1).max(), "max|dv|", np.abs(v0 - v1).max(),
              "geoms equal", (g0 == g1).all(), "dist max diff", np.abs(d0 - d1).max())
        sys.exit(1)
    assert ph[0].warn_flags.max() == 0, ph[0].warn_flags
print(f"split == whole over {len(ctrl)} x 2 mj_steps, bit for bit; max contacts {maxcon}")
