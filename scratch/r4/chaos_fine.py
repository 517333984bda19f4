import sys, os, warnings
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from chaos_control import *
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=False)
m = si.model
blob = engine.make_blob(m, si.key_joint_ids)
ctrl_seq, _ = bench.load_actions(m)
base = traj(m, blob, ctrl_seq, 700)
t = traj(m, blob, ctrl_seq, 700, eps=1e-15, seed=0)
r = rel(t, base)
prev = 0
for i in range(0, 700):
    if r[i] > 3 * max(prev, 1e-16) or i % 50 == 0:
        print(i, f"{r[i]:.2e}", "dof", int(np.argmax(np.abs(t[i]-base[i])/np.maximum(np.abs(base[i]),1e-2))))
    prev = max(prev, r[i])
