"""Which geom pairs are in contact along the replay (oracle), and how often."""
import sys, os, warnings
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from collections import Counter
from robopianist_amd.model import scene
from robopianist_amd import engine
from oracle.rp_oracle import Oracle
import bench
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=False)
m = si.model
o = Oracle(m, engine.make_blob(m, si.key_joint_ids)); o.reset()
ctrl, _ = bench.load_actions(m)
gm = m.names["geom"]
cnt = Counter(); steps = 0; ncs = []
o.forward()
print("contacts at the reset pose:", [(gm[int(c[13])].split('/')[-1], gm[int(c[14])].split('/')[-1], round(c[0], 5)) for c in o.contact.reshape(-1, 16)])
for t in range(ctrl.shape[0]):
    o.ctrl[:] = ctrl[t]
    for k in range(10):
        o.step(1); steps += 1
        con = o.contact.reshape(-1, 16); ncs.append(len(con))
        for c in set((gm[int(c[13])].split('/')[-1], gm[int(c[14])].split('/')[-1]) for c in con):
            cnt[c] += 1
print("mj_steps", steps, "mean contacts", np.mean(ncs))
for (a, b), n in cnt.most_common(40):
    print(f"{a:22s} {b:22s} {n / steps:6.3f}")
