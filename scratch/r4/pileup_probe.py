"""Find a hand-on-hand pose with more than 32 contacts (oracle only)."""
import sys, os, warnings
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from robopianist_amd.model import scene
from robopianist_amd import engine
from oracle.rp_oracle import Oracle
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
m = si.model
blob = engine.make_blob(m, si.key_joint_ids)
o = Oracle(m, blob); o.reset()
jn = m.names["joint"]
print([ (i, n) for i, n in enumerate(jn) if "forearm" in n or "WRJ" in n])
q0 = o.qpos.copy()
print("qpos0 forearm:", [(n, q0[i]) for i, n in enumerate(jn) if "forearm" in n])
idx = {n: i for i, n in enumerate(jn)}
for dx in np.linspace(0.0, 0.5, 26):
    o.reset()
    for n, i in idx.items():
        if n.endswith("forearm_tx"):
            o.qpos[i] = q0[i] + (dx if n.startswith("rh") else -dx) * (1 if len(sys.argv) < 2 else float(sys.argv[1]))
    o.step(1)
    c = o.contact.reshape(-1, 16)
    print(f"dx {dx:.3f} ncon {o.ncon} warnings {o.warnings}")
