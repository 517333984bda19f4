"""How often a Newton iteration re-factors an UNCHANGED Hessian (same quadratic-active set as the iteration before).
Needs the diagnostic hook rpo_debug_newton_stats (a counter in newton_direction comparing efc_state == 1 with the previous
iteration's), which is NOT in the committed oracle (global counters race under its OpenMP bench).  Result on the config-2
replay: 7727 direction solves (4.89 per mj_step), 402 of them (5.2 %) on an unchanged active set -- reusing the factor across
iterations is not worth building."""
import sys, os, warnings, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from robopianist_amd.model import scene
from robopianist_amd import engine
from oracle.rp_oracle import Oracle, lib
import bench
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=False)
m = si.model
o = Oracle(m, engine.make_blob(m, si.key_joint_ids)); o.reset()
ctrl, _ = bench.load_actions(m)
L = lib(); out = (ctypes.c_longlong * 4)()
L.rpo_debug_newton_stats(out, 1)
for t in range(ctrl.shape[0]):
    o.ctrl[:] = ctrl[t]
    o.step(10)
L.rpo_debug_newton_stats(out, 0)
tot, same, first = out[0], out[1], out[2]
print(f"Newton direction solves {tot} ({tot / (10 * ctrl.shape[0]):.2f} per mj_step), with the active set of the previous iteration: {same} ({same / tot:.3f}); first iterations {first}")
