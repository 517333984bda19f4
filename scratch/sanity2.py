import warnings; warnings.simplefilter('ignore')
import sys, os; sys.path.insert(0,'/root/repo'); os.environ["RP_SKIP_SELF_CHECK"]="1"
import numpy as np
from robopianist_amd import engine
from robopianist_amd.model import scene
from oracle.rp_oracle import Oracle
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
m = si.model
lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
ctrl = (lo + 0.6 * (hi - lo))
blob = engine.make_blob(m, si.key_joint_ids)
orc = Oracle(m, blob); orc.ctrl[:] = ctrl; orc.step(6)
for mode in ("step6", "6xstep1"):
    for prec in (64, 32):
        e = engine.BatchedPhysics(m, si.key_joint_ids, 2, precision=prec, blob=blob, self_check=False)
        print("   after create: |qvel|max %.2e |warm|max %.2e |qpos-qpos0| %.2e time %s qfrc %.2e" % (
            np.abs(e.get(engine.QVEL)).max(), np.abs(e.get(engine.QACC_WARMSTART)).max(),
            np.abs(e.qpos.astype(np.float64)[0]-m.qpos0).max(), e.get(engine.TIME), np.abs(e.get(engine.QFRC_APPLIED)).max()))
        e.set(engine.CTRL, ctrl[None, :])
        if mode == "step6": e.step(6)
        else:
            for _ in range(6): e.step(1)
        q = e.qpos.astype(np.float64)
        import time as _t; _t.sleep(0.3)
        q2 = e.qpos.astype(np.float64)
        print("   time field", e.get(engine.TIME), "re-read differs by %.2e" % np.abs(q2-q).max(), "ctrl on device", float(np.abs(e.get(engine.CTRL)[0]-ctrl).max()))
        print(mode, prec, "max|dq| vs oracle %.2e" % np.abs(q[0]-orc.qpos).max(), "env1-env0 %.1e" % np.abs(q[1]-q[0]).max(), "warn", e.warn_flags.max())
        del e
