import os, sys, warnings
warnings.simplefilter("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robopianist_amd import engine
from robopianist_amd.model import scene
from oracle.rp_oracle import Oracle
scenes = {
 "3dof reduced": dict(reduced_action_space=True, forearm_dofs=("forearm_tx", "forearm_ty", "forearm_yaw")),
 "5dof left": dict(hands=("left",), forearm_dofs=("forearm_ty", "forearm_tz", "forearm_roll", "forearm_pitch", "forearm_yaw")),
 "6dof": dict(forearm_dofs=("forearm_tx", "forearm_ty", "forearm_tz", "forearm_roll", "forearm_pitch", "forearm_yaw")),
}
for name, kw in scenes.items():
    si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True, **kw)
    m = si.model
    for prec in (64, 32):
        phys = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=2, precision=prec, self_check=False)
        orc = Oracle(m, phys.blob)
        rng = np.random.default_rng(0)
        lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
        c = lo + rng.uniform(0.2, 0.8, m.nu) * (hi - lo)
        phys.set(engine.CTRL, c[None, :]); orc.ctrl[:] = c
        phys.forward(); 
        sx = phys.get(engine.SITE_XPOS)[0]
        print(name, prec, "after forward: site_xpos finite", np.isfinite(sx).all(), "ncon", phys.get(engine.NCON)[0], orc.ncon, "warn", phys.warn_flags[0])
        for k in range(3):
            phys.step(1); orc.step(1)
            q = phys.qpos[0].astype(np.float64); v = phys.get(engine.QVEL)[0].astype(np.float64)
            print("   step", k, "finite", np.isfinite(q).all(), "max|dq|", np.nanmax(np.abs(q - orc.qpos)), "max|dv|", np.nanmax(np.abs(v - orc.qvel)),
                  "warn", phys.warn_flags[0], "nan dofs", np.flatnonzero(~np.isfinite(v))[:10], "iter", phys.get(engine.SOLVER_ITER)[0] & 255, orc.solver_iter)

# bisect: the deep builds on the standard scene (4-link trunks)
os.environ["RP_FORCE_DEEP"] = "1"
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
m = si.model
for prec in (64,):
    phys = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=2, precision=prec, self_check=False)
    orc = Oracle(m, phys.blob)
    rng = np.random.default_rng(0)
    lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
    c = lo + rng.uniform(0.2, 0.8, m.nu) * (hi - lo)
    phys.set(engine.CTRL, c[None, :]); orc.ctrl[:] = c
    for k in range(3):
        phys.step(1); orc.step(1)
        q = phys.qpos[0].astype(np.float64); v = phys.get(engine.QVEL)[0].astype(np.float64)
        print("FORCED DEEP on the standard scene: step", k, "max|dq|", np.abs(q - orc.qpos).max(), "max|dv|", np.abs(v - orc.qvel).max(), "iter", phys.get(engine.SOLVER_ITER)[0] & 255, orc.solver_iter)
