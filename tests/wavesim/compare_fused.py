"""TEST INFRASTRUCTURE: fused substeps (one launch per rp_step) vs one launch per stage, both on the CPU wave
emulator: control steps of 10 mj_steps, restarted from the oracle's state.  Optional argument 3: cap of the light
class (rp_set_lean_solver(e, n)), so that envs leave it mid-step and finish in the clean-up kernel."""
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
from oracle.rp_oracle import Oracle

def main():
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    scen = sys.argv[2] if len(sys.argv) > 2 else "random"
    cap = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    sensors = len(sys.argv) > 4 and sys.argv[4] == "sensors"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
    ctrl = {"random": lambda: tgp.ctrl_sequence(si.model, nsteps, 1), "press": lambda: tgp.key_press_sequence(si, nsteps),
            "wrist": lambda: tgp.wrist_press_sequence(si, nsteps), "wild": lambda: tgp.ctrl_sequence(si.model, nsteps, 7, hold=5, lo_frac=0.0, hi_frac=1.0)}[scen]()
    E = 3
    a = engine.BatchedPhysics(si.model, si.key_joint_ids, n_envs=E, precision=64)
    b = engine.BatchedPhysics(si.model, si.key_joint_ids, n_envs=E, precision=64, blob=a.blob)
    a.set_lean_solver(cap); b.set_lean_solver(cap)
    a.set_fused_substeps(True); b.set_fused_substeps(False)
    assert a.fused_substeps and not b.fused_substeps
    if sensors:
        a.set_acc_sensors(True); b.set_acc_sensors(True)
    orc = Oracle(si.model, a.blob)
    gain = np.array([1.0, 0.9, 1.1])[:, None]
    worst = worst_o = 0.0; bailed = 0
    for c in ctrl:
        for p in (a, b):
            p.set(engine.QPOS, np.repeat(orc.qpos[None, :], E, 0)); p.set(engine.QVEL, np.repeat(orc.qvel[None, :], E, 0))
            p.set(engine.QACC_WARMSTART, np.repeat(orc.qacc_warmstart[None, :], E, 0)); p.set(engine.CTRL, c[None, :] * gain)
        orc.ctrl[:] = c
        v0 = orc.qvel.copy()
        a.step(10); b.step(10); orc.step(10)
        bailed += int((a.get(engine.DEBUG_HANDOVER_HDR)[:, 7] < 10).sum()) if cap > 1 else 0
        den = max(np.abs(orc.qvel - v0).max(), 1e-9)
        worst = max(worst, np.abs(a.qvel - b.qvel).max() / den, np.abs(a.qpos - b.qpos).max())
        worst_o = max(worst_o, np.abs(a.qvel[0] - orc.qvel).max() / den)
        if sensors:
            worst = max(worst, np.abs(a.get(engine.SENSOR_TORQUE) - b.get(engine.SENSOR_TORQUE)).max(), np.abs(a.get(engine.SENSOR_TOUCH) - b.get(engine.SENSOR_TOUCH)).max())
        assert np.array_equal(a.get(engine.NCON), b.get(engine.NCON))
        assert a.warn_flags.max() == 0 and b.warn_flags.max() == 0
    print(f"{scen} {nsteps} control steps, cap {cap}{' sensors' if sensors else ''}: fused vs per-stage {worst:.2e}, fused vs oracle {worst_o:.2e}")

if __name__ == "__main__":
    main()
