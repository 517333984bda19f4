"""TEST INFRASTRUCTURE: lean solver stage vs full-capacity solver stage, both on the CPU wave emulator, teacher-forced
from the oracle's trajectory.  Prints the worst per-step difference between the two builds and against the oracle."""
import os, sys, time, warnings
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
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    scen = sys.argv[2] if len(sys.argv) > 2 else "random"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
    ctrl = {"random": lambda: tgp.ctrl_sequence(si.model, nsteps, 1), "press": lambda: tgp.key_press_sequence(si, nsteps),
            "wrist": lambda: tgp.wrist_press_sequence(si, nsteps), "wild": lambda: tgp.ctrl_sequence(si.model, nsteps, 7, hold=5, lo_frac=0.0, hi_frac=1.0)}[scen]()
    a = engine.BatchedPhysics(si.model, si.key_joint_ids, n_envs=1, precision=64)
    b = engine.BatchedPhysics(si.model, si.key_joint_ids, n_envs=1, precision=64, blob=a.blob)
    a.set_lean_solver(True); b.set_lean_solver(False)
    orc = Oracle(si.model, a.blob)
    worst_ab = worst_ao = 0.0; nd_steps = nk_steps = 0; maxnd = maxcon = light = 0
    for c in ctrl:
        for p in (a, b):
            p.set(engine.QPOS, orc.qpos[None, :]); p.set(engine.QVEL, orc.qvel[None, :])
            p.set(engine.QACC_WARMSTART, orc.qacc_warmstart[None, :]); p.set(engine.CTRL, c[None, :])
        orc.ctrl[:] = c
        v0 = orc.qvel.copy()
        a.step(1); b.step(1); orc.step(1)
        it = int(a.get(engine.SOLVER_ITER)[0]); itb = int(b.get(engine.SOLVER_ITER)[0])
        assert it == itb, (it & 255, itb & 255)
        nd = (it >> 8) & 255; nkt = (it >> 16) & 255
        nd_steps += nd > 0; nk_steps += nkt > 0; maxnd = max(maxnd, nd); maxcon = max(maxcon, int(a.get(engine.NCON)[0]))
        light += int(a.get(engine.DEBUG_HANDOVER_HDR)[0, 6] == 1)
        den = max(np.abs(orc.qvel - v0).max(), 1e-9)
        worst_ab = max(worst_ab, np.abs(a.qvel[0] - b.qvel[0]).max() / den)
        worst_ao = max(worst_ao, np.abs(a.qvel[0] - orc.qvel).max() / den)
        assert a.warn_flags.max() == 0 and b.warn_flags.max() == 0
    print(f"{scen} {nsteps} steps: lean vs full {worst_ab:.2e}, lean vs oracle {worst_ao:.2e}; steps with dense rows {nd_steps} (max {maxnd}), "
          f"with touched keys {nk_steps}, max contacts {maxcon}, light steps {light}")

if __name__ == "__main__":
    main()
