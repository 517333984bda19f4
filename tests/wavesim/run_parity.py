"""TEST INFRASTRUCTURE: teacher-forced / free-running parity of the engine's kernels, run on the CPU wave
emulator (tests/wavesim), against the oracle.  Usage: python tests/wavesim/run_parity.py [nsteps] [scenario]"""
import os, sys, time, warnings
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("RP_ENGINE_LIB", os.path.join(HERE, "_build", "librp_engine_wavesim.so"))
os.environ["RP_SKIP_SELF_CHECK"] = "1"
import numpy as np
import test_gpu_parity as tgp
from robopianist_amd.model import scene

def main():
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    scen = sys.argv[2] if len(sys.argv) > 2 else "random"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
    if scen == "random":
        ctrl = tgp.ctrl_sequence(si.model, nsteps, 1)
    elif scen == "press":
        ctrl = tgp.key_press_sequence(si, nsteps)
    else:
        ctrl = tgp.wrist_press_sequence(si, nsteps)
    t0 = time.time()
    worst, maxcon = tgp.teacher_forced(si, 64, ctrl)
    print(f"teacher-forced fp64 ({scen}, {nsteps} steps): worst rel dv {worst:.2e}, max contacts {maxcon}, {time.time()-t0:.1f} s")

if __name__ == "__main__":
    main()
