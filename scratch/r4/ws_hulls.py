"""Teacher-forced parity of a hand whose colliders all are ~n-vertex hulls, on the CPU wave emulator."""
import os, sys, warnings, time
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("RP_ENGINE_LIB", os.path.join(ROOT, "tests", "wavesim", "_build", "librp_engine_wavesim.so"))
os.environ["RP_SKIP_SELF_CHECK"] = "1"; os.environ.setdefault("WAVESIM_SITE", "0")
import test_gpu_parity as tgp
from robopianist_amd.model import scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=False, mesh_colliders=n)
t0 = time.time()
worst, maxcon = tgp.teacher_forced(si, 64, tgp._replay_ctrl(si)[400:400 + steps])
print(f"large hulls ({n} vertices): worst rel dv {worst:.2e}, max contacts {maxcon}, {time.time() - t0:.1f} s")
