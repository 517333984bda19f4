"""The engine's OWN kernel sources (robopianist_amd/csrc/*.hpp, rp_engine.hip), compiled for the CPU wave
emulator in tests/wavesim and stepped against the oracle: a no-GPU check of the HIP code paths themselves
(lane roles, LDS hand-overs, DPP / readlane reductions, the solver) at the teacher-forced 1e-9 bar.
Test infrastructure only -- the product never loads the emulator build (engine.load_library opens
csrc/librp_engine.so; this test points RP_ENGINE_LIB at the emulator build in a subprocess)."""
import os
import re
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WS = os.path.join(HERE, "wavesim")


@pytest.fixture(scope="module")
def wavesim_lib():
    subprocess.check_call([os.path.join(WS, "build.sh")], stdout=subprocess.DEVNULL)
    return os.path.join(WS, "_build", "librp_engine_wavesim.so")


def _run(lib, *args):
    env = dict(os.environ, RP_ENGINE_LIB=lib, WAVESIM_SITE="0")
    out = subprocess.run([sys.executable, os.path.join(WS, "run_parity.py"), *args], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    m = re.search(r"worst rel dv ([0-9.e+-]+), max contacts (\d+)", out.stdout)
    assert m, out.stdout
    return float(m.group(1)), int(m.group(2))


@pytest.mark.parametrize("scenario,nsteps,mincon", [("random", 80, 4), ("wrist", 120, 6)])
def test_kernels_on_the_wave_emulator_match_the_oracle(wavesim_lib, scenario, nsteps, mincon):
    worst, maxcon = _run(wavesim_lib, str(nsteps), scenario)
    assert maxcon >= mincon, maxcon
    assert worst < 1e-9, worst


def test_fused_substeps_on_the_wave_emulator_match_the_per_stage_schedule(wavesim_lib):
    """rp_fused_steps_kernel / rp_cleanup_steps_kernel (all substeps of a step in one launch) against one launch
    per stage, both on the emulator: control steps of ten mj_steps with the sensor stage on, the light class capped
    so that envs change class mid-step (tests/wavesim/compare_fused.py)."""
    env = dict(os.environ, RP_ENGINE_LIB=wavesim_lib, WAVESIM_SITE="0")
    out = subprocess.run([sys.executable, os.path.join(WS, "compare_fused.py"), "4", "wild", "40", "sensors"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    m = re.search(r"fused vs per-stage ([0-9.e+-]+), fused vs oracle ([0-9.e+-]+)", out.stdout)
    assert m, out.stdout
    assert float(m.group(1)) < 1e-9 and float(m.group(2)) < 1e-9, out.stdout


def _run_parity_fn(lib, fn, *args, timeout=900):
    """Runs one scenario function of tests/test_gpu_parity.py (engine vs oracle) against the emulator build."""
    env = dict(os.environ, RP_ENGINE_LIB=lib, WAVESIM_SITE="0", RP_SKIP_SELF_CHECK="1")
    code = ("import sys, warnings; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_parity as t\nfrom robopianist_amd.model import scene\n"
            "warnings.simplefilter('ignore')\n"
            "si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)\n"
            "print('RESULT', *t.%s(si, %s))\n") % (os.path.dirname(HERE), HERE, fn, ", ".join(repr(a) for a in args))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return [float(x) for x in re.search(r"RESULT (.*)", out.stdout).group(1).split()]


def test_pile_up_beyond_32_contacts_on_the_wave_emulator(wavesim_lib):
    """Round 4's contact capacity (64 contacts / 640 contact Jacobian entries per env: overflow records behind the
    position stage's LDS staging, 64 contact lanes in the full-capacity solver stage) without a GPU: hand-in-hand
    poses, teacher-forced at 1e-9 (the GPU twin: test_teacher_forced_fp64_pile_up_beyond_32_contacts)."""
    worst, maxcon, maxent, beyond = _run_parity_fn(wavesim_lib, "pile_up", 60)
    assert maxcon > 36 and beyond >= 4 and maxent > 300, (maxcon, beyond, maxent)
    assert worst < 1e-9, worst


def test_legacy_step_false_on_the_wave_emulator(wavesim_lib):
    """rp_set_legacy_step(e, 0) (dm_control's legacy_step=False: mj_step order) without a GPU: same state, outputs of the
    state before the last integration (the GPU twin: test_legacy_step_false_publishes_the_outputs_of_mj_step)."""
    maxcon, checks = _run_parity_fn(wavesim_lib, "legacy_step_off", 70)
    assert checks == 350 and maxcon >= 2, (maxcon, checks)


def test_split_position_stage_on_the_wave_emulator(wavesim_lib):
    """The split position stage (front part / pooled narrow phase / back part, csrc/rp_collide.hpp) against the one-kernel
    stage on the emulator, bit for bit, hull fingertips along the replay (the GPU twin:
    test_split_position_stage_is_bit_identical)."""
    env = dict(os.environ, RP_ENGINE_LIB=wavesim_lib, WAVESIM_SITE="0", RP_SKIP_SELF_CHECK="1")
    out = subprocess.run([sys.executable, os.path.join(WS, "compare_split.py"), "60", "hull", "replay", "400"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "bit for bit" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_box_box_face_contacts_on_the_wave_emulator(wavesim_lib):
    """Box-box pairs with four to eight points (the palm flat on the keys) on the emulator."""
    worst, maxcon, pairs, most = _run_parity_fn(wavesim_lib, "palm_flat", 12)
    assert pairs >= 20 and most >= 4, (pairs, most)
    assert worst < 1e-9, worst


def test_large_hull_colliders_on_the_wave_emulator(wavesim_lib):
    """Every hand collider a 200-vertex hull (the graph walk of model/hull.py, MESH = 2 kernel builds) on the emulator."""
    env = dict(os.environ, RP_ENGINE_LIB=wavesim_lib, WAVESIM_SITE="0", RP_SKIP_SELF_CHECK="1")
    code = ("import sys, warnings; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_parity as t\nfrom robopianist_amd.model import scene\n"
            "warnings.simplefilter('ignore')\n"
            "si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=False, mesh_colliders=200)\n"
            "print('RESULT', *t.teacher_forced(si, 64, t._replay_ctrl(si)[400:480]))\n") % (os.path.dirname(HERE), HERE)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    worst, maxcon = [float(x) for x in re.search(r"RESULT (.*)", out.stdout).group(1).split()]
    assert maxcon >= 4 and worst < 1e-9, (worst, maxcon)


def test_impratio_and_cylinder_colliders_on_the_wave_emulator(wavesim_lib):
    """Round 6 without a GPU: opt.impratio = 10 in the friction regularisation and mjGEOM_CYLINDER colliders through the
    portal refinement (MESH = 2 kernel builds), the engine's own kernels against the oracle at the teacher-forced 1e-9
    (the GPU twins: test_teacher_forced_fp64_impratio, test_teacher_forced_fp64_cylinder_colliders)."""
    env = dict(os.environ, RP_ENGINE_LIB=wavesim_lib, WAVESIM_SITE="0", RP_SKIP_SELF_CHECK="1")
    code = ("import sys, warnings; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_parity as t\nfrom robopianist_amd.model import scene\n"
            "warnings.simplefilter('ignore')\n"
            "si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True, impratio=10.0)\n"
            "print('IMPRATIO', *t.teacher_forced(si, 64, t._replay_ctrl(si)[520:600]))\n"
            "si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True, cylinder_colliders=True, impratio=10.0)\n"
            "st = {}\n"
            "r = t.pile_up(si, 30, lo_dx=0.07, hi_dx=0.09, stable_only=True, stats=st)\n"
            "print('CYLINDER', r[0], st.get('compared', 0), sum(v for k, v in st.items() if isinstance(k, tuple) and 5 in k))\n"
            ) % (os.path.dirname(HERE), HERE)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(HERE))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    worst, maxcon = [float(x) for x in re.search(r"IMPRATIO (.*)", out.stdout).group(1).split()]
    assert maxcon >= 6 and worst < 1e-9, (worst, maxcon)
    worst, compared, cyl = [float(x) for x in re.search(r"CYLINDER (.*)", out.stdout).group(1).split()]
    assert compared >= 10 and cyl >= 8 and worst < 1e-9, (worst, compared, cyl)
