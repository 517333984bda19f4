"""The C-ABI library loads and exports every symbol include/*.h declares;
without a GPU the product path fails loudly (no CPU fallback)."""
import ctypes
import os
import re
import sys

import pytest
import torch

from robopianist_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    out = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not h.endswith(".h"):
            continue
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        out |= set(re.findall(r"\b(rp_[a-z_0-9]+)\s*\(", src))
    return sorted(out)


def test_header_and_binding_agree():
    from robopianist_amd import task_kernels
    assert _declared_symbols() == sorted(tuple(engine.EXPORTED_SYMBOLS) + tuple(task_kernels.EXPORTED_SYMBOLS))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(engine.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} missing from librp_engine.so"


def test_bad_arguments_are_rejected_without_touching_a_device():
    L = engine.load_library()
    h = ctypes.c_void_p()
    assert L.rp_create(b"xxxx", 4, 1, 0, 32, ctypes.byref(h)) != 0
    assert b"blob" in L.rp_last_error() or b"magic" in L.rp_last_error()
    assert L.rp_create(None, 0, 1, 0, 32, ctypes.byref(h)) != 0
    assert L.rp_create(b"xxxx", 4, 1, 0, 16, ctypes.byref(h)) != 0
    assert b"precision" in L.rp_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure(piano_only_scene):
    with pytest.raises(engine.EngineError, match="no HIP device|HIP"):
        engine.BatchedPhysics(piano_only_scene.model, piano_only_scene.key_joint_ids, n_envs=2)
    from robopianist_amd.suite.physics import TorchPhysics
    with pytest.raises(engine.EngineError):
        TorchPhysics(piano_only_scene, 2)


def test_key_trace_decoding():
    import numpy as np
    t = np.zeros((1, 2, 4), np.uint32)
    t[0, 0, 0] = 0b101
    t[0, 1, 2] = 1 << 3  # key 64 + 3
    bits = engine.decode_key_trace(t)
    assert bits.shape == (1, 2, 88)
    assert bits[0, 0].nonzero()[0].tolist() == [0, 2]
    assert bits[0, 1].nonzero()[0].tolist() == [67]


def test_task_abi_rejects_bad_arguments_without_a_device():
    """include/rp_task.h: argument errors are reported through the return code /
    rp_task_last_error before anything is launched."""
    from robopianist_amd import task_kernels as tk
    L = tk._lib()
    assert L.rp_task_advance(None, None) == -1 and b"null args" in L.rp_task_last_error()
    assert L.rp_task_rewards(None, None) == -1
    a = tk.RewardArgs()
    a.n_envs, a.precision = 4, 16
    assert L.rp_task_rewards(ctypes.byref(a), None) == -1 and b"precision" in L.rp_task_last_error()
    a.precision, a.n_envs = 64, 0
    assert L.rp_task_rewards(ctypes.byref(a), None) == -1 and b"n_envs" in L.rp_task_last_error()
    a.n_envs, a.hand_filter = 4, 3
    assert L.rp_task_rewards(ctypes.byref(a), None) == -1 and b"hand_filter" in L.rp_task_last_error()
    a.hand_filter = 0
    assert L.rp_task_rewards(ctypes.byref(a), None) == -1 and b"null array" in L.rp_task_last_error()
    p = tk.AdvanceArgs()
    p.rw = a
    p.bank_len = 0
    assert L.rp_task_advance(ctypes.byref(p), None) == -1 and b"bad sizes" in L.rp_task_last_error()
    r = tk.RasterArgs()
    assert L.rp_task_rasterize(None, None) == -1
    r.precision = 8
    assert L.rp_task_rasterize(ctypes.byref(r), None) == -1 and b"precision" in L.rp_task_last_error()
    r.precision, r.n_songs, r.bank_len, r.fps = 64, 1, 16, 0.0
    assert L.rp_task_rasterize(ctypes.byref(r), None) == -1 and b"bad sizes" in L.rp_task_last_error()
    r.fps = 20.0
    assert L.rp_task_rasterize(ctypes.byref(r), None) == -1 and b"null array" in L.rp_task_last_error()


def test_task_abi_struct_layout_matches_the_header():
    """The ctypes mirrors in task_kernels.py must list the header's fields in order (a
    mismatch would silently shift every pointer after it)."""
    from robopianist_amd import task_kernels as tk
    src = open(os.path.join(ROOT, "include", "rp_task.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)

    def fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), src, flags=re.S).group(1)
        names = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            for part in stmt.split(","):
                names.append(re.findall(r"[A-Za-z_][A-Za-z_0-9]*", part)[-1])
        return names

    assert fields("rp_task_reward_args") == [f[0] for f in tk.RewardArgs._fields_]
    assert fields("rp_task_advance_args") == [f[0] for f in tk.AdvanceArgs._fields_]
    assert fields("rp_task_raster_args") == [f[0] for f in tk.RasterArgs._fields_]


def _build_c_demo(tmp_path):
    import subprocess
    exe = str(tmp_path / "c_abi_demo")
    csrc = os.path.join(ROOT, "robopianist_amd", "csrc")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-o", exe,
                           "-L" + csrc, "-lrp_engine", "-Wl,-rpath," + csrc])
    blob = str(tmp_path / "scene.blob")
    subprocess.check_call([sys.executable, "-m", "robopianist_amd.tools.dump_blob", blob, "--gravity_compensation"],
                          cwd=ROOT)
    return exe, blob


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_plain_c_host_links_against_the_abi_and_fails_loudly_without_a_gpu(tmp_path):
    """examples/c_abi_demo.c uses include/rp_engine.h only (gcc, no torch, no Python at run
    time): the boundary is a plain C ABI."""
    import subprocess
    exe, blob = _build_c_demo(tmp_path)
    r = subprocess.run([exe, blob, "4", "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_plain_c_host_reproduces_the_python_binding_bitwise(tmp_path):
    import subprocess
    import warnings
    import numpy as np
    from robopianist_amd.model import scene
    exe, blob = _build_c_demo(tmp_path)
    r = subprocess.run([exe, blob, "8", "5"], capture_output=True, text=True, check=True)
    lines = r.stdout.strip().split("\n")
    assert lines[0].startswith("nv 140 nu 44 envs 8 steps 5 warn 0")
    q_c = np.array([float(x) for x in lines[1].split()])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
    phys = engine.BatchedPhysics(si.model, si.key_joint_ids, n_envs=8, precision=64)
    ctrl = (0.2 + 0.01 * (np.arange(8 * 44) % 44)).reshape(8, 44)
    phys.reset()
    phys.set(engine.CTRL, ctrl)
    for _ in range(5):
        phys.step(10)
    assert np.array_equal(phys.qpos[0].astype(np.float64), q_c)


@pytest.mark.gpu
def test_engine_created_before_torch_touches_the_gpu():
    """A fresh process that builds the engine first and uses torch's GPU side afterwards: torch's bundled
    HIP runtime must be the one the engine binds to (robopianist_amd/engine.py: load_library)."""
    import subprocess
    import sys
    code = (
        "import warnings; warnings.simplefilter('ignore')\n"
        "from robopianist_amd import engine\n"
        "from robopianist_amd.model import scene\n"
        "si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)\n"
        "p = engine.BatchedPhysics(si.model, si.key_joint_ids, n_envs=4, precision=64)\n"
        "p.step(2)\n"
        "import torch\n"
        "v = p.view(engine.QPOS)\n"
        "assert v.is_cuda and bool(torch.isfinite(v).all()) and float(torch.ones(3, device='cuda').sum()) == 3.0\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-1500:]
