import os
import sys
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _hip_device_present() -> bool:
    """True when a HIP device can be opened (the engine has no CPU fallback: without one rp_create fails)."""
    if not os.path.exists("/dev/kfd"):
        return False
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return False


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests SKIP (instead of failing in rp_create) on a machine without a HIP device, so `pytest tests`
    without `-m` is green on a CPU-only box.  On the GPU box nothing is skipped."""
    if not any("gpu" in item.keywords for item in items) or _hip_device_present():
        return
    skip = pytest.mark.skip(reason="no HIP device on this machine (the engine has no CPU fallback)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def two_hand_scene():
    from robopianist_amd.model import scene
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)


@pytest.fixture(scope="session")
def piano_only_scene():
    from robopianist_amd.model import scene
    return scene.build_scene(hands=(), add_piano_actuators=True)
