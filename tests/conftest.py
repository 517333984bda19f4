import os
import sys
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


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
