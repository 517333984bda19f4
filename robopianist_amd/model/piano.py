"""88-key piano scene fragment.

Numbers restate the reference's piano exactly:
  constants  robopianist/models/piano/piano_constants.py:22-85
  layout     robopianist/models/piano/piano_mjcf.py:25-402
The piano is fully specified in the reference tree, so this half of the model
is exact (unlike the Shadow Hand, see `shadow_hand.py`).
"""

from __future__ import annotations

import math
from typing import List

from robopianist_amd.model import spec

# piano_constants.py:22-36
NUM_KEYS = 88
NUM_WHITE_KEYS = 52
WHITE_KEY_WIDTH = 0.0225
WHITE_KEY_LENGTH = 0.15
WHITE_KEY_HEIGHT = WHITE_KEY_WIDTH
SPACING_BETWEEN_WHITE_KEYS = 0.001
N_SPACES_BETWEEN_WHITE_KEYS = NUM_WHITE_KEYS - 1
BLACK_KEY_WIDTH = 0.01
BLACK_KEY_LENGTH = 0.09
BLACK_KEY_HEIGHT = 0.018
PIANO_LENGTH = (NUM_WHITE_KEYS * WHITE_KEY_WIDTH) + (
    N_SPACES_BETWEEN_WHITE_KEYS * SPACING_BETWEEN_WHITE_KEYS
)
# piano_constants.py:38-50
WHITE_KEY_X_OFFSET = 0.0
WHITE_KEY_Z_OFFSET = WHITE_KEY_HEIGHT / 2
BLACK_KEY_X_OFFSET = -WHITE_KEY_LENGTH / 2 + BLACK_KEY_LENGTH / 2
BLACK_OFFSET_FROM_WHITE = 0.0125
BLACK_KEY_Z_OFFSET = WHITE_KEY_HEIGHT + BLACK_OFFSET_FROM_WHITE - BLACK_KEY_HEIGHT / 2
BASE_HEIGHT = 0.04
BASE_LENGTH = 0.1
BASE_WIDTH = PIANO_LENGTH
BASE_SIZE = (BASE_LENGTH / 2, BASE_WIDTH / 2, BASE_HEIGHT / 2)
BASE_X_OFFSET = -WHITE_KEY_LENGTH / 2 - 0.5 * BASE_LENGTH - 0.002
BASE_POS = (BASE_X_OFFSET, 0.0, BASE_HEIGHT / 2)
# piano_constants.py:52-76
WHITE_KEY_TRAVEL_DISTANCE = 0.01
WHITE_KEY_JOINT_MAX_ANGLE = math.atan(WHITE_KEY_TRAVEL_DISTANCE / WHITE_KEY_LENGTH)
BLACK_KEY_TRAVEL_DISTANCE = 0.008
BLACK_KEY_JOINT_MAX_ANGLE = math.atan(BLACK_KEY_TRAVEL_DISTANCE / BLACK_KEY_LENGTH)
WHITE_KEY_MASS = 0.04
BLACK_KEY_MASS = 0.02
KEY_SPRINGREF_DEG = -1.0
KEY_STIFFNESS = 2.0
KEY_DAMPING = 0.05
KEY_ARMATURE = 0.001

# Pattern of black keys within an octave starting at A0 (piano.py:172-176).
_BLACK_PATTERN = (0, 1, 0, 0, 1, 0, 1, 0, 0, 1, 0, 1)


def is_key_black(key_id: int) -> bool:
    """Mirror of `Piano.is_key_black` (piano.py:173-176)."""
    return bool(_BLACK_PATTERN[key_id % 12])


WHITE_KEY_INDICES = [k for k in range(NUM_KEYS) if not is_key_black(k)]
BLACK_KEY_INDICES = [k for k in range(NUM_KEYS) if is_key_black(k)]


def key_positions() -> List[tuple]:
    """Body position of every key, indexed by key id (piano_mjcf.py:168-400)."""
    pitch = WHITE_KEY_WIDTH + SPACING_BETWEEN_WHITE_KEYS
    pos = {}
    # White keys (piano_mjcf.py:168-179).
    for i in range(NUM_WHITE_KEYS):
        y = -PIANO_LENGTH * 0.5 + WHITE_KEY_WIDTH * 0.5 + i * pitch
        pos[WHITE_KEY_INDICES[i]] = (WHITE_KEY_X_OFFSET, y, WHITE_KEY_Z_OFFSET)
    # Lone black key on the far left, key id 1 (piano_mjcf.py:248-257).
    y = WHITE_KEY_WIDTH + 0.5 * (-PIANO_LENGTH + SPACING_BETWEEN_WHITE_KEYS)
    pos[1] = (BLACK_KEY_X_OFFSET, y, BLACK_KEY_Z_OFFSET)
    # Twin black keys (piano_mjcf.py:283-295): white index 2, 9, 16, ...
    twins = [4, 6, 16, 18, 28, 30, 40, 42, 52, 54, 64, 66, 76, 78]
    n = 0
    for twin_index in range(2, NUM_WHITE_KEYS - 1, 7):
        for j in range(2):
            y = -PIANO_LENGTH * 0.5 + (j + 1) * pitch + twin_index * pitch
            pos[twins[n]] = (BLACK_KEY_X_OFFSET, y, BLACK_KEY_Z_OFFSET)
            n += 1
    # Triplet black keys (piano_mjcf.py:332-344): white index 5, 12, 19, ...
    triplets = [9, 11, 13, 21, 23, 25, 33, 35, 37, 45, 47, 49, 57, 59, 61,
                69, 71, 73, 81, 83, 85]
    n = 0
    for triplet_index in range(5, NUM_WHITE_KEYS - 1, 7):
        for j in range(3):
            y = -PIANO_LENGTH * 0.5 + (j + 1) * pitch + triplet_index * pitch
            pos[triplets[n]] = (BLACK_KEY_X_OFFSET, y, BLACK_KEY_Z_OFFSET)
            n += 1
    assert sorted(pos) == list(range(NUM_KEYS))
    return [pos[k] for k in range(NUM_KEYS)]


def key_qpos_max() -> List[float]:
    return [
        BLACK_KEY_JOINT_MAX_ANGLE if is_key_black(k) else WHITE_KEY_JOINT_MAX_ANGLE
        for k in range(NUM_KEYS)
    ]


def build(add_actuators: bool = False, physics_timestep: float = 0.005):
    """Returns (base_body, [key bodies], [actuators]).

    `physics_timestep` sets the piano geoms' solref = (2*dt, 1), which is what
    `PianoOnlyTask.__init__` does (robopianist/suite/tasks/base.py:59-66).
    """
    solref = (physics_timestep * 2, 1.0)
    # contype=0 / conaffinity=1: piano_mjcf.py:48-53.
    geom_kw = dict(contype=0, conaffinity=1, solref=solref)
    base = spec.Body(
        name="piano/base",
        pos=BASE_POS,
        geoms=[spec.Geom("piano/base_geom", spec.GEOM_BOX, BASE_SIZE, **geom_kw)],
    )
    keys = []
    actuators = []
    positions = key_positions()
    for k in range(NUM_KEYS):
        black = is_key_black(k)
        if black:
            size = (BLACK_KEY_LENGTH / 2, BLACK_KEY_WIDTH / 2, BLACK_KEY_HEIGHT / 2)
            mass = BLACK_KEY_MASS
            qmax = BLACK_KEY_JOINT_MAX_ANGLE
            prefix = "black"
        else:
            size = (WHITE_KEY_LENGTH / 2, WHITE_KEY_WIDTH / 2, WHITE_KEY_HEIGHT / 2)
            mass = WHITE_KEY_MASS
            qmax = WHITE_KEY_JOINT_MAX_ANGLE
            prefix = "white"
        joint = spec.Joint(
            name=f"piano/{prefix}_joint_{k}",
            type=spec.JNT_HINGE,
            pos=(-size[0], 0.0, 0.0),
            axis=(0.0, 1.0, 0.0),
            range=(0.0, qmax),
            stiffness=KEY_STIFFNESS,
            springref=KEY_SPRINGREF_DEG * math.pi / 180,
            damping=KEY_DAMPING,
            armature=KEY_ARMATURE,
        )
        body = spec.Body(
            name=f"piano/{prefix}_key_{k}",
            pos=positions[k],
            joints=[joint],
            geoms=[
                spec.Geom(
                    f"piano/{prefix}_key_geom_{k}", spec.GEOM_BOX, size, mass=mass,
                    **geom_kw,
                )
            ],
            sites=[spec.Site(f"piano/{prefix}_key_site_{k}")],
        )
        keys.append(body)
        if add_actuators:
            # general actuator, gain 1, no bias, ctrlrange [0, qmax]
            # (piano_mjcf.py:56-62,80-81,99-100).
            actuators.append(
                spec.Actuator(
                    name=f"piano/{prefix}_actuator_{k}",
                    joint=joint.name,
                    gain=1.0,
                    bias=(0.0, 0.0, 0.0),
                    ctrlrange=(0.0, qmax),
                )
            )
    return base, keys, actuators
