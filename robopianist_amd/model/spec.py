"""Plain-Python scene description (the role MJCF plays for the reference).

A `Scene` is a tree of `Body` objects carrying joints, geoms, sites, plus
tendons, actuators and options.  It is consumed by `compile.compile_scene`,
which flattens it to the struct-of-arrays "model blob" that both the CPU
oracle (`oracle/`) and the HIP engine (`robopianist_amd/csrc/`) read.

The reference builds the same information through dm_control's PyMJCF
(`robopianist/models/piano/piano_mjcf.py:25-402`,
`robopianist/models/hands/shadow_hand.py:91-311`) and lets MuJoCo compile it.
"""

from __future__ import annotations

import dataclasses
from typing import List, Optional, Sequence, Tuple

import numpy as np

# Geom type ids follow MuJoCo's mjtGeom enum so that pair ordering
# (geom1.type <= geom2.type) matches the reference engine's convention.
GEOM_PLANE = 0
GEOM_SPHERE = 2
GEOM_CAPSULE = 3
GEOM_CYLINDER = 5  # size = (radius, half height); collides through mjc_Convex (support function), like hulls
GEOM_BOX = 6
GEOM_MESH = 7  # a convex hull given by its vertices (MuJoCo collides meshes through their hulls)

JNT_SLIDE = 2  # mjJNT_SLIDE
JNT_HINGE = 3  # mjJNT_HINGE

TRN_JOINT = 0
TRN_TENDON = 3  # mjTRN_TENDON

# MuJoCo defaults.
DEFAULT_SOLREF = (0.02, 1.0)
DEFAULT_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)
DEFAULT_FRICTION = (1.0, 0.005, 0.0001)


@dataclasses.dataclass
class Joint:
    name: str
    type: int = JNT_HINGE
    pos: Sequence[float] = (0.0, 0.0, 0.0)
    axis: Sequence[float] = (0.0, 0.0, 1.0)
    range: Optional[Tuple[float, float]] = None  # None => not limited.
    stiffness: float = 0.0
    springref: float = 0.0
    damping: float = 0.0
    armature: float = 0.0
    frictionloss: float = 0.0
    solref_limit: Sequence[float] = DEFAULT_SOLREF
    solimp_limit: Sequence[float] = DEFAULT_SOLIMP
    solref_friction: Sequence[float] = DEFAULT_SOLREF
    solimp_friction: Sequence[float] = DEFAULT_SOLIMP
    margin: float = 0.0


@dataclasses.dataclass
class Geom:
    name: str
    type: int
    size: Sequence[float]  # MuJoCo convention: half-sizes / (radius, half-length).
    pos: Sequence[float] = (0.0, 0.0, 0.0)
    quat: Sequence[float] = (1.0, 0.0, 0.0, 0.0)
    contype: int = 1
    conaffinity: int = 1
    condim: int = 3
    friction: Sequence[float] = DEFAULT_FRICTION
    solref: Sequence[float] = DEFAULT_SOLREF
    solimp: Sequence[float] = DEFAULT_SOLIMP
    solmix: float = 1.0
    margin: float = 0.0
    gap: float = 0.0
    priority: int = 0
    mass: Optional[float] = None  # If set, contributes to the body inertia.
    vertices: Optional[Sequence[Sequence[float]]] = None  # GEOM_MESH: hull vertices in the geom frame


@dataclasses.dataclass
class Site:
    name: str
    pos: Sequence[float] = (0.0, 0.0, 0.0)
    # > 0: the site is also the (spherical) zone of a touch sensor of this radius
    # (robopianist/models/hands/shadow_hand.py:248-270: r = 0.01 at every fingertip)
    touch_radius: float = 0.0


@dataclasses.dataclass
class Body:
    name: str
    pos: Sequence[float] = (0.0, 0.0, 0.0)
    quat: Sequence[float] = (1.0, 0.0, 0.0, 0.0)
    # Explicit inertial (mass, ipos, iquat, diaginertia); if None it is inferred
    # from the geoms that carry a `mass`.
    mass: Optional[float] = None
    ipos: Sequence[float] = (0.0, 0.0, 0.0)
    iquat: Sequence[float] = (1.0, 0.0, 0.0, 0.0)
    inertia: Sequence[float] = (0.0, 0.0, 0.0)
    gravcomp: float = 0.0
    joints: List[Joint] = dataclasses.field(default_factory=list)
    geoms: List[Geom] = dataclasses.field(default_factory=list)
    sites: List[Site] = dataclasses.field(default_factory=list)
    children: List["Body"] = dataclasses.field(default_factory=list)

    def add(self, child: "Body") -> "Body":
        self.children.append(child)
        return child

    def walk(self):
        yield self
        for c in self.children:
            yield from c.walk()

    def find(self, name: str) -> "Body":
        for b in self.walk():
            if b.name == name:
                return b
        raise KeyError(name)


@dataclasses.dataclass
class Tendon:
    """Fixed tendon: length = sum coef_i * qpos[joint_i]."""

    name: str
    joints: Sequence[str]
    coefs: Sequence[float]


@dataclasses.dataclass
class Actuator:
    """`general`-style actuator: force = gain*ctrl + b0 + b1*length + b2*velocity."""

    name: str
    joint: Optional[str] = None
    tendon: Optional[str] = None
    gain: float = 1.0
    bias: Sequence[float] = (0.0, 0.0, 0.0)
    ctrlrange: Optional[Tuple[float, float]] = None
    forcerange: Optional[Tuple[float, float]] = None
    gear: float = 1.0

    @staticmethod
    def position(name, kp, ctrlrange, forcerange=None, joint=None, tendon=None):
        """MuJoCo `<position>` shortcut: gain=kp, bias=(0, -kp, 0)."""
        return Actuator(
            name=name,
            joint=joint,
            tendon=tendon,
            gain=kp,
            bias=(0.0, -kp, 0.0),
            ctrlrange=ctrlrange,
            forcerange=forcerange,
        )


@dataclasses.dataclass
class Options:
    timestep: float = 0.002
    gravity: Sequence[float] = (0.0, 0.0, -9.81)
    tolerance: float = 1e-8
    iterations: int = 100
    ls_iterations: int = 50
    ls_tolerance: float = 0.01
    impratio: float = 1.0
    refsafe: bool = True


@dataclasses.dataclass
class Scene:
    world: Body
    tendons: List[Tendon] = dataclasses.field(default_factory=list)
    actuators: List[Actuator] = dataclasses.field(default_factory=list)
    excludes: List[Tuple[str, str]] = dataclasses.field(default_factory=list)
    options: Options = dataclasses.field(default_factory=Options)


# ---------------------------------------------------------------------------
# Small quaternion / rotation helpers (w, x, y, z convention, like MuJoCo).
# ---------------------------------------------------------------------------


def quat_normalize(q) -> np.ndarray:
    q = np.asarray(q, dtype=np.float64)
    return q / np.linalg.norm(q)


def quat_mul(a, b) -> np.ndarray:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.array(
        [
            a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
            a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
            a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
            a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
        ]
    )


def quat_to_mat(q) -> np.ndarray:
    w, x, y, z = quat_normalize(q)
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
        ]
    )


def axis_angle_to_quat(axis, angle) -> np.ndarray:
    axis = np.asarray(axis, dtype=np.float64)
    n = np.linalg.norm(axis)
    if n < 1e-14:
        return np.array([1.0, 0.0, 0.0, 0.0])
    axis = axis / n
    s = np.sin(angle / 2)
    return np.array([np.cos(angle / 2), *(axis * s)])
