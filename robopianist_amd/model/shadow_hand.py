"""STAND-IN Shadow Hand E3M5 model (26 DoF with the reference's forearm joints).

PROVENANCE — READ THIS.  The reference loads the hand from
`mujoco_menagerie@1afc8be6:shadow_hand/{right,left}_hand.xml`
(robopianist/models/hands/shadow_hand_constants.py:52-53,
scripts/install_deps.sh:81-88).  That submodule is EMPTY in /root/reference
and there is no network, so the XML and its meshes are unavailable.  What the
reference *does* pin — and what this file reproduces exactly — is:

  * topology and names: 24 joints / 20 actuators (shadow_hand_constants.py:21-22),
    WRJ2 first (shadow_hand_test.py:101-106), fingertip body order
    (shadow_hand_constants.py:33-40), actuator names A_THJ5/A_THJ1/A_LFJ5
    (shadow_hand.py:71-79), `plastic_collision` geoms, `*_distal_pst` fingertips;
  * everything the reference ADDS: forearm joints + position actuators
    (shadow_hand.py:41-69,272-311), fingertip sites at z=0.026/0.0275
    (shadow_hand.py:81-82,192-207), hand placement (suite/tasks/base.py:34-37).

Every other number below (link offsets, inertias, joint ranges/damping,
actuator gains, collision primitive sizes) is restated FROM MEMORY of the
menagerie XML and is UNVERIFIED.  Differences from the real asset that are
deliberate:
  * mesh colliders (forearm, lfmetacarpal, fingertips) are replaced by
    primitives; fingertips are capsules, i.e. the reference's
    `primitive_fingertip_collisions=True` mode (shadow_hand.py:144-152) with
    hand-picked capsule sizes instead of MuJoCo's mesh-fit ones;
  * cylinder colliders (wrist, knuckles) are capsules by default (the benchmark's kernel builds carry no
    cylinder code); `cylinder_colliders=True` restores them as cylinders of the same radius / half length, collided
    through the support-function path like hulls (round 6: both narrow phases handle mjGEOM_CYLINDER).
Round 6 also made two corrections to the stand-in itself: the scene carries the hand XML's `<option impratio="10"/>`
(model/scene.py), and the forearm's wrist box sits 12 mm lower (`standin_wrist_clearance`, now the
default): with the remembered z = 0.181 two RIGID links two joints apart overlapped by 7.6 mm at the end of WRJ2's
range and sat in contact on 43-61 % of the replay's mj_steps -- an artefact of this file, not of the hand.
The compiled blob format is generic: a real mjModel dumped to the same arrays
drops in without touching the engine.
"""

from __future__ import annotations

import enum
import math
from typing import Dict, List, Sequence, Tuple

import numpy as np

from robopianist_amd.model import spec

NQ = 24  # shadow_hand_constants.py:21
NU = 20  # shadow_hand_constants.py:22

FINGERTIP_BODIES = ("thdistal", "ffdistal", "mfdistal", "rfdistal", "lfdistal")

_FINGERTIP_OFFSET = 0.026  # shadow_hand.py:81
_THUMBTIP_OFFSET = 0.0275  # shadow_hand.py:82
_TOUCH_RADIUS = 0.01       # shadow_hand.py:261

# shadow_hand.py:41-69. (type, axis, stiffness, range, reflect)
FOREARM_DOFS: Dict[str, tuple] = {
    "forearm_tx": (spec.JNT_SLIDE, (-1, 0, 0), 300.0, (-1.0, 1.0), False),
    "forearm_ty": (spec.JNT_SLIDE, (0, 0, 1), 300.0, (0.0, 0.06), False),
    "forearm_tz": (spec.JNT_SLIDE, (0, 1, 0), 1000.0, (-0.04, 0.0), False),
    "forearm_roll": (spec.JNT_HINGE, (0, 0, 1), 300.0, (-0.25, 0.25), False),
    "forearm_pitch": (spec.JNT_HINGE, (1, 0, 0), 50.0, (0.0, 0.15), False),
    "forearm_yaw": (spec.JNT_HINGE, (0, -1, 0), 300.0, (-0.25, 0.25), True),
}
DEFAULT_FOREARM_DOFS = ("forearm_tx", "forearm_ty")  # shadow_hand.py:85

_REDUCED_ACTION_SPACE_EXCLUDED_DOFS = ("A_THJ5", "A_THJ1", "A_LFJ5")  # :71-75
_REDUCED_THUMB_RANGE = (0.0, 0.698132)  # :77
_RESTRICTED_WRJ2_RANGE = (-0.174533, 0.174533)  # :69

# [MEM] contact parameters of the menagerie `plastic` default class.
_PLASTIC = dict(solimp=(0.5, 0.99, 0.0001, 0.5, 2.0), solref=(0.005, 1.0))

# [MEM] joint classes: axis, range, damping, (kp, ctrlrange, forcerange).
_J_DEFAULT = dict(damping=0.05, armature=0.0002, frictionloss=0.01)
_CLASSES = {
    "wrist_y": dict(axis=(0, 1, 0), range=(-0.523599, 0.174533), damping=0.5,
                    kp=10.0, forcerange=(-10.0, 10.0)),
    "wrist_x": dict(axis=(1, 0, 0), range=(-0.698132, 0.488692), damping=0.5,
                    kp=8.0, forcerange=(-5.0, 5.0)),
    "thbase": dict(axis=(0, 0, -1), range=(-1.0472, 1.0472), kp=0.4,
                   forcerange=(-3.0, 3.0)),
    "thproximal": dict(axis=(1, 0, 0), range=(0.0, 1.22173), kp=1.0,
                       forcerange=(-2.0, 2.0)),
    "thhub": dict(axis=(1, 0, 0), range=(-0.20944, 0.20944), kp=0.5,
                  forcerange=(-1.0, 1.0)),
    "thmiddle": dict(axis=(0, -1, 0), range=(-0.698132, 0.698132), kp=1.5,
                     forcerange=(-1.0, 1.0)),
    "thdistal": dict(axis=(1, 0, 0), range=(-0.261799, 1.5708), kp=1.0,
                     forcerange=(-1.0, 1.0)),
    "metacarpal": dict(axis=(0.573576, 0, 0.819152), range=(0.0, 0.785398),
                       kp=1.0, forcerange=(-1.0, 1.0)),
    "knuckle": dict(axis=(0, -1, 0), range=(-0.349066, 0.349066), kp=1.0,
                    forcerange=(-1.0, 1.0)),
    "proximal": dict(axis=(1, 0, 0), range=(-0.261799, 1.5708), kp=1.0,
                     forcerange=(-1.0, 1.0)),
    "middle_distal": dict(axis=(1, 0, 0), range=(0.0, 1.5708), kp=0.5,
                          ctrlrange=(0.0, 3.1415), forcerange=(-1.0, 1.0)),
}


def _mirror_vec(v, left: bool):
    v = tuple(float(x) for x in v)
    return (-v[0], v[1], v[2]) if left else v


def _mirror_axis(a, left: bool):
    # Rotation axes are pseudo-vectors: reflect in x, then flip the sense.
    a = tuple(float(x) for x in a)
    return (a[0], -a[1], -a[2]) if left else a


def _mirror_quat(q, left: bool):
    q = tuple(float(x) for x in q)
    return (q[0], q[1], -q[2], -q[3]) if left else q


class HandSide(enum.Enum):
    """Which hand is modelled (robopianist/models/hands/base.py:26-30)."""

    LEFT = enum.auto()
    RIGHT = enum.auto()


class HandBuilder:
    """Builds one hand as a `spec.Body` subtree plus tendons and actuators."""

    def __init__(
        self,
        side: str,
        forearm_dofs: Sequence[str] = DEFAULT_FOREARM_DOFS,
        restrict_wrist_yaw_range: bool = False,
        reduced_action_space: bool = False,
        primitive_fingertip_collisions: bool = True,
        standin_wrist_clearance: bool = True,
        cylinder_colliders: bool = False,
    ):
        assert side in ("right", "left")
        # `standin_wrist_clearance` (default since round 6): with the from-memory numbers (z = 0.181) the forearm's wrist
        # box and the palm boxes -- two rigid links, two joints apart -- interpenetrate by up to 7.6 mm when WRJ2 alone
        # is driven to the negative end of its range, and that pair was in contact on 43 % / 61 % of the replay's
        # mj_steps (oracle/standin_report.py).  On: the box sits 12 mm lower and the single-joint sweep finds no
        # rigid-link overlap apart from neighbouring fingers abducted into each other (tests/test_standin_report.py).
        # False = rounds 1-5's geometry, kept for comparisons.
        self.standin_wrist_clearance = bool(standin_wrist_clearance)
        self.primitive_fingertips = bool(primitive_fingertip_collisions)
        self.cylinder_colliders = bool(cylinder_colliders)
        for d in forearm_dofs:
            if d not in FOREARM_DOFS:
                # Same error behaviour as shadow_hand.py:283-287.
                raise ValueError(
                    f"Invalid forearm DOF: {d}. Valid DOFs are: {FOREARM_DOFS}."
                )
        self.side = side
        self.left = side == "left"
        self.prefix = "lh_" if self.left else "rh_"
        self.model_name = self.prefix + "shadow_hand"
        self.forearm_dofs = tuple(forearm_dofs)
        self.restrict_wrist_yaw_range = restrict_wrist_yaw_range
        self.reduced_action_space = reduced_action_space
        self.tendons: List[spec.Tendon] = []
        self.actuators: List[spec.Actuator] = []
        # Python-side ordered views, mirroring ShadowHand.joints/.actuators.
        self.joint_names: List[str] = []
        self.actuator_names: List[str] = []
        self.fingertip_sites: List[str] = []
        self.root: spec.Body = self._build()

    # -- helpers ---------------------------------------------------------
    def _n(self, name: str) -> str:
        return f"{self.model_name}/{self.prefix}{name}"

    def _joint(self, name: str, cls: str) -> spec.Joint:
        c = _CLASSES[cls]
        kw = dict(_J_DEFAULT)
        if "damping" in c:
            kw["damping"] = c["damping"]
        rng = c["range"]
        if name == "WRJ2" and self.restrict_wrist_yaw_range:
            rng = _RESTRICTED_WRJ2_RANGE
        if name == "THJ2" and self.reduced_action_space:
            rng = _REDUCED_THUMB_RANGE
        return spec.Joint(
            name=self._n(name),
            type=spec.JNT_HINGE,
            axis=_mirror_axis(c["axis"], self.left),
            range=rng,
            **kw,
        )

    def _body(self, name, pos, mass, ipos, iquat, inertia, quat=(1, 0, 0, 0)):
        return spec.Body(
            name=self._n(name),
            pos=_mirror_vec(pos, self.left),
            quat=_mirror_quat(quat, self.left),
            mass=mass,
            ipos=_mirror_vec(ipos, self.left),
            iquat=_mirror_quat(iquat, self.left),
            inertia=inertia,
        )

    def _capsule(self, name, radius, half, pos=(0, 0, 0), quat=(1, 0, 0, 0)):
        return spec.Geom(
            self._n(name), spec.GEOM_CAPSULE, (radius, half, 0.0),
            pos=_mirror_vec(pos, self.left), quat=_mirror_quat(quat, self.left),
            **_PLASTIC,
        )

    def _cylinder(self, name, radius, half, pos=(0, 0, 0), quat=(1, 0, 0, 0)):
        """Wrist / knuckle colliders: cylinders in the menagerie XML [MEM]; capsules unless `cylinder_colliders`."""
        if not self.cylinder_colliders:
            return self._capsule(name, radius, half, pos, quat)
        return spec.Geom(
            self._n(name), spec.GEOM_CYLINDER, (radius, half, 0.0),
            pos=_mirror_vec(pos, self.left), quat=_mirror_quat(quat, self.left),
            **_PLASTIC,
        )

    def _fingertip(self, name, radius, half, pos):
        """Collision geom of a distal phalanx (`*distal_pst`).  The reference's default is the
        menagerie mesh, collided through its convex hull; `primitive_fingertip_collisions=True`
        turns it into the capsule MuJoCo fits to the mesh (shadow_hand.py:105-107,144-152).  The mesh
        is not available here: the stand-in hull is the 26-vertex polytope inscribed in that same
        capsule (poles, one ring on each cap at 45 degrees, the two rings of the cylinder; 6 azimuths),
        so the two modes describe the same fingertip with the two collision pipelines."""
        if self.primitive_fingertips:
            return self._capsule(name, radius, half, pos)
        verts = [(0.0, 0.0, half + radius), (0.0, 0.0, -half - radius)]
        c45 = math.sqrt(0.5)
        for k in range(6):
            a = 2 * math.pi * k / 6
            ca, sa = math.cos(a), math.sin(a)
            verts += [(radius * ca, radius * sa, half), (radius * ca, radius * sa, -half),
                      (radius * c45 * ca, radius * c45 * sa, half + radius * c45),
                      (radius * c45 * ca, radius * c45 * sa, -half - radius * c45)]
        if self.left:  # mirrored hand: mirror the hull like every other position
            verts = [tuple(_mirror_vec(v, True)) for v in verts]
        return spec.Geom(self._n(name), spec.GEOM_MESH, (radius, radius, half + radius),
                         pos=_mirror_vec(pos, self.left), vertices=verts, **_PLASTIC)

    def _box(self, name, size, pos=(0, 0, 0), quat=(1, 0, 0, 0)):
        return spec.Geom(
            self._n(name), spec.GEOM_BOX, size,
            pos=_mirror_vec(pos, self.left), quat=_mirror_quat(quat, self.left),
            **_PLASTIC,
        )

    # -- the tree (numbers: [MEM] menagerie E3M5, see module docstring) -----
    def _finger(self, palm, f, knuckle_pos):
        kn = palm.add(self._body(f + "knuckle", knuckle_pos, 0.008, (0, 0, 0),
                                 (0.5, 0.5, -0.5, 0.5), (3.2e-07, 2.6e-07, 2.6e-07)))
        kn.joints.append(self._joint(f.upper() + "J4", "knuckle"))
        kn.geoms.append(self._cylinder(f + "knuckle_col", 0.008, 0.001,
                                       quat=(1, 0, 1, 0)))
        self._finger_chain(kn, f)

    def _finger_chain(self, kn, f):
        pr = kn.add(self._body(f + "proximal", (0, 0, 0), 0.03, (0, 0, 0.0225),
                               (1, 0, 0, 1), (1e-05, 9.8e-06, 1.8e-06)))
        pr.joints.append(self._joint(f.upper() + "J3", "proximal"))
        pr.geoms.append(self._capsule(f + "proximal_col", 0.009, 0.02, (0, 0, 0.025)))
        mi = pr.add(self._body(f + "middle", (0, 0, 0.045), 0.017, (0, 0, 0.0125),
                               (1, 0, 0, 1), (2.7e-06, 2.6e-06, 8.7e-07)))
        mi.joints.append(self._joint(f.upper() + "J2", "middle_distal"))
        mi.geoms.append(self._capsule(f + "middle_col", 0.009, 0.0125, (0, 0, 0.0125)))
        di = mi.add(self._body(f + "distal", (0, 0, 0.025), 0.013, (0, 0, 0.0130769),
                               (1, 0, 0, 1), (1.28092e-06, 1.12092e-06, 5.3e-07)))
        di.joints.append(self._joint(f.upper() + "J1", "middle_distal"))
        # Stand-in for mesh `f_distal_pst` in primitive (capsule) mode.
        di.geoms.append(self._fingertip(f + "distal_pst", 0.0075, 0.0065, (0, 0, 0.012)))
        # fingertip site (shadow_hand.py:192-207) and, at the same place, the r = 0.01 zone of the
        # fingertip's touch sensor (:248-270)
        di.sites.append(spec.Site(self._n(f + "distal_site"), (0, 0, _FINGERTIP_OFFSET), touch_radius=_TOUCH_RADIUS))

    def _build(self) -> spec.Body:
        fa = self._body("forearm", (0, 0, 0), 3.0, (0, 0, 0.09), (1, 0, 0, 0),
                        (0.0138, 0.0138, 0.00744))
        # Stand-in for mesh `forearm_collision`: a capsule along the arm, plus the
        # menagerie's box near the wrist.
        fa.geoms.append(self._capsule("forearm_col", 0.04, 0.07, (0, 0, 0.08)))
        fa.geoms.append(self._box("forearm_box", (0.035, 0.035, 0.035),
                                  (0.01, 0, 0.169 if self.standin_wrist_clearance else 0.181), (0.380188, 0.924909, 0, 0)))
        wr = fa.add(self._body("wrist", (0.01, 0, 0.21301), 0.1, (0, 0, 0.029),
                               (0.5, 0.5, 0.5, 0.5), (6.4e-05, 4.38e-05, 3.5e-05)))
        wr.joints.append(self._joint("WRJ2", "wrist_y"))
        wr.geoms.append(self._cylinder("wrist_col", 0.0135, 0.015,
                                       quat=(0.5, 0.5, 0.5, -0.5)))
        palm = wr.add(self._body("palm", (0, 0, 0.034), 0.3, (0, 0, 0.035),
                                 (1, 0, 0, 1), (0.0005287, 0.0003581, 0.000191)))
        palm.joints.append(self._joint("WRJ1", "wrist_x"))
        palm_boxes = [
            ((0.031, 0.0035, 0.049), (0.011, 0.0085, 0.038), (1, 0, 0, 0)),
            ((0.018, 0.0085, 0.049), (-0.002, -0.0035, 0.038), (1, 0, 0, 0)),
            ((0.013, 0.0085, 0.005), (0.029, -0.0035, 0.082), (1, 0, 0, 0)),
            ((0.013, 0.007, 0.009), (0.0265, -0.001, 0.07),
             (0.987241, 0.0990545, 0.0124467, 0.124052)),
            ((0.0105, 0.0135, 0.012), (0.0315, -0.0085, 0.001), (1, 0, 0, 0)),
            ((0.011, 0.0025, 0.015), (0.0125, -0.015, 0.004),
             (0.971338, 0, 0, -0.237703)),
            ((0.009, 0.012, 0.002), (0.011, 0, 0.089), (1, 0, 0, 0)),
            ((0.01, 0.012, 0.02), (-0.03, 0, 0.009), (1, 0, 0, 0)),
        ]
        for i, (size, pos, quat) in enumerate(palm_boxes):
            palm.geoms.append(self._box(f"palm_col{i}", size, pos, quat))

        # Body order in the XML is ff, mf, rf, lf, th [MEM]; the actuator list
        # (below) has the thumb right after the wrist.
        self._finger(palm, "ff", (0.033, 0, 0.095))
        self._finger(palm, "mf", (0.011, 0, 0.099))
        self._finger(palm, "rf", (-0.011, 0, 0.095))
        # Little finger has an extra metacarpal joint.
        mc = palm.add(self._body("lfmetacarpal", (-0.033, 0, 0.02071), 0.03,
                                 (0, 0, 0.04),
                                 (1, 0, 0, 1), (1.638e-05, 1.45e-05, 4.272e-06)))
        mc.joints.append(self._joint("LFJ5", "metacarpal"))
        mc.geoms.append(self._box("lfmetacarpal_col", (0.011, 0.012, 0.025),
                                  (0.002, 0, 0.033)))
        kn = mc.add(self._body("lfknuckle", (0, 0, 0.06579), 0.008, (0, 0, 0),
                               (0.5, 0.5, -0.5, 0.5), (3.2e-07, 2.6e-07, 2.6e-07)))
        kn.joints.append(self._joint("LFJ4", "knuckle"))
        kn.geoms.append(self._cylinder("lfknuckle_col", 0.008, 0.001, quat=(1, 0, 1, 0)))
        self._finger_chain(kn, "lf")
        # Thumb.
        tb = palm.add(self._body("thbase", (0.034, -0.00858, 0.029), 0.01, (0, 0, 0),
                                 (1, 0, 0, 0), (1.6e-07, 1.6e-07, 1.6e-07),
                                 quat=(0.92388, 0, 0.382683, 0)))
        tb.joints.append(self._joint("THJ5", "thbase"))
        tp = tb.add(self._body("thproximal", (0, 0, 0), 0.04, (0, 0, 0.019),
                               (1, 0, 0, 0), (1.36e-05, 1.36e-05, 3.13e-06)))
        tp.joints.append(self._joint("THJ4", "thproximal"))
        tp.geoms.append(self._capsule("thproximal_col", 0.013, 0.019, (0, 0, 0.019)))
        th = tp.add(self._body("thhub", (0, 0, 0.038), 0.005, (0, 0, 0),
                               (1, 0, 0, 0), (1e-06, 1e-06, 3e-07)))
        th.joints.append(self._joint("THJ3", "thhub"))
        tm = th.add(self._body("thmiddle", (0, 0, 0), 0.02, (0, 0, 0.016),
                               (1, 0, 0, 0), (5.1e-06, 5.1e-06, 1.21e-06)))
        tm.joints.append(self._joint("THJ2", "thmiddle"))
        tm.geoms.append(self._capsule("thmiddle_col", 0.011, 0.016, (0, 0, 0.016)))
        td = tm.add(self._body("thdistal", (0, 0, 0.032), 0.017, (0, 0, 0.0145588),
                               (1, 0, 0, 1), (2.37794e-06, 2.27794e-06, 1e-06),
                               quat=(1, 0, 0, -1)))
        td.joints.append(self._joint("THJ1", "thdistal"))
        td.geoms.append(self._fingertip("thdistal_pst", 0.009, 0.005, (0, 0, 0.0135)))
        td.sites.append(spec.Site(self._n("thdistal_site"), (0, 0, _THUMBTIP_OFFSET), touch_radius=_TOUCH_RADIUS))

        # Joint order as PyMJCF's find_all("joint") returns it (document order;
        # forearm joints are appended to the root body AFTER its child bodies, so
        # they come last: shadow_hand.py:121-124,159-160 and the test at
        # shadow_hand_test.py:101-106 which expects joints[0] == WRJ2).
        xml_joint_order = [
            "WRJ2", "WRJ1",
            "FFJ4", "FFJ3", "FFJ2", "FFJ1",
            "MFJ4", "MFJ3", "MFJ2", "MFJ1",
            "RFJ4", "RFJ3", "RFJ2", "RFJ1",
            "LFJ5", "LFJ4", "LFJ3", "LFJ2", "LFJ1",
            "THJ5", "THJ4", "THJ3", "THJ2", "THJ1",
        ]
        # [MEM] menagerie actuator order.
        act_order = [
            ("A_WRJ2", "WRJ2", "wrist_y"), ("A_WRJ1", "WRJ1", "wrist_x"),
            ("A_THJ5", "THJ5", "thbase"), ("A_THJ4", "THJ4", "thproximal"),
            ("A_THJ3", "THJ3", "thhub"), ("A_THJ2", "THJ2", "thmiddle"),
            ("A_THJ1", "THJ1", "thdistal"),
            ("A_FFJ4", "FFJ4", "knuckle"), ("A_FFJ3", "FFJ3", "proximal"),
            ("A_FFJ0", None, "middle_distal"),
            ("A_MFJ4", "MFJ4", "knuckle"), ("A_MFJ3", "MFJ3", "proximal"),
            ("A_MFJ0", None, "middle_distal"),
            ("A_RFJ4", "RFJ4", "knuckle"), ("A_RFJ3", "RFJ3", "proximal"),
            ("A_RFJ0", None, "middle_distal"),
            ("A_LFJ5", "LFJ5", "metacarpal"), ("A_LFJ4", "LFJ4", "knuckle"),
            ("A_LFJ3", "LFJ3", "proximal"), ("A_LFJ0", None, "middle_distal"),
        ]
        removed_joints = set()
        if self.reduced_action_space:
            # shadow_hand.py:162-182: actuator AND joint are removed.
            for a in _REDUCED_ACTION_SPACE_EXCLUDED_DOFS:
                removed_joints.add(a[2:])
        for b in fa.walk():
            b.joints = [j for j in b.joints
                        if j.name.split(self.prefix)[-1] not in removed_joints]
        # 4 fixed tendons xFJ0 = xFJ2 + xFJ1 [MEM].
        for f in ("FF", "MF", "RF", "LF"):
            self.tendons.append(spec.Tendon(self._n(f + "J0"),
                                            (self._n(f + "J2"), self._n(f + "J1")),
                                            (1.0, 1.0)))
        for aname, jname, cls in act_order:
            if self.reduced_action_space and aname in _REDUCED_ACTION_SPACE_EXCLUDED_DOFS:
                continue
            c = _CLASSES[cls]
            ctrlrange = c.get("ctrlrange", c["range"])
            if aname == "A_WRJ2" and self.restrict_wrist_yaw_range:
                ctrlrange = _RESTRICTED_WRJ2_RANGE
            if aname == "A_THJ2" and self.reduced_action_space:
                ctrlrange = _REDUCED_THUMB_RANGE
            kw = dict(name=self._n(aname), kp=c["kp"], ctrlrange=ctrlrange,
                      forcerange=c["forcerange"])
            if jname is None:
                kw["tendon"] = self._n(aname[2:])
            else:
                kw["joint"] = self._n(jname)
            self.actuators.append(spec.Actuator.position(**kw))
        self.joint_names = [self._n(j) for j in xml_joint_order
                            if j not in removed_joints]

        # Forearm DoFs (shadow_hand.py:272-311). Damping is filled in by the
        # compiler (critical damping from dof_M0, mujoco_utils.physics_utils
        # .get_critical_damping_from_stiffness); we mark it with damping=-kp.
        for dof_name in self.forearm_dofs:
            jtype, axis, kp, rng, reflect = FOREARM_DOFS[dof_name]
            axis = tuple(float(a) for a in axis)
            if self.left and reflect:
                axis = tuple(-a for a in axis)
            jn = f"{self.model_name}/{dof_name}"
            fa.joints.append(spec.Joint(name=jn, type=jtype, axis=axis, range=rng,
                                        damping=-kp))
            self.actuators.append(spec.Actuator.position(
                name=jn, kp=kp, ctrlrange=rng, joint=jn))
            self.joint_names.append(jn)
        self.actuator_names = [a.name for a in self.actuators]
        self.fingertip_sites = [self._n(n + "_site") for n in FINGERTIP_BODIES]
        return fa

    # -- post-build edits used by the task (suite/tasks/base.py:149-197) -----
    def set_pose(self, pos, quat):
        self.root.pos = tuple(float(x) for x in pos)
        self.root.quat = tuple(float(x) for x in quat)

    def compensate_gravity(self):
        for b in self.root.walk():
            b.gravcomp = 1.0

    def set_forearm_tx_range(self, rng):
        jn = f"{self.model_name}/forearm_tx"
        for j in self.root.joints:
            if j.name == jn:
                j.range = tuple(rng)
        for a in self.actuators:
            if a.name == jn:
                a.ctrlrange = tuple(rng)

    def disable_hand_collisions(self):
        """piano_with_shadow_hands.py:476-489: contype=1, conaffinity=0."""
        for b in self.root.walk():
            for g in b.geoms:
                if g.contype == 0 and g.conaffinity == 0:
                    continue
                g.conaffinity = 0
                g.contype = 1
