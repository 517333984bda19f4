"""Scene assembly for the RoboPianist tasks.

Mirrors what the reference does with composer entities:
  PianoOnlyTask.__init__   robopianist/suite/tasks/base.py:45-70
  PianoTask._add_hand      robopianist/suite/tasks/base.py:149-197
  Stage (non-colliding floor) robopianist/models/arenas/stage.py:61-68
Body order = world, piano base, 88 keys, right hand, left hand, which is the
order `arena.attach` produces (piano first, then right, then left), hence
dof order keys[0:88], right[88:114], left[114:140].
"""

from __future__ import annotations

import dataclasses
import warnings
from typing import Dict, List, Optional, Sequence

import numpy as np

from robopianist_amd.model import compile as mcompile
from robopianist_amd.model import piano as piano_model
from robopianist_amd.model import shadow_hand, spec

# suite/tasks/base.py:28-39
PHYSICS_TIMESTEP = 0.005
CONTROL_TIMESTEP = 0.05
LEFT_HAND_POSITION = (0.4, -0.15, 0.13)
LEFT_HAND_QUATERNION = (-1, -1, 1, 1)
RIGHT_HAND_POSITION = (0.4, 0.15, 0.13)
RIGHT_HAND_QUATERNION = (-1, -1, 1, 1)


@dataclasses.dataclass
class HandInfo:
    side: str
    name: str
    joint_ids: np.ndarray  # Python-order joints (24 hand joints, then forearm)
    actuator_ids: np.ndarray  # 20 hand actuators then forearm actuators
    fingertip_site_ids: np.ndarray  # th, ff, mf, rf, lf
    root_body_id: int
    root_site_id: int  # site at the root body's origin (-1 unless build_scene(root_sites=True))
    forearm_geom_ids: np.ndarray
    n_forearm_dofs: int


@dataclasses.dataclass
class SceneInfo:
    model: mcompile.Model
    key_joint_ids: np.ndarray
    key_geom_ids: np.ndarray
    key_body_ids: np.ndarray
    key_actuator_ids: Optional[np.ndarray]
    hands: Dict[str, HandInfo]
    piano_size: tuple


def _inscribed_hull(g: "spec.Geom", n: int):
    """n points on the surface of the capsule / box-inscribed ellipsoid of geom g (its own frame): Fibonacci
    directions, each mapped to the primitive's support point (capsule) or scaled by the half sizes (box)."""
    i = np.arange(n) + 0.5
    z = 1.0 - 2.0 * i / n
    phi = np.pi * (1.0 + 5.0 ** 0.5) * i
    r = np.sqrt(1.0 - z * z)
    u = np.stack([r * np.cos(phi), r * np.sin(phi), z], 1)
    if g.type == spec.GEOM_CAPSULE:
        rad, half = float(g.size[0]), float(g.size[1])
        p = rad * u
        p[:, 2] += np.where(u[:, 2] >= 0, half, -half)
    else:
        p = u * np.asarray(list(g.size)[:3], float)[None, :]
    return [tuple(float(x) for x in row) for row in p]


def _meshify(root: "spec.Body", n: int) -> None:
    def walk(b):
        for k, g in enumerate(b.geoms):
            if g.type in (spec.GEOM_CAPSULE, spec.GEOM_BOX):
                b.geoms[k] = dataclasses.replace(g, type=spec.GEOM_MESH, vertices=_inscribed_hull(g, n))
        for c in b.children:
            walk(c)
    walk(root)


def build_scene(
    hands: Sequence[str] = ("right", "left"),
    add_piano_actuators: bool = False,
    gravity_compensation: bool = False,
    primitive_fingertip_collisions: bool = False,
    reduced_action_space: bool = False,
    attachment_yaw: float = 0.0,
    forearm_dofs: Sequence[str] = shadow_hand.DEFAULT_FOREARM_DOFS,
    physics_timestep: float = PHYSICS_TIMESTEP,
    disable_hand_collisions: bool = False,
    root_sites: bool = False,
    mesh_colliders: int = 0,
    standin_wrist_clearance: bool = True,
    cylinder_colliders: bool = False,
    impratio: Optional[float] = None,
) -> SceneInfo:
    """`mesh_colliders` = n > 0 (extension, for tests and the large-hull bench figure): every collider of the hands
    becomes a convex hull of ~n vertices inscribed in its stand-in primitive -- what the reference's default hand looks
    like to the collision pipeline, where forearm, wrist, palm, thumb links and fingertips all are `plastic_collision`
    meshes collided through their hulls (/root/reference/robopianist/models/hands/shadow_hand.py:144-152,
    shadow_hand_constants.py:52-53).
    `impratio`: `opt.impratio` of the compiled scene.  None = what the reference's scene ends up with: the hand XML's
    `<option impratio="10"/>` (`mjcf.from_path`, models/hands/shadow_hand.py:122, and `arena.attach` merge
    the hand's options into the root) when a hand is attached, MuJoCo's default 1 for the piano alone.
    `cylinder_colliders`: wrist / knuckle colliders as cylinders (the menagerie's types) instead of capsules.
    `standin_wrist_clearance=False`: rounds 1-5's forearm box (a rigid-link overlap; model/shadow_hand.py)."""
    if hands and not primitive_fingertip_collisions:
        warnings.warn(
            "The menagerie fingertip meshes are not available (mujoco_menagerie is not vendored in the "
            "reference checkout): the fingertips collide through a stand-in convex hull (the 26-vertex "
            "polytope inscribed in the capsule that primitive_fingertip_collisions=True uses).",
            stacklevel=2,
        )
    world = spec.Body(name="world")
    base, keys, piano_acts = piano_model.build(add_piano_actuators, physics_timestep)
    world.add(base)
    for k in keys:
        world.add(k)
    scene = spec.Scene(world=world)
    scene.options.timestep = physics_timestep
    scene.options.impratio = float(impratio) if impratio is not None else (10.0 if hands else 1.0)
    scene.actuators.extend(piano_acts)

    builders = {}
    for side in hands:
        hb = shadow_hand.HandBuilder(
            side=side, forearm_dofs=forearm_dofs,
            reduced_action_space=reduced_action_space,
            primitive_fingertip_collisions=primitive_fingertip_collisions,
            standin_wrist_clearance=standin_wrist_clearance,
            cylinder_colliders=cylinder_colliders,
        )
        position = RIGHT_HAND_POSITION if side == "right" else LEFT_HAND_POSITION
        quaternion = RIGHT_HAND_QUATERNION if side == "right" else LEFT_HAND_QUATERNION
        # base.py:176-183: yaw about world z, sign flipped for the left hand.
        sign = -1 if side == "left" else 1
        rotate_by = spec.axis_angle_to_quat((0, 0, 1), np.radians(sign * attachment_yaw))
        # mju_mulQuat does not normalise; normalisation happens at compile.
        final_quat = spec.quat_mul(rotate_by, np.asarray(quaternion, float))
        hb.set_pose(position, final_quat)
        if gravity_compensation:
            hb.compensate_gravity()
        # base.py:160-163,189-194: forearm_tx range spans the keyboard.
        joint_range = [-piano_model.BASE_SIZE[1] - position[1],
                       piano_model.BASE_SIZE[1] - position[1]]
        if "forearm_tx" in forearm_dofs:
            hb.set_forearm_tx_range(joint_range)
        if disable_hand_collisions:
            hb.disable_hand_collisions()
        if root_sites:
            # origin of the hand's root body, for HandObservables.position (hands/base.py:111-114)
            hb.root.sites.append(spec.Site(hb._n("forearm_origin_site"), (0.0, 0.0, 0.0)))
        if mesh_colliders:
            _meshify(hb.root, int(mesh_colliders))
        world.add(hb.root)
        scene.tendons.extend(hb.tendons)
        scene.actuators.extend(hb.actuators)
        # [MEM] menagerie <contact><exclude> pairs.
        scene.excludes.append((hb._n("wrist"), hb._n("forearm")))
        scene.excludes.append((hb._n("thproximal"), hb._n("thmiddle")))
        builders[side] = hb

    m = mcompile.compile_scene(scene)
    n = m.names
    key_joint_ids = np.array([n["joint"].index(k.joints[0].name) for k in keys], np.int32)
    key_geom_ids = np.array([n["geom"].index(k.geoms[0].name) for k in keys], np.int32)
    key_body_ids = np.array([n["body"].index(k.name) for k in keys], np.int32)
    key_act_ids = None
    if add_piano_actuators:
        key_act_ids = np.array([n["actuator"].index(a.name) for a in piano_acts], np.int32)
    infos = {}
    for side, hb in builders.items():
        root_id = n["body"].index(hb.root.name)
        infos[side] = HandInfo(
            side=side,
            name=hb.model_name,
            joint_ids=np.array([n["joint"].index(j) for j in hb.joint_names], np.int32),
            actuator_ids=np.array([n["actuator"].index(a) for a in hb.actuator_names], np.int32),
            fingertip_site_ids=np.array([n["site"].index(s) for s in hb.fingertip_sites], np.int32),
            root_body_id=root_id,
            root_site_id=(n["site"].index(hb._n("forearm_origin_site")) if root_sites else -1),
            forearm_geom_ids=np.array(
                [g for g in range(m.ngeom) if m.geom_bodyid[g] == root_id], np.int32),
            n_forearm_dofs=len(hb.forearm_dofs),
        )
    return SceneInfo(
        model=m, key_joint_ids=key_joint_ids, key_geom_ids=key_geom_ids,
        key_body_ids=key_body_ids, key_actuator_ids=key_act_ids, hands=infos,
        piano_size=piano_model.BASE_SIZE,
    )
