"""Pins the CPU oracle (oracle/rp_oracle.c) with what IS available.

PARITY UNPINNED w.r.t. MuJoCo: the reference holds no numeric physics vectors and
MuJoCo is absent.  The anchors here are the reference's one
physical inequality (piano_with_shadow_hands_test.py:228-242) and analytic known
answers (SURVEY.md §8c "analytic known-answers")."""
import warnings

import numpy as np
import pytest

from oracle.rp_oracle import Oracle
from robopianist_amd.model import compile as mc
from robopianist_amd.model import scene


def _oracle(si, **overrides):
    m = si.model
    for k, v in overrides.items():
        m[k] = v
    return Oracle(m, mc.to_blob(m))


def _fresh_two_hands(**kw):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return scene.build_scene(primitive_fingertip_collisions=True, **kw)


def test_mass_matrix_matches_jacobian_sum(two_hand_scene):
    o = _oracle(two_hand_scene)
    rng = np.random.default_rng(0)
    m = two_hand_scene.model
    o.qpos[88:] = rng.uniform(-0.3, 0.3, 52)
    o.forward()
    M = mc.mass_matrix(m, mc.kinematics(m, o.qpos.copy()))
    np.testing.assert_allclose(o.qM.reshape(m.nv, m.nv), M, atol=1e-13)


def test_keys_rest_on_their_lower_limit(piano_only_scene):
    """Spring preload 2*0.01745 = 0.0349 N.m beats gravity 0.0294 (white) / 0.0088
    (black): the keys sit slightly below q=0 on the soft limit."""
    o = _oracle(piano_only_scene)
    o.step(400)
    q = o.qpos.copy()
    assert np.all(q < 0) and np.all(q > -2e-3)
    assert np.abs(o.qvel).max() < 1e-6
    assert o.ncon == 0 and o.nefc == 88


def test_free_key_is_a_damped_oscillator(piano_only_scene):
    """Isolated key away from its limits = 1-dof spring/damper/gravity; compare with a
    fine-step integration of the scalar ODE."""
    si = scene.build_scene(hands=(), add_piano_actuators=True)
    m = si.model
    o = Oracle(m, mc.to_blob(m))
    k = 40
    q0 = 0.03
    o.qpos[k] = q0
    o.forward()
    I, kk, b = m.dof_M0[k], m.jnt_stiffness[k], m.dof_damping[k]
    qref = m.qpos_spring[k]
    mass = m.body_mass[si.key_body_ids[k]]
    hx = m.geom_size[si.key_geom_ids[k], 0]
    h = m.opt_timestep
    # same implicit-damping Euler written out for the scalar system
    q, v = q0, 0.0
    for _ in range(20):
        f = -kk * (q - qref) - b * v + mass * 9.81 * hx * np.cos(q)
        v += h * f / (I + h * b)
        q += h * v
        o.step()
        if not (0 < q < m.jnt_range[k, 1]):
            break
        assert o.qpos[k] == pytest.approx(q, abs=1e-12)


def test_energy_drift_is_first_order_in_dt():
    drift = []
    for dt in (2e-4, 1e-4):
        si = _fresh_two_hands()
        m = si.model
        m["dof_damping"][:] = 0; m["dof_frictionloss"][:] = 0; m["jnt_limited"][:] = 0
        m["jnt_stiffness"][:] = 0; m["npair"] = 0; m["pair_geom"] = np.zeros((0, 2), np.int32)
        m["actuator_gainprm"][:] = 0; m["actuator_biasprm"][:] = 0; m["opt_timestep"] = dt
        o = Oracle(m, mc.to_blob(m))
        rng = np.random.default_rng(0)
        o.qpos[88:] = rng.uniform(-0.3, 0.3, 52)
        o.qvel[88:] = rng.uniform(-3, 3, 52)
        o.qvel[[88, 89, 114, 115]] *= 0.1
        o.forward()

        def energy():
            kin = mc.kinematics(m, o.qpos.copy())
            M = o.qM.reshape(m.nv, m.nv)
            return 0.5 * o.qvel @ M @ o.qvel + np.sum(m.body_mass * 9.81 * kin["xipos"][:, 2])

        e0 = energy()
        o.step(int(round(0.05 / dt)))
        drift.append(abs(energy() - e0))
    assert drift[1] < 2e-3
    assert drift[0] / drift[1] == pytest.approx(2.0, rel=0.05)


def test_gravity_compensation_cancels_gravity():
    si = _fresh_two_hands(gravity_compensation=True)
    o = _oracle(si)
    o.forward()
    # hands hang still: passive gravcomp force == gravity bias on every hand dof
    np.testing.assert_allclose(o.qfrc_passive[88:] - o.qfrc_bias[88:], 0, atol=1e-9)


def test_solver_kkt_residual(two_hand_scene):
    """At the solution: M qacc - qfrc_smooth - J^T f = 0 with f consistent with the
    active set (limits/contacts push only)."""
    o = _oracle(two_hand_scene)
    m = two_hand_scene.model
    rng = np.random.default_rng(3)
    lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
    o.ctrl[:] = lo + rng.uniform(0.2, 0.8, m.nu) * (hi - lo)
    worst = 0.0
    for _ in range(150):
        o.step()
        o.forward()
        M = o.qM.reshape(m.nv, m.nv)
        J = o.efc_J.reshape(-1, m.nv)
        f = o.efc_force.copy()
        res = M @ o.qacc - o.qfrc_smooth - J.T @ f
        scale = max(1.0, np.abs(o.qfrc_smooth).max())
        worst = max(worst, np.abs(res).max() / scale)
    assert worst < 1e-6
    assert o.ncon >= 0


def test_three_newton_metres_press_every_key(two_hand_scene):
    """piano_with_shadow_hands_test.py:228-242: qfrc_applied = 3 on all key joints =>
    within one 0.01 s control step (2 substeps) a non-goal key is inside the 0.5 deg
    activation band."""
    o = _oracle(two_hand_scene)
    m = two_hand_scene.model
    o.qfrc_applied[:88] = 3.0
    o.step(2)
    qmax = m.jnt_range[:88, 1]
    state = np.clip(o.qpos[:88], 0, qmax)
    assert (np.abs(state - qmax) <= 0.00872665).all()


def test_capsule_box_contact_geometry(two_hand_scene):
    """A fingertip lowered onto a white key: one contact, normal along -z (capsule ->
    box), distance = gap between capsule surface and key top."""
    si = two_hand_scene
    m = si.model
    o = _oracle(si)
    names = m.names["joint"]
    j3 = names.index("rh_shadow_hand/rh_FFJ3")
    o.qpos[j3] = 1.2
    o.forward()
    c = o.contact.reshape(-1, 16)
    for row in c:
        g1, g2 = int(row[13]), int(row[14])
        assert m.geom_type[g1] <= m.geom_type[g2]
        n = row[4:7]
        assert np.linalg.norm(n) == pytest.approx(1.0)
        assert row[0] < 0


def test_gravity_bias_is_the_gradient_of_the_potential_energy(two_hand_scene):
    """RNE at zero velocity: qfrc_bias = dU/dq with U = -sum_b m_b g . com_b (central finite
    differences through the kinematics of model/compile.py), on a random hand pose."""
    m = two_hand_scene.model
    o = _oracle(two_hand_scene)
    rng = np.random.default_rng(5)
    q = np.zeros(m.nv)
    q[88:] = rng.uniform(-0.4, 0.4, m.nv - 88)
    q = np.clip(q, m.jnt_range[:, 0], m.jnt_range[:, 1])

    def potential(qq):
        kin = mc.kinematics(m, qq)
        u = 0.0
        for b in range(1, m.nbody):
            com = kin["xpos"][b] + kin["xmat"][b] @ m.body_ipos[b]
            u -= m.body_mass[b] * float(np.dot(m.opt_gravity, com))
        return u

    o.qpos[:] = q
    o.qvel[:] = 0
    o.forward()
    bias = o.qfrc_bias.copy()
    eps = 1e-6
    for j in list(range(88, m.nv, 5)) + [3, 40]:
        dq = np.zeros(m.nv); dq[j] = eps
        fd = (potential(q + dq) - potential(q - dq)) / (2 * eps)
        assert abs(bias[j] - fd) < 1e-7 * max(1.0, abs(fd)), (j, bias[j], fd)


def test_unconstrained_acceleration_solves_m_qacc_equals_qfrc_smooth(two_hand_scene):
    """qacc_smooth = M^-1 (passive + actuator + applied - bias), checked as a residual with the
    dense mass matrix on a moving random pose."""
    m = two_hand_scene.model
    o = _oracle(two_hand_scene)
    rng = np.random.default_rng(6)
    o.qpos[88:] = rng.uniform(-0.3, 0.3, m.nv - 88)
    o.qvel[:] = rng.uniform(-2, 2, m.nv)
    lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
    o.ctrl[:] = lo + rng.uniform(0.1, 0.9, m.nu) * (hi - lo)
    o.forward()
    M = o.qM.reshape(m.nv, m.nv)
    np.testing.assert_allclose(o.qfrc_smooth, o.qfrc_passive + o.qfrc_actuator + o.qfrc_applied - o.qfrc_bias,
                               atol=1e-12)
    res = M @ o.qacc_smooth - o.qfrc_smooth
    assert np.abs(res).max() < 1e-9 * max(1.0, np.abs(o.qfrc_smooth).max())
    assert np.allclose(M, M.T) and np.linalg.eigvalsh(M).min() > 0


# ---- acceleration-stage sensors (mj_rnePostConstraint / mj_sensorAcc restated) -------------------
def _contact_rich_state(si, steps=80, seed=3):
    m = si.model
    o = _oracle(si)
    rng = np.random.default_rng(seed)
    lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
    for s in range(steps):
        if s % 20 == 0:
            o.ctrl[:] = lo + rng.uniform(0.1, 0.9, m.nu) * (hi - lo)
        o.step(1)
    # after a step the acceleration-stage quantities (qacc, efc_force, sensors) belong to the state
    # BEFORE the Euler update while the position-stage ones were recomputed for the new state;
    # forward() makes everything refer to the current state
    o.forward()
    return o


def test_torque_sensor_equals_the_joint_space_force_balance(two_hand_scene):
    """Known answer tying cfrc_int (body-level Newton-Euler with the constrained qacc and the contact
    forces) to the joint-space dynamics: for a hinge whose anchor is its body's origin, the sensed
    torque about the joint axis is everything the joint itself transmits,
        passive + actuator + applied + (friction-loss and limit rows of that dof) - armature * qacc,
    because M qacc + bias = those + J_contact^T f and M = CRB + armature."""
    m = two_hand_scene.model
    o = _contact_rich_state(two_hand_scene)
    assert o.ncon >= 3
    nv, nefc = m.nv, o.nefc
    J = o.efc_J.reshape(nefc, nv).copy()
    f = o.efc_force.copy()
    ncon_rows = 4 * o.ncon
    joint_rows = J[:nefc - ncon_rows].T @ f[:nefc - ncon_rows]      # friction loss + limits (unit rows)
    expect = o.qfrc_passive + o.qfrc_actuator + o.qfrc_applied + joint_rows - m.dof_armature * o.qacc
    tau = o.sensor_torque.copy()
    sel = [j for j in range(88, nv) if m.jnt_type[j] == 3 and np.allclose(m.jnt_pos[j], 0)]
    assert len(sel) >= 40
    np.testing.assert_allclose(tau[sel], expect[sel], rtol=0, atol=1e-9 * max(1.0, np.abs(expect[sel]).max()))
    assert np.abs(tau[sel]).max() > 1e-3     # (a non-trivial state)
    # and the full balance closes: M qacc + bias - (all forces) = 0
    M = o.qM.reshape(nv, nv)
    res = M @ o.qacc + o.qfrc_bias - o.qfrc_passive - o.qfrc_actuator - o.qfrc_applied - J.T @ f
    assert np.abs(res).max() < 1e-7


def test_touch_sensor_sums_the_normal_forces_inside_the_fingertip_zone(two_hand_scene):
    """Independent numpy restatement from the oracle's contact list and row forces."""
    m = two_hand_scene.model
    hits = 0
    for seed in (3, 4, 5):
        o = _contact_rich_state(two_hand_scene, steps=120, seed=seed)
        con = o.contact.reshape(-1, 16).copy()
        f = o.efc_force.copy()[o.nefc - 4 * o.ncon:].reshape(-1, 4)
        want = np.zeros(m.nsite)
        sx = o.site_xpos.reshape(-1, 3)
        for c, fr in zip(con, f):
            fn = fr.sum()
            if fn <= 0:
                continue
            pos, n = c[1:4], c[4:7]
            b1, b2 = m.geom_bodyid[int(c[13])], m.geom_bodyid[int(c[14])]
            for s in np.flatnonzero(m.site_touch_radius > 0):
                sb = m.site_bodyid[s]
                if sb not in (b1, b2):
                    continue
                ray = -n if sb == b2 else n
                oc = pos - sx[s]
                bq, cq = oc @ ray, oc @ oc - m.site_touch_radius[s] ** 2
                det = bq * bq - cq
                if det >= 1e-15 and -bq + np.sqrt(det) >= 0:
                    want[s] += fn
        np.testing.assert_allclose(o.sensor_touch, want, rtol=1e-12, atol=1e-12)
        hits += int((want > 0).sum())
    assert hits >= 1, "no fingertip touched anything in these rollouts"


# ---- box-box narrow phase (known answers on a two-box scene) ---------------------------------------
def _two_box_scene(top_quat=(1, 0, 0, 0), hinges=()):
    """A 4 x 4 x 2 cm box on a vertical slide (plus optional hinges) above a static 10 x 10 x 2 cm box."""
    from robopianist_amd.model import spec
    world = spec.Body(name="world")
    world.geoms.append(spec.Geom("floor_box", spec.GEOM_BOX, (0.05, 0.05, 0.01), pos=(0, 0, 0.01)))
    joints = [spec.Joint("z", type=spec.JNT_SLIDE, axis=(0, 0, 1), damping=0.5)]
    for i, ax in enumerate(hinges):
        joints.append(spec.Joint(f"h{i}", type=spec.JNT_HINGE, axis=ax, damping=0.01))
    top = spec.Body(name="top", pos=(0, 0, 0.05), quat=top_quat, joints=joints,
                    geoms=[spec.Geom("top_box", spec.GEOM_BOX, (0.02, 0.02, 0.01), mass=0.1)])
    world.add(top)
    sc = spec.Scene(world=world)
    m = mc.compile_scene(sc)
    return m, Oracle(m, mc.to_blob(m))


def test_box_on_box_face_contact_settles():
    m, o = _two_box_scene()
    assert m.npair == 1
    o.step(600)
    con = o.contact.reshape(-1, 16)
    assert o.ncon == 4                                    # the four corners of the small face (rounds 1-3: the deepest three)
    np.testing.assert_allclose(con[:, 4:7], np.tile([0, 0, 1.0], (4, 1)), atol=1e-12)   # floor (geom1) -> top
    # rest: top box centre at 0.02 + 0.01 - penetration, a soft-contact penetration of well under 1 mm
    z = 0.05 + o.qpos[0]
    assert 0.0290 < z < 0.0300 and abs(o.qvel[0]) < 1e-6
    np.testing.assert_allclose(con[:, 0], z - 0.03, atol=1e-12)      # dist = -penetration
    assert set(map(tuple, np.round(np.abs(con[:, 1:3]), 12))) == {(0.02, 0.02)}      # corners of the small face
    assert len(set(map(tuple, np.round(con[:, 1:3], 12)))) == 4                       # ... all four of them
    np.testing.assert_allclose(con[:, 3], 0.02 + 0.5 * (z - 0.03), atol=1e-12)       # midway between the surfaces
    # static balance: the contact normal forces carry the weight
    f = o.efc_force[o.nefc - 4 * o.ncon:].sum()
    assert f == pytest.approx(0.1 * 9.81, rel=1e-3)


def test_box_box_rotated_face_and_edge_edge_cases():
    # (1) top box turned 45 degrees about z: its four corners still lie inside the floor face
    c, s_ = np.cos(np.pi / 8), np.sin(np.pi / 8)
    m, o = _two_box_scene(top_quat=(c, 0, 0, s_))
    o.qpos[0] = -0.0205; o.forward()
    con = o.contact.reshape(-1, 16)
    assert o.ncon == 4 and np.allclose(con[:, 0], -0.0005)
    r = np.hypot(con[:, 1], con[:, 2])
    np.testing.assert_allclose(r, 0.02 * np.sqrt(2), atol=1e-12)
    # (1b) a 6 x 6 cm face turned 45 degrees over the 10 x 10 cm floor face... the other way round: the small face is
    # the floor's.  A 4 x 4 cm square turned 45 degrees inside a 10 x 10 cm one stays a square; clipped by a 5 x 5 cm
    # one it becomes an octagon: eight contacts [mjc_BoxBox's maximum], all on the rim of the clipping rectangle or of
    # the turned square
    from robopianist_amd.model import spec
    world = spec.Body(name="world")
    world.geoms.append(spec.Geom("floor_box", spec.GEOM_BOX, (0.025, 0.025, 0.01), pos=(0, 0, 0.01)))
    top = spec.Body(name="top", pos=(0, 0, 0.05), quat=(c, 0, 0, s_),
                    joints=[spec.Joint("z", type=spec.JNT_SLIDE, axis=(0, 0, 1), damping=0.5)],
                    geoms=[spec.Geom("top_box", spec.GEOM_BOX, (0.02, 0.02, 0.01), mass=0.1)])
    world.add(top)
    m8 = mc.compile_scene(spec.Scene(world=world))
    o8 = Oracle(m8, mc.to_blob(m8))
    o8.qpos[0] = -0.0205; o8.forward()
    con = o8.contact.reshape(-1, 16)
    assert o8.ncon == 8 and np.allclose(con[:, 0], -0.0005)
    np.testing.assert_allclose(con[:, 4:7], np.tile([0, 0, 1.0], (8, 1)), atol=1e-12)
    on_rect = np.isclose(np.abs(con[:, 1:3]).max(axis=1), 0.025, atol=1e-12)
    on_diamond = np.isclose(np.abs(con[:, 1]) + np.abs(con[:, 2]), 0.02 * np.sqrt(2), atol=1e-12)
    assert np.all(on_rect & on_diamond) and len(set(map(tuple, np.round(con[:, 1:3], 9)))) == 8
    # (2) a big top face over a small floor: the reference corners / crossings come from the other box
    m, o = _two_box_scene(hinges=((1, 0, 0),))
    o.qpos[0] = -0.0202; o.qpos[1] = 0.3; o.forward()       # tilted about x: one edge of the top box digs in
    con = o.contact.reshape(-1, 16)
    assert 1 <= o.ncon <= 8 and np.all(con[:, 0] <= 0)
    np.testing.assert_allclose(con[:, 4:7], np.tile([0, 0, 1.0], (o.ncon, 1)), atol=1e-12)
    assert np.all(con[:, 2] < 0)                              # the lowered edge is on the -y side
    # (3) edge against edge: top box rolled 45 deg about x and yawed 90 deg - its lowest edge (along x
    # after the yaw... along y) crosses the floor's top edge region -> single contact, normal = +-(e1 x e2)
    from robopianist_amd.model import spec
    world = spec.Body(name="world")
    world.geoms.append(spec.Geom("floor_box", spec.GEOM_BOX, (0.05, 0.05, 0.01), pos=(0, 0, 0.01),
                                 quat=tuple(spec.axis_angle_to_quat(np.array([0.0, 1.0, 0.0]), np.pi / 4))))
    xr = -0.04 / np.sqrt(2)      # world x of the floor box's top ridge (it runs along y)
    top = spec.Body(name="top", pos=(xr, 0, 0.05), mass=0.1, inertia=(1e-4, 1e-4, 1e-4),
                    joints=[spec.Joint("z", type=spec.JNT_SLIDE, axis=(0, 0, 1))],
                    geoms=[spec.Geom("top_box", spec.GEOM_BOX, (0.02, 0.02, 0.01),
                                     quat=tuple(spec.axis_angle_to_quat(np.array([1.0, 0.0, 0.0]), np.pi / 4)))])
    world.add(top)
    m = mc.compile_scene(spec.Scene(world=world))
    o = Oracle(m, mc.to_blob(m))
    # floor box rotated about y: its top ridge runs along y at height 0.01 + (0.05 + 0.01)/sqrt(2); the top
    # box's lowest ridge runs along x at 0.05 + q - (0.02 + 0.01)/sqrt(2)
    ridge = 0.01 + 0.06 / np.sqrt(2)
    low0 = 0.05 - 0.03 / np.sqrt(2)
    o.qpos[0] = ridge - low0 - 0.001; o.forward()            # 1 mm of penetration, ridge on ridge
    con = o.contact.reshape(-1, 16)
    assert o.ncon == 1
    np.testing.assert_allclose(con[0, 0], -0.001, atol=1e-12)
    np.testing.assert_allclose(con[0, 4:7], [0, 0, 1.0], atol=1e-12)
    # the top box's lowest ridge runs along x at world y = -0.01 / sqrt(2)
    np.testing.assert_allclose(con[0, 1:4], [xr, -0.01 / np.sqrt(2), ridge - 0.0005], atol=1e-12)


# ---- convex hulls (GEOM_MESH) through MPR -------------------------------------------------------------
def _hull_scene(verts, floor="box"):
    from robopianist_amd.model import spec
    world = spec.Body(name="world")
    if floor == "box":
        world.geoms.append(spec.Geom("floor", spec.GEOM_BOX, (0.05, 0.05, 0.01), pos=(0, 0, 0.01)))
    else:
        world.geoms.append(spec.Geom("floor", spec.GEOM_CAPSULE, (0.01, 0.05, 0.0), pos=(0, 0, 0.01),
                                     quat=tuple(spec.axis_angle_to_quat(np.array([0.0, 1.0, 0.0]), np.pi / 2))))
    top = spec.Body(name="top", pos=(0, 0, 0.05), mass=0.1, inertia=(1e-4, 1e-4, 1e-4),
                    joints=[spec.Joint("z", type=spec.JNT_SLIDE, axis=(0, 0, 1), damping=0.5),
                            spec.Joint("x", type=spec.JNT_SLIDE, axis=(1, 0, 0), damping=0.5)],
                    geoms=[spec.Geom("hull", spec.GEOM_MESH, (0, 0, 0), vertices=verts)])
    world.add(top)
    m = mc.compile_scene(spec.Scene(world=world))
    return m, Oracle(m, mc.to_blob(m))


def _octahedron(r):
    return [(r, 0, 0), (-r, 0, 0), (0, r, 0), (0, -r, 0), (0, 0, r), (0, 0, -r)]


def test_hull_on_box_known_answers():
    # (1) a cube-shaped hull resting on the box: one contact, straight up, carrying the weight
    h = 0.02
    cube = [(sx * h, sy * h, sz * 0.01) for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]
    m, o = _hull_scene(cube)
    assert m.npair == 1 and m.geom_rbound[1] == pytest.approx(np.sqrt(2 * h * h + 1e-4))
    o.step(600)
    con = o.contact.reshape(-1, 16)
    assert o.ncon == 1
    np.testing.assert_allclose(con[0, 4:7], [0, 0, 1.0], atol=1e-9)          # box (geom1) -> hull
    z = 0.05 + o.qpos[0]
    assert 0.0290 < z < 0.0300
    np.testing.assert_allclose(con[0, 0], z - 0.03, atol=1e-9)
    f = o.efc_force[o.nefc - 4:].sum()
    assert f == pytest.approx(0.1 * 9.81, rel=1e-3)
    # (2) an octahedron pressed 1 mm into the face, vertex first: depth, normal and position are exact
    m, o = _hull_scene(_octahedron(0.015))
    o.qpos[0] = 0.02 + 0.015 - 0.001 - 0.05; o.qpos[1] = 0.004; o.forward()
    con = o.contact.reshape(-1, 16)
    assert o.ncon == 1
    np.testing.assert_allclose(con[0, 0], -0.001, atol=1e-6)                 # (the MPR tolerance)
    np.testing.assert_allclose(con[0, 4:7], [0, 0, 1.0], atol=1e-6)
    np.testing.assert_allclose(con[0, 1:4], [0.004, 0, 0.02 - 0.0005], atol=2e-4)   # ~midway, under the vertex (witness blend)
    # (3) clear of the box by 0.1 mm: no contact
    o.qpos[0] += 0.0011; o.forward()
    assert o.ncon == 0


def test_hull_against_capsule_matches_the_sphere_case():
    """Vertex-down octahedron on a lying capsule (r = 1 cm): the contact is the vertex against the
    cylinder, as for a point: depth = r - height of the vertex above the axis."""
    m, o = _hull_scene(_octahedron(0.015), floor="capsule")
    assert m.geom_type.tolist() == [3, 7] and m.npair == 1
    o.qpos[0] = (0.01 + 0.01 + 0.015 - 0.0007) - 0.05; o.forward()      # vertex 0.7 mm inside the cylinder
    con = o.contact.reshape(-1, 16)
    assert o.ncon == 1
    np.testing.assert_allclose(con[0, 0], -0.0007, atol=2e-6)
    np.testing.assert_allclose(con[0, 4:7], [0, 0, 1.0], atol=2e-2)   # (curved surface: the portal normal is good to sqrt(tol / r))


def _cylinder_scene(quat=(1, 0, 0, 0), impratio=1.0):
    """A cylinder (r = 2 cm, half height 1 cm; mjGEOM_CYLINDER) on two slides above the static 10 x 10 x 2 cm box."""
    from robopianist_amd.model import spec
    world = spec.Body(name="world")
    world.geoms.append(spec.Geom("floor", spec.GEOM_BOX, (0.05, 0.05, 0.01), pos=(0, 0, 0.01)))
    top = spec.Body(name="top", pos=(0, 0, 0.05), mass=0.1, inertia=(1e-4, 1e-4, 1e-4),
                    joints=[spec.Joint("z", type=spec.JNT_SLIDE, axis=(0, 0, 1), damping=0.5),
                            spec.Joint("x", type=spec.JNT_SLIDE, axis=(1, 0, 0), damping=0.5)],
                    geoms=[spec.Geom("cyl", spec.GEOM_CYLINDER, (0.02, 0.01, 0.0), quat=quat)])
    world.add(top)
    sc = spec.Scene(world=world)
    sc.options.impratio = impratio
    m = mc.compile_scene(sc)
    return m, Oracle(m, mc.to_blob(m))


def test_cylinder_on_box_known_answers():
    """Round 6: cylinders collide through the support-function path [MJ: mjc_Convex; mjc_support, mjGEOM_CYLINDER] -- the
    reference's hand keeps its wrist / knuckle colliders as cylinders (models/hands/shadow_hand.py:144-152 retypes only
    the fingertip meshes).  Closed-form depth and normal for a cylinder flat on a box face and for a tilted one."""
    from robopianist_amd.model import spec
    r, h, top = 0.02, 0.01, 0.02
    m, o = _cylinder_scene()
    assert m.geom_type.tolist() == [6, 5] and m.npair == 1 and m.geom_rbound[1] == pytest.approx(np.hypot(r, h))
    assert m.pair_geom.tolist() == [[1, 0]]                 # geom-type order: (cylinder 5, box 6)
    # (1) flat, 0.8 mm into the face, centred: the centre ray IS the axis -- exact
    o.qpos[0] = (top + h - 0.0008) - 0.05; o.forward()
    con = o.contact.reshape(-1, 16)
    assert o.ncon == 1
    np.testing.assert_allclose(con[0, 0], -0.0008, atol=1e-12)
    np.testing.assert_allclose(con[0, 4:7], [0, 0, -1.0], atol=1e-12)        # cylinder (geom1) -> box: straight down
    assert con[0, 3] == pytest.approx(top - 0.0004, abs=1e-12)               # midway between the two faces
    assert np.hypot(con[0, 1], con[0, 2]) <= r                                # (somewhere under the cap: the witness blend)
    # (2) flat and off-centre by 7 mm: same depth and normal (the flat cap against the flat face), to the MPR tolerance
    o.qpos[1] = 0.007; o.forward()
    con = o.contact.reshape(-1, 16)
    assert o.ncon == 1
    np.testing.assert_allclose(con[0, 0], -0.0008, atol=2e-6)
    np.testing.assert_allclose(con[0, 4:7], [0, 0, -1.0], atol=2e-3)
    # (3) resting: carries its weight
    o.reset(); o.step(600)
    assert o.ncon == 1 and abs(o.qvel[0]) < 1e-6
    assert o.efc_force[o.nefc - 4:].sum() == pytest.approx(0.1 * 9.81, rel=1e-3)
    # (4) tilted by 30 degrees about x: the lowest point of the rim is r sin(t) + h cos(t) below the centre
    th = np.radians(30.0)
    m, o = _cylinder_scene(quat=tuple(spec.axis_angle_to_quat(np.array([1.0, 0.0, 0.0]), th)))
    low = r * np.sin(th) + h * np.cos(th)
    o.qpos[0] = (top + low - 0.0005) - 0.05; o.forward()
    con = o.contact.reshape(-1, 16)
    assert o.ncon == 1
    np.testing.assert_allclose(con[0, 0], -0.0005, atol=2e-6)
    np.testing.assert_allclose(con[0, 4:7], [0, 0, -1.0], atol=2e-2)
    np.testing.assert_allclose(con[0, 1:4], [0, -r * np.cos(th) + h * np.sin(th), top - 0.00025], atol=5e-4)   # under the rim's lowest point
    # (5) clear of the face by 0.1 mm: no contact
    o.qpos[0] += 0.0006; o.forward()
    assert o.ncon == 0


def test_impratio_divides_the_friction_regularisation():
    """opt.impratio [MJ: mj_makeImpedance, pyramidal cone]: R of a contact's pyramid edges is 2 mu^2 R_n with the
    regularised mu = friction * sqrt(1 / impratio) -- at impratio 10 (the reference's hand XML) the edges
    are ten times stiffer than at MuJoCo's default 1; the Jacobian rows (the cone itself) do not change."""
    out = {}
    for ir in (1.0, 10.0):
        m, o = _cylinder_scene(impratio=ir)
        assert m.opt_impratio == ir
        o.qpos[0] = (0.02 + 0.01 - 0.0008) - 0.05; o.qvel[1] = 0.05; o.forward()
        assert o.ncon == 1 and o.nefc == 4
        out[ir] = (o.efc_D.copy(), o.efc_J.copy(), o.qacc.copy())
    np.testing.assert_allclose(out[10.0][0], 10.0 * out[1.0][0], rtol=1e-12)
    assert np.array_equal(out[10.0][1], out[1.0][1])
    assert np.abs(out[10.0][2] - out[1.0][2]).max() > 1e-3      # ... and the solution notices


def test_hull_fingertips_build_and_press_keys():
    """primitive_fingertip_collisions=False: the ten distal phalanges collide as 26-vertex hulls
    (the reference's default mesh mode, shadow_hand.py:105-107), one contact per fingertip-key pair."""
    from robopianist_amd import engine
    from robopianist_amd.model import spec
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        si = scene.build_scene(primitive_fingertip_collisions=False, gravity_compensation=True)
    m = si.model
    assert int((m.geom_type == spec.GEOM_MESH).sum()) == 10 and m.nmeshvert == 260
    o = Oracle(m, engine.make_blob(m, si.key_joint_ids))
    from test_gpu_parity import wrist_press_sequence      # scripted fingertips-onto-the-keys sequence
    mesh_key = 0
    for c in wrist_press_sequence(si, 300):
        o.ctrl[:] = c
        o.step(1)
        for cc in o.contact.reshape(-1, 16):
            g1, g2 = int(cc[13]), int(cc[14])
            if m.geom_type[g2] == spec.GEOM_MESH and m.names["geom"][g1].startswith("piano"):
                mesh_key += 1
    assert o.warnings == 0 and mesh_key > 50
    assert np.abs(o.qpos[:88]).max() > 0.01      # keys went down


# ---- the line search (PrimalSearch as restated in oracle/rp_oracle.c) on hand-made problems ------------
def _line_search(rows, quad, gtol, ls_iterations=50):
    """rows: list of (type, jar, jv, D, floss, R); type 0 friction loss, 1 limit, 2 contact."""
    import ctypes
    from oracle import rp_oracle
    L = rp_oracle.lib()
    n = len(rows)
    I = (ctypes.c_int * max(n, 1))(*[int(r[0]) for r in rows])
    cols = [(ctypes.c_double * max(n, 1))(*[float(r[k]) for r in rows]) for k in range(1, 6)]
    q = (ctypes.c_double * 3)(*quad)
    ev = ctypes.c_int(0)
    L.rpo_debug_line_search.restype = ctypes.c_double
    a = L.rpo_debug_line_search(n, I, cols[0], cols[1], cols[2], cols[3], cols[4], q, ctypes.c_double(gtol),
                                int(ls_iterations), ctypes.byref(ev))
    return a, ev.value


def _phi(rows, quad, a):
    """phi and phi' of the same problem, written independently of the C code."""
    c = quad[0] + quad[1] * a + quad[2] * a * a
    g = quad[1] + 2 * quad[2] * a
    for t, jar, jv, D, f, R in rows:
        x = jar + a * jv
        if t == 0:
            if x <= -R * f:
                c += -0.5 * R * f * f - f * x; g += -f * jv
            elif x >= R * f:
                c += -0.5 * R * f * f + f * x; g += f * jv
            else:
                c += 0.5 * D * x * x; g += D * x * jv
        elif x < 0:
            c += 0.5 * D * x * x; g += D * x * jv
    return c, g


def test_line_search_pure_quadratic_is_one_newton_step():
    # phi = 3 - 4 a + 2 a^2: minimum at a = 1, found by the first Newton point (2 evaluations: 0 and 1)
    a, ev = _line_search([], (3.0, -4.0, 2.0), 1e-10)
    assert a == 1.0 and ev == 2
    # ... also with rows that stay in one zone: an active contact row adds D (jar + a jv)^2 / 2
    rows = [(2, -1.0, 0.25, 8.0, 0, 0)]
    a, ev = _line_search(rows, (0.0, -1.0, 1.0), 1e-12)
    # phi' = -1 + 2a + 8 (−1 + a/4)/4 = -3 + 2.5 a  ->  a = 1.2 (row still active there: -1 + 0.3 < 0)
    assert abs(a - 1.2) < 1e-14 and ev == 2


def test_line_search_finds_the_minimum_across_zone_changes():
    """Random piecewise-quadratic line costs (limit / contact rows switching on and off, friction-loss rows
    crossing their three zones): the result satisfies PrimalSearch's own stopping rule, |phi'(alpha)| < gtol,
    never costs more than alpha = 0, and agrees with a bisection on phi' (which is monotone) to the accuracy
    gtol implies."""
    rng = np.random.default_rng(5)
    reached_bracket = 0
    for trial in range(300):
        n = int(rng.integers(1, 12))
        rows = []
        for _ in range(n):
            t = int(rng.integers(0, 3))
            D = float(10 ** rng.uniform(-1, 3))
            R = 1.0 / D
            rows.append((t, float(rng.normal()), float(rng.normal()), D, float(10 ** rng.uniform(-2, 0)), R))
        q2 = float(10 ** rng.uniform(-2, 1))
        # a descent direction: phi'(0) < 0
        c0, g0 = _phi(rows, (0.0, 0.0, q2), 0.0)
        q1 = -abs(float(rng.normal())) - max(0.0, g0) - 0.1
        quad = (0.0, q1, q2)
        gtol = 1e-9
        a, ev = _line_search(rows, quad, gtol)
        c, g = _phi(rows, quad, a)
        assert a > 0 and abs(g) < gtol, (trial, a, g)
        assert c <= _phi(rows, quad, 0.0)[0]
        assert ev <= 50
        reached_bracket += ev > 4
        # bisection on the monotone phi'
        lo, hi = 0.0, 1.0
        while _phi(rows, quad, hi)[1] < 0:
            hi *= 2
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            if _phi(rows, quad, mid)[1] < 0:
                lo = mid
            else:
                hi = mid
        # |phi'| < gtol and phi'' >= 2 q2 bound the distance to the root
        assert abs(a - 0.5 * (lo + hi)) <= gtol / (2 * q2) + 1e-12, (trial, a, lo, hi)
    assert reached_bracket > 30   # (the bracketed phase is exercised, not only the first Newton steps)


def test_line_search_respects_the_evaluation_limit():
    rows = [(2, -1.0 + 0.01 * k, 1.0, 50.0 * (k + 1), 0, 0) for k in range(10)]
    quad = (0.0, -5.0, 0.01)
    a_full, ev_full = _line_search(rows, quad, 1e-13)
    a_cut, ev_cut = _line_search(rows, quad, 1e-13, ls_iterations=3)
    assert ev_cut <= 4 and ev_full >= ev_cut     # (the limit is checked between evaluations, as in MuJoCo)
    assert _phi(rows, quad, a_cut)[0] <= _phi(rows, quad, 0.0)[0]


def test_hull_replay_chaos_control_and_tolerance_free_mpr_termination():
    """(1) The chaos control of the free-running parity figure.  On ROUNDS 1-5's stand-in (forearm box overlapping the
    palm, impratio 1) the Twinkle replay with hull fingertips was chaotic: the oracle started 1e-15 away from itself
    separated by more than north_star's 1e-4 within 1000 mj_steps, so no second implementation could be held to 1e-4
    free-running there.  On round 6's stand-in (the rigid-link overlap removed, the hand XML's impratio = 10) the same
    control stays below 1e-8: the trajectory is contractive again and the 1e-4 bar is asserted on the headline config
    (tests/test_gpu_parity.py::test_replay_fp64_1000_steps_hull).  (2) A tolerance-free termination of the portal
    refinement for polytope pairs (stop when the support vertex already is a portal vertex) against MuJoCo's uniform
    1e-6 rule: the same trajectory to 1e-9 over 600 mj_steps."""
    import os
    from robopianist_amd import engine
    from oracle import rp_oracle
    def build(prim, **kw):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=prim, **kw)
        m = si.model
        a = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "twinkle_twinkle_actions.npy")).astype(np.float64)[:, :-1]
        lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
        return m, engine.make_blob(m, si.key_joint_ids), lo + (np.clip(a, -1, 1) + 1.0) * 0.5 * (hi - lo)
    m, blob, ctrl = build(False)
    hull = [r["max_rel_qpos_error"] for r in rp_oracle.chaos_control(m, blob, ctrl, seeds=(0, 1, 2), eps0=1e-15)]
    assert max(hull) < 1e-8, hull
    m5, blob5, ctrl5 = build(False, standin_wrist_clearance=False, impratio=1.0)
    hull5 = [r["max_rel_qpos_error"] for r in rp_oracle.chaos_control(m5, blob5, ctrl5, seeds=(0, 1, 2), eps0=1e-15)]
    assert max(hull5) > 1e-4, hull5
    def traj(discrete):
        rp_oracle.set_mpr_experiment(1e-6, discrete)
        try:
            o = rp_oracle.Oracle(m, blob)
            out = np.zeros((600, m.nv))
            for i in range(600):
                o.ctrl[:] = ctrl[i // 10]
                o.step(1)
                out[i] = o.qpos
            return out
        finally:
            rp_oracle.set_mpr_experiment(1e-6, False)
    np.testing.assert_allclose(traj(False), traj(True), rtol=0, atol=1e-9)
    mc, blobc, ctrlc = build(True)
    cap = [r["max_rel_qpos_error"] for r in rp_oracle.chaos_control(mc, blobc, ctrlc, seeds=(0, 1), eps0=1e-15)]
    assert max(cap) < 1e-4, cap


def test_hull_support_walk_finds_the_scans_vertex():
    """Large hulls (more than 32 vertices) are supported by a walk over their vertex graph (model/hull.py,
    rp_oracle.c: geom_support): on a convex polytope the walk ends at the vertex a full scan finds, so the contact of
    a 300-vertex hull resting on a box equals the one computed with the graph switched off."""
    from robopianist_amd.model import spec
    i = np.arange(300) + 0.5
    z = 1.0 - 2.0 * i / 300
    phi = np.pi * (1.0 + 5.0 ** 0.5) * i
    r = np.sqrt(1.0 - z * z)
    verts = np.stack([0.03 * r * np.cos(phi), 0.02 * r * np.sin(phi), 0.015 * z], 1)
    def build():
        world = spec.Body(name="world")
        world.geoms.append(spec.Geom("floor", spec.GEOM_BOX, (0.05, 0.05, 0.01), pos=(0, 0, 0.01)))
        top = spec.Body(name="top", pos=(0, 0, 0.05), mass=0.1, inertia=(1e-4, 1e-4, 1e-4),
                        joints=[spec.Joint("z", type=spec.JNT_SLIDE, axis=(0, 0, 1), damping=0.5),
                                spec.Joint("rx", type=spec.JNT_HINGE, axis=(1, 0, 0), damping=0.01),
                                spec.Joint("ry", type=spec.JNT_HINGE, axis=(0, 1, 0), damping=0.01)],
                        geoms=[spec.Geom("hull", spec.GEOM_MESH, (0, 0, 0), vertices=verts.tolist())])
        world.add(top)
        return mc.compile_scene(spec.Scene(world=world))
    m = build()
    o = Oracle(m, mc.to_blob(m))
    assert int(m.geom_vertgraph.sum()) == 1 and int(m.mesh_graph[:, 0].max()) <= 23 and int(m.nmeshvert) == 300
    m2 = build()
    m2["geom_vertgraph"] = np.zeros_like(m2["geom_vertgraph"])      # same hull, scanned
    o2 = Oracle(m2, mc.to_blob(m2))
    rng = np.random.default_rng(0)
    hits = 0
    for _ in range(60):
        q = np.array([rng.uniform(-0.03, -0.005), rng.uniform(-3, 3), rng.uniform(-3, 3)])
        for oo in (o, o2):
            oo.qpos[:] = q; oo.forward()
        assert o.ncon == o2.ncon
        if o.ncon:
            np.testing.assert_allclose(o.contact.reshape(-1, 16)[:, :7], o2.contact.reshape(-1, 16)[:, :7], rtol=0, atol=1e-12)
            hits += 1
    assert hits >= 20


def test_oracle_reproduces_its_regression_fixture():
    """tests/golden/oracle_regression.npz (make_oracle_regression.py): the oracle's own trajectory on the benchmark
    scene and action stream, both fingertip colliders, inside the smooth window.  Pins the oracle and the model
    builders against accidental change -- NOT against MuJoCo (parity unpinned)."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec_ = importlib.util.spec_from_file_location("make_oracle_regression", os.path.join(here, "golden", "make_oracle_regression.py"))
    mod = importlib.util.module_from_spec(spec_); spec_.loader.exec_module(mod)
    ref = np.load(os.path.join(here, "golden", "oracle_regression.npz"))
    assert tuple(ref["marks"]) == mod.MARKS
    for name, prim in (("hull", False), ("capsule", True)):
        q, ncon = mod.rollout(prim)
        assert np.array_equal(ncon, ref[f"ncon_{name}"]), name
        np.testing.assert_allclose(q, ref[f"qpos_{name}"], rtol=0, atol=1e-10, err_msg=name)
