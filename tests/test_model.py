"""Structural tests of the compiled scene, mirroring the reference's
models/piano/piano_test.py:25-65 and models/hands/shadow_hand_test.py:44-124."""
import warnings

import numpy as np
import pytest

from robopianist_amd.model import compile as mc
from robopianist_amd.model import engine_tables, piano, scene, shadow_hand, spec


def test_piano_counts_and_order(piano_only_scene):
    m = piano_only_scene.model
    assert len(piano_only_scene.key_joint_ids) == 88 == piano.NUM_KEYS
    assert list(piano_only_scene.key_joint_ids) == list(range(88))  # sorted by key id
    # white / black alternate like a real keyboard, 52 + 36
    blacks = [piano.is_key_black(k) for k in range(88)]
    assert sum(blacks) == 36 and not blacks[0] and blacks[1]
    ys = m.body_pos[piano_only_scene.key_body_ids, 1]
    assert np.all(np.diff(ys) > 0)  # left to right
    assert m.nu == 88 and m.nv == 88
    # known answers derived from piano_constants.py
    assert m.dof_M0[0] == pytest.approx(1.3017e-3, rel=1e-3)
    assert m.dof_M0[1] == pytest.approx(1.0545e-3, rel=1e-3)
    assert m.jnt_range[0, 1] == pytest.approx(np.arctan(0.01 / 0.15))
    assert m.jnt_range[1, 1] == pytest.approx(np.arctan(0.008 / 0.09))
    assert piano.PIANO_LENGTH == pytest.approx(1.221)
    assert m.npair == 0  # keys never collide with keys or base


def test_two_hand_scene_counts(two_hand_scene):
    m = two_hand_scene.model
    assert m.nq == m.nv == 140 and m.nu == 44 and m.ntendon == 8
    for side in ("right", "left"):
        h = two_hand_scene.hands[side]
        assert len(h.joint_ids) == shadow_hand.NQ + 2
        assert len(h.actuator_ids) == shadow_hand.NU + 2
        names = [m.names["joint"][j].split("/")[-1] for j in h.joint_ids]
        assert names[0].endswith("WRJ2")                     # shadow_hand_test.py:101-106
        assert names[-2:] == ["forearm_tx", "forearm_ty"]
        assert len(h.fingertip_site_ids) == 5
        tips = [m.names["site"][s].split("/")[-1] for s in h.fingertip_site_ids]
        assert [t[3:5] for t in tips] == ["th", "ff", "mf", "rf", "lf"]
    # dof order: keys, right hand, left hand
    assert m.names["joint"][88].startswith("rh_shadow_hand")
    assert m.names["joint"][114].startswith("lh_shadow_hand")
    # forearm_tx spans the keyboard (base.py:160-163,189-194)
    r = two_hand_scene.hands["right"]
    tx = r.joint_ids[-2]
    np.testing.assert_allclose(m.jnt_range[tx], [-0.6105 - 0.15, 0.6105 - 0.15])
    # critical damping 2 sqrt(M0 k) (shadow_hand.py:299-301)
    assert m.dof_damping[tx] == pytest.approx(2 * np.sqrt(m.dof_M0[tx] * 300.0))
    # position actuators: gain kp, bias -kp
    np.testing.assert_allclose(m.actuator_biasprm[:, 1], -m.actuator_gainprm)


def test_invalid_forearm_dof_raises():
    with pytest.raises(ValueError):
        shadow_hand.HandBuilder("right", forearm_dofs=("forearm_bogus",))


@pytest.mark.parametrize("reduced", [False, True])
@pytest.mark.parametrize("dofs", [(), ("forearm_tx",), ("forearm_tx", "forearm_ty", "forearm_roll")])
def test_hand_variants(reduced, dofs):
    hb = shadow_hand.HandBuilder("left", forearm_dofs=dofs, reduced_action_space=reduced)
    assert len(hb.joint_names) == shadow_hand.NQ + len(dofs) - 3 * reduced
    assert len(hb.actuator_names) == shadow_hand.NU + len(dofs) - 3 * reduced


def test_disable_hand_collisions_leaves_only_hand_piano_pairs():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        si = scene.build_scene(disable_hand_collisions=True, primitive_fingertip_collisions=True)
    m = si.model
    piano_geoms = set(int(g) for g in si.key_geom_ids) | {0}
    for a, b in m.pair_geom:
        assert (int(a) in piano_geoms) != (int(b) in piano_geoms)


def test_palm_boxes_cannot_reach_keys(two_hand_scene):
    """Justifies dropping box-box pairs (compile.py): within joint ranges the hands'
    box colliders stay above the highest key surface."""
    m = two_hand_scene.model
    rng = np.random.default_rng(0)
    key_top = 0.0225 + 0.0125 + 0.002
    boxes = [g for g in range(m.ngeom) if m.geom_type[g] == spec.GEOM_BOX
             and m.geom_bodyid[g] > 89]
    lowest = np.inf
    for _ in range(200):
        q = np.zeros(m.nv)
        for j in range(88, m.nv):
            q[j] = rng.uniform(*m.jnt_range[j])
        kin = mc.kinematics(m, q)
        for g in boxes:
            b = m.geom_bodyid[g]
            c = kin["xpos"][b] + kin["xmat"][b] @ m.geom_pos[g]
            lowest = min(lowest, c[2] - m.geom_rbound[g])
    assert lowest > key_top


def test_blob_round_trip_and_engine_tables(two_hand_scene):
    m = two_hand_scene.model
    t = engine_tables.build_engine_tables(m, two_hand_scene.key_joint_ids)
    assert t["eng_nlink"][0] == 52 and t["eng_ntree"][0] == 2 and t["eng_maxdepth"][0] == 9
    assert t["eng_npair"][0] + t["eng_nkeycap"][0] * 88 == m.npair
    blob = mc.to_blob(m, extra=t)
    assert blob[:4] == (0x52504D42).to_bytes(4, "little")
    # every link's ancestors have smaller lane ids (the kernels rely on it)
    anc = t["eng_link_anc"]
    for i in range(52):
        a = anc[i][anc[i] >= 0]
        assert np.all(a <= i) and a[-1] == i
    # descendant table is the transpose of the ancestor table
    desc = t["eng_link_desc"]
    for i in range(52):
        for d in range(9):
            for k in desc[i, d]:
                if k >= 0:
                    assert anc[k][t["eng_link_depth"][i]] == i and t["eng_link_depth"][k] == d


def test_welded_bodies_are_fused_into_their_parent_links():
    """reduced_action_space removes three joints per hand (shadow_hand.py:73-77,162-171): the
    finger segments that lose their joint stay rigidly attached to their parent.  The engine
    tables fuse them: link mass / first moment add up to the bodies', every hand geom and site
    still maps to a link, and geoms of a welded body carry the static offset."""
    import warnings
    from robopianist_amd.model import scene, spec
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        si = scene.build_scene(reduced_action_space=True, primitive_fingertip_collisions=True)
    m = si.model
    t = engine_tables.build_engine_tables(m, si.key_joint_ids)
    nl = int(t["eng_nlink"][0])
    assert nl == 2 * (21 + 2)
    hand_bodies = [b for b in range(1, m.nbody) if m.body_weldid[b] != 0 and b not in set(si.key_body_ids)]
    welded = [b for b in hand_bodies if m.body_jntnum[b] == 0]
    assert len(welded) == 6  # thbase/thhub..., thdistal, lfmetacarpal per hand
    np.testing.assert_allclose(t["eng_link_mass"].sum(), m.body_mass[hand_bodies].sum(), rtol=1e-14)
    # every fingertip site still resolves to a link
    assert int(t["eng_nsite"][0]) >= 10 and (t["eng_site_link"] >= 0).all()
    # world pose of a welded body's geom through the tables == through the body tree, at qpos0
    kin = mc.kinematics(m, m.qpos0)
    checked = 0
    for i, g in enumerate(t["eng_geom_modelid"]):
        b = int(m.geom_bodyid[g])
        if b in welded:
            a = b
            while m.body_jntnum[a] == 0:
                a = int(m.body_parentid[a])
            want = kin["xpos"][b] + kin["xmat"][b] @ m.geom_pos[g]
            got = kin["xpos"][a] + kin["xmat"][a] @ t["eng_geom_pos"][i]
            np.testing.assert_allclose(got, want, atol=1e-14)
            checked += 1
    assert checked >= 2  # the fingertip capsules of the welded thumb distal segments


# ---- pin of "the piano half is exact": the in-tree piano against what the REFERENCE's own builder
# builds (tests/golden/piano_layout.json, produced by running robopianist/models/piano/piano_mjcf.py
# against a recording mjcf stand-in: tests/golden/make_piano_golden.py)
def _golden_piano():
    import json, os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "piano_layout.json")))


def test_piano_golden_is_current_when_the_reference_is_present():
    import os, subprocess, sys
    if not os.path.isdir(os.environ.get("RP_REFERENCE", "/root/reference")):
        pytest.skip("reference checkout not present (GPU box)")
    script = os.path.join(os.path.dirname(__file__), "golden", "make_piano_golden.py")
    assert subprocess.run([sys.executable, script, "--check"]).returncode == 0


def test_every_piano_constant_matches_the_reference():
    g = _golden_piano()
    ours = {k: v for k, v in vars(piano).items() if k.isupper()}
    alias = {"WHITE_KEY_SPRINGREF": "KEY_SPRINGREF_DEG", "BLACK_KEY_SPRINGREF": "KEY_SPRINGREF_DEG",
             "WHITE_KEY_STIFFNESS": "KEY_STIFFNESS", "BLACK_KEY_STIFFNESS": "KEY_STIFFNESS",
             "WHITE_JOINT_DAMPING": "KEY_DAMPING", "BLACK_JOINT_DAMPING": "KEY_DAMPING",
             "WHITE_JOINT_ARMATURE": "KEY_ARMATURE", "BLACK_JOINT_ARMATURE": "KEY_ARMATURE"}
    cosmetic = {"WHITE_KEY_COLOR", "BLACK_KEY_COLOR", "BASE_COLOR", "ACTUATOR_DYNPRM", "ACTUATOR_GAINPRM"}
    checked = 0
    for name, ref in g["piano_constants"].items():
        if name in cosmetic:
            continue
        mine = ours[alias.get(name, name)]
        assert np.array_equal(np.asarray(mine, float), np.asarray(ref, float)), name   # bit-equal
        checked += 1
    assert checked >= 35
    from robopianist_amd.music import constants as mconst
    for name, ref in g["music_constants"].items():
        assert getattr(mconst, name) == ref, name
    assert mconst.NOTES == g["music_notes"]


@pytest.mark.parametrize("actuated", [False, True])
def test_compiled_piano_equals_the_reference_builders_output(actuated):
    g = _golden_piano()["piano_with_actuators" if actuated else "piano"]
    assert g["compiler"] == {"angle": "radian", "autolimits": True}
    base, keys, acts = piano.build(add_actuators=actuated)
    gb = g["bodies"]
    assert len(gb) == 89 and gb[0]["name"] == "base"
    assert np.array_equal(np.asarray(base.pos, float), np.asarray(gb[0]["pos"], float))
    assert np.array_equal(np.asarray(base.geoms[0].size, float), np.asarray(gb[0]["geom"]["size"], float))
    assert (base.geoms[0].contype, base.geoms[0].conaffinity) == (gb[0]["geom"]["contype"], gb[0]["geom"]["conaffinity"])
    for k, (mine, ref) in enumerate(zip(keys, gb[1:])):
        assert mine.name == "piano/" + ref["name"], k          # key id order == sorted body order
        eq = lambda a, b: np.array_equal(np.asarray(a, float), np.asarray(b, float))
        assert eq(mine.pos, ref["pos"]), (k, mine.pos, ref["pos"])
        gm, gr = mine.geoms[0], ref["geom"]
        assert gm.name == "piano/" + gr["name"] and gr["type"] == "box" and gm.type == spec.GEOM_BOX
        assert eq(gm.size, gr["size"]) and gm.mass == gr["mass"]
        assert (gm.contype, gm.conaffinity) == (gr["contype"], gr["conaffinity"])
        jm, jr = mine.joints[0], ref["joint"]
        assert jm.name == "piano/" + jr["name"] and jr["type"] == "hinge" and jm.type == spec.JNT_HINGE
        for field in ("pos", "axis", "range"):
            assert eq(getattr(jm, field), jr[field]), (k, field)
        for field in ("stiffness", "springref", "damping", "armature"):
            assert float(getattr(jm, field)) == float(jr[field]), (k, field)
        assert mine.sites[0].name == "piano/" + ref["site"]["name"]
    if actuated:
        assert len(acts) == 88 == len(g["actuators"])
        for am, ar in zip(acts, g["actuators"]):
            assert am.name == "piano/" + ar["name"] and am.joint == "piano/" + ar["joint"]
            assert ar["gaintype"] == "fixed" and ar["biastype"] == "none" and ar["dyntype"] == "none"
            assert am.gain == ar["gainprm"][0] and tuple(am.bias) == tuple(float(x) for x in ar["biasprm"])
            assert np.array_equal(np.asarray(am.ctrlrange, float), np.asarray(ar["ctrlrange"], float))
