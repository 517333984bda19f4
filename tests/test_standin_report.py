"""How much of the headline workload is a stand-in artefact: the report itself is under test
here -- that it runs, what it counts, and the findings DESIGN.md quotes (so that a change of the stand-in geometry that
moves them is noticed)."""
import os
import sys
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def hull_scene():
    from robopianist_amd.model import scene
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=False)


def _replay_rows(m):
    a = np.load(os.path.join(ROOT, "tests", "golden", "twinkle_twinkle_actions.npy")).astype(np.float64)[:, :-1]
    lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
    return lo + (np.clip(a, -1, 1) + 1.0) * 0.5 * (hi - lo)


def test_contact_residency_along_the_replay(hull_scene):
    from oracle import standin_report
    from robopianist_amd import engine
    m = hull_scene.model
    r = standin_report.contact_residency(m, engine.make_blob(m, hull_scene.key_joint_ids), _replay_rows(m))
    s = r["share_of_contacts"]
    print(r)
    assert r["mj_steps"] == 1580 and abs(sum(s.values()) - 1.0) < 1e-12
    # the finding: most contacts of the replay are the stand-in hand touching ITSELF (the policy was trained on the real
    # hand); keys are a minority.  (Round 4's judge counted 2366 self vs 785 key contacts in 600 mj_steps.)
    assert 0.5 < s["hand_self"] < 0.9 and s["hand_key"] > 0.1
    # (rounds 1-5: the most resident pair was forearm_box vs a palm box, in contact on 43-61 % of the mj_steps -- two RIGID
    # links overlapping at the end of WRJ2's range.  Round 6 moved the box (model/shadow_hand.py: standin_wrist_clearance,
    # now the default): that pair must be gone from the list, and the contacts per mj_step went from 6.7 to 5.0.)
    assert not any("forearm_box" in p for row in r["pair_residency_top"] for p in row["pair"]), r["pair_residency_top"]
    assert r["mean_contacts"] < 5.6, r["mean_contacts"]


def test_the_rounds_1_to_5_geometry_still_shows_the_forearm_box_artefact():
    """The finding that motivated the change stays reproducible: with standin_wrist_clearance=False the forearm's wrist
    box is the most resident contact pair of the replay."""
    from oracle import standin_report
    from robopianist_amd import engine
    from robopianist_amd.model import scene
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=False, standin_wrist_clearance=False, impratio=1.0)
    r = standin_report.contact_residency(si.model, engine.make_blob(si.model, si.key_joint_ids), _replay_rows(si.model)[:80])
    assert r["pair_residency_top"][0]["pair"][0].endswith("forearm_box"), r["pair_residency_top"][:3]


def test_joint_range_sweep_lists_the_self_penetrations_of_the_standin(hull_scene):
    from oracle import standin_report
    from robopianist_amd import engine
    m = hull_scene.model
    rows = standin_report.joint_range_sweep(m, engine.make_blob(m, hull_scene.key_joint_ids), samples=7)
    print(rows[:12])
    assert all(set(r) == {"joint", "pair", "max_depth_m", "at_q"} for r in rows)
    # at qpos0 nothing touches (reset pose: no contacts), so every finding below is a single joint driven to an end of
    # ITS OWN range -- the overlaps the stand-in's from-memory collision boxes allow and a real hand's cannot
    deepest = rows[0] if rows else None
    assert deepest is None or deepest["max_depth_m"] < 0.022   # (nothing grotesque: about 2 cm)
    # ROUND 6: no overlap between non-adjacent RIGID links is left.  What the sweep still finds is (a) a finger abducted
    # (xFJ4) into its NEIGHBOUR -- two different fingers, which a real hand's fingers do to each other as well -- and (b) a
    # proximal phalanx folded onto the palm / metacarpal at the very end of xFJ3's range, by 1.2 mm at most.  Nothing
    # involves the forearm, the wrist or two palm boxes.
    for r in rows:
        j, (a, b) = r["joint"], r["pair"]
        assert "forearm" not in a and "forearm" not in b and "wrist" not in a and "wrist" not in b, r
        if j[3:] in ("FFJ4", "MFJ4", "RFJ4", "LFJ4"):
            assert a[3:5] != b[3:5], r                      # (two different fingers)
        else:
            assert j[3:] in ("FFJ3", "MFJ3", "RFJ3", "LFJ3") and r["max_depth_m"] < 1.3e-3, r
