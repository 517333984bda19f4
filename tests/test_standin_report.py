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
    assert r["pair_residency_top"][0]["pair"][0].endswith("forearm_box")   # (forearm box vs palm box: the most resident pair)


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
    assert deepest is None or deepest["max_depth_m"] < 0.02   # (nothing grotesque: under 2 cm)
