"""Pins against REAL MuJoCo, the moment its golden files exist (oracle/make_golden.py writes them on a machine
with `mujoco` + the menagerie hand: tests/golden/mujoco/config<N>.npz), and -- always -- the importer that makes
them usable: mjModel dump -> compile.Model -> blob (robopianist_amd/tools/mjmodel_to_blob.py).

  * round trip on the stand-in scene (runs here): Model -> MuJoCo-shaped dump -> Model, identical arrays, identical
    blob, identical oracle trajectory;
  * golden consumption (skipped while no file is present; reference pin: /root/reference/setup.py:39 mujoco>=3.1.1):
    the ORACLE and, on a GPU, the ENGINE replay MuJoCo's own recorded rollout of BASELINE configs 2-4 on MuJoCo's
    own model -- teacher-forced per mj_step at 1e-9 with equal contact counts, free-running at 1e-4 over 1000 steps.
"""
import glob
import os
import sys
import warnings

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "mujoco", "config*.npz")))


def _standin(**kw):
    from robopianist_amd.model import scene
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return scene.build_scene(gravity_compensation=True, **kw)


def _blob_entries(blob: bytes):
    """name -> (dtype code, raw bytes) of a model blob (compile.to_blob)."""
    import struct
    magic, version, n = struct.unpack_from("<III", blob, 0)
    out = {}
    for i in range(n):
        off = 12 + i * 64
        name = blob[off:off + 40].split(b"\0")[0].decode()
        dt, ndim, count, o = struct.unpack_from("<iiqq", blob, off + 40)
        out[name] = (dt, blob[o:o + count * (8 if dt == 0 else 4)])
    return out


@pytest.mark.parametrize("primitive", [True, False, "large hulls", "cylinders"])
def test_importer_round_trip_on_the_standin(tmp_path, primitive):
    """(third case: every hand collider a 200-vertex convex hull -- what a dump of the real hand carries, where forearm,
    wrist, palm, thumb links and fingertips all are meshes; the importer rebuilds the vertex graphs of the support walk.
    Fourth case (round 6): the wrist / knuckle colliders as CYLINDERS, mjGEOM_CYLINDER = 5, with opt.impratio = 10 --
    the two things the reference's hand XML holds that rounds 1-5 could not import, shadow_hand.py:122,144-152)"""
    from robopianist_amd import engine
    from robopianist_amd.tools import mjmodel_to_blob as imp
    si = _standin(primitive_fingertip_collisions=False, mesh_colliders=200) if primitive == "large hulls" else (
        _standin(primitive_fingertip_collisions=False, cylinder_colliders=True, impratio=10.0) if primitive == "cylinders" else
        _standin(primitive_fingertip_collisions=primitive))
    if primitive == "cylinders":
        assert int((si.model.geom_type == 5).sum()) == 10 and si.model.opt_impratio == 10.0
    path = os.path.join(tmp_path, "standin.npz")
    np.savez_compressed(path, **imp.npz_from_model(si.model))
    m2, keys = imp.model_from_npz(path)
    assert np.array_equal(keys, si.key_joint_ids)
    for k, v in si.model.items():
        if isinstance(v, np.ndarray):
            assert k in m2, k
            assert np.array_equal(np.asarray(m2[k], v.dtype).reshape(v.shape), v), k
        elif isinstance(v, (int, float)):
            assert m2[k] == v, (k, m2[k], v)
    assert m2.names == si.model.names
    # same blob content (the table of named arrays; the order of the entries follows dict insertion order)
    a, b = _blob_entries(engine.make_blob(m2, keys)), _blob_entries(engine.make_blob(si.model, si.key_joint_ids))
    assert sorted(a) == sorted(b)
    for k in a:
        assert a[k] == b[k], k


@pytest.mark.parametrize("field,value,what", [
    ("model_neq", 1, "equality"), ("model_npair", 2, "contact pairs"), ("model_wrap_type", 3, "tendon wrapping"),
    ("model_actuator_gaintype", 1, "gain"), ("model_actuator_biastype", 2, "bias"), ("model_actuator_dyntype", 1, "activation"),
    ("model_opt_mpr", [1e-8, 50.0], "tolerance"), ("model_opt", None, "cone"),
    ("model_tendon_limited", 1, "tendon limits"), ("model_tendon_frictionloss", 0.01, "tendon frictionloss"),
    ("model_tendon_damping", 0.1, "tendon damping"), ("model_tendon_stiffness", 1.0, "tendon stiffness"),
    ("model_opt_disableflags_other", 1 << 4, "disableflags"), ("model_opt_enableflags", 1, "enableflags"),
    ("model_opt_nativeccd", 1, "native convex collision"),
    ("model_geom_priority", 1, "geom_priority"), ("model_geom_condim", 4, "condim"), ("model_geom_solmix", 0.5, "solmix"),
    ("model_geom_margin", 0.001, "margin")])
def test_importer_rejects_dynamics_the_engine_does_not_model(field, value, what):
    """A real-MuJoCo dump with equality constraints, explicit contact pairs, spatial tendons, non-position actuators,
    activation dynamics, an elliptic cone or a non-default convex-collision tolerance must raise, not import.  Round 6:
    neither may tendon limits / frictionloss / damping / stiffness, disable / enable flags beyond refsafe, a recording
    made with the native (GJK / EPA) convex pipeline, or colliding geoms with a priority, condim != 3, solmix != 1 or a
    margin -- everything the restatement does not model is rejected, never imported silently."""
    from robopianist_amd.model import scene
    from robopianist_amd.tools import mjmodel_to_blob as imp
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
    d = imp.npz_from_model(si.model)
    imp.model_from_npz(dict(d))   # the supported dump passes
    bad = dict(d)
    if field == "model_opt":
        o = np.array(d["model_opt"], float); o[6] = 1; bad[field] = o   # elliptic cone
    elif np.ndim(d[field]) == 0 or isinstance(value, list):
        bad[field] = np.asarray(value)
    else:
        v = np.array(d[field]); v[-1] = value; bad[field] = v   # (the last item: for geom arrays a hand collider)
    with pytest.raises(ValueError, match=what):
        imp.model_from_npz(bad)


@pytest.mark.parametrize("mj_type,name", [(4, "ellipsoid"), (2, "sphere"), (0, "plane")])
def test_importer_rejects_collision_geoms_the_narrow_phase_does_not_handle(mj_type, name):
    """A real dump whose colliding geoms include an ellipsoid / sphere / plane must raise and NAME the geoms
    (the oracle's pair loop would otherwise skip those pairs silently); the same type on a visual geom (contype =
    conaffinity = 0) imports."""
    from robopianist_amd.model import scene
    from robopianist_amd.tools import mjmodel_to_blob as imp
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
    d = imp.npz_from_model(si.model)
    gn = [str(x) for x in d["names_geom"]]
    g = next(i for i, n in enumerate(gn) if n.endswith("wrist_col"))
    bad = dict(d)
    t = np.array(d["model_geom_type"]); t[g] = mj_type; bad["model_geom_type"] = t
    with pytest.raises(ValueError, match=r"unsupported collision geom types.*wrist_col \(%s\)" % name):
        imp.model_from_npz(bad)
    ok = dict(bad)
    for k in ("model_geom_contype", "model_geom_conaffinity"):
        v = np.array(d[k]); v[g] = 0; ok[k] = v
    imp.model_from_npz(ok)   # a visual geom of that type never collides


def test_oracle_steps_the_imported_model_like_the_compiled_one(tmp_path):
    from robopianist_amd import engine
    from robopianist_amd.tools import mjmodel_to_blob as imp
    from oracle.rp_oracle import Oracle
    si = _standin(primitive_fingertip_collisions=True)
    path = os.path.join(tmp_path, "standin.npz")
    np.savez_compressed(path, **imp.npz_from_model(si.model))
    a = Oracle(si.model, engine.make_blob(si.model, si.key_joint_ids))
    m2, keys = imp.model_from_npz(path)
    b = Oracle(m2, imp.blob_from_npz(path))
    rng = np.random.default_rng(0)
    lo, hi = si.model.actuator_ctrlrange[:, 0], si.model.actuator_ctrlrange[:, 1]
    for t in range(40):
        c = lo + rng.uniform(0.1, 0.9, si.model.nu) * (hi - lo)
        a.ctrl[:] = c; b.ctrl[:] = c
        a.step(1); b.step(1)
        assert np.array_equal(a.qpos, b.qpos) and a.ncon == b.ncon


# ---- golden consumption -------------------------------------------------------------------------------------
def _load_golden(path):
    from robopianist_amd.tools import mjmodel_to_blob as imp
    d = dict(np.load(path, allow_pickle=False))
    model, keys = imp.model_from_npz(d)
    from robopianist_amd import engine
    return d, model, keys, engine.make_blob(model, keys)


def _teacher_forced(step_fn, d, nsub):
    """step_fn(qpos, qvel, warm, ctrl) -> (qpos', qvel', ncon).  Worst per-step discrepancy relative to the step's
    largest velocity change, over the recorded rollout."""
    worst = 0.0
    T = d["ctrl"].shape[0]
    for t in range(T):
        for s in range(nsub):
            i = t * nsub + s
            q, v, nc = step_fn(d["qpos"][i], d["qvel"][i], d["qacc_warmstart"][i], d["ctrl"][t])
            assert nc == int(d["ncon"][i]), (i, nc, int(d["ncon"][i]))
            den = max(np.abs(d["qvel"][i + 1] - d["qvel"][i]).max(), 1e-9)
            worst = max(worst, np.abs(v - d["qvel"][i + 1]).max() / den, np.abs(q - d["qpos"][i + 1]).max() / max(np.abs(d["qpos"][i + 1]).max(), 1e-2))
    return worst


def _check_oracle(path, free_tol=1e-4):
    from oracle.rp_oracle import Oracle
    d, model, keys, blob = _load_golden(path)
    nsub = int(d["n_substeps"]) if "n_substeps" in d else 10
    orc = Oracle(model, blob)

    def step(q, v, w, c):
        orc.qpos[:] = q; orc.qvel[:] = v; orc.qacc_warmstart[:] = w; orc.ctrl[:] = c
        # (round 5: the stage data of the WRITTEN state first -- `step` is mj_step2; mj_step1 and would otherwise solve
        # on the data of the oracle's own previous state, which only a recording made by this oracle reproduces)
        orc.step1()
        orc.step(1)
        return orc.qpos.copy(), orc.qvel.copy(), orc.ncon
    assert _teacher_forced(step, d, nsub) < 1e-9
    # free-running: identical actions from the reset state, 1e-4 relative over (up to) 1000 mj_steps
    orc.reset(); orc.qpos[:] = d["qpos"][0]; orc.qvel[:] = d["qvel"][0]; orc.qacc_warmstart[:] = d["qacc_warmstart"][0]
    n = min(1000, d["qpos"].shape[0] - 1)
    for i in range(n):
        orc.ctrl[:] = d["ctrl"][i // nsub]
        orc.step(1)
    rel = np.abs(orc.qpos - d["qpos"][n]) / np.maximum(np.abs(d["qpos"][n]), 1e-2)
    assert rel.max() < free_tol, rel.max()


def _check_engine(path, free_tol=1e-4):
    from robopianist_amd import engine
    d, model, keys, blob = _load_golden(path)
    nsub = int(d["n_substeps"]) if "n_substeps" in d else 10
    phys = engine.BatchedPhysics(model, keys, n_envs=2, precision=64, blob=blob)

    def step(q, v, w, c):
        phys.set(engine.QPOS, q[None, :]); phys.set(engine.QVEL, v[None, :])
        phys.set(engine.QACC_WARMSTART, w[None, :]); phys.set(engine.CTRL, c[None, :])
        phys.step(1)
        assert phys.warn_flags.max() == 0
        return phys.qpos[0].astype(np.float64), phys.qvel[0].astype(np.float64), int(phys.get(engine.NCON)[0])
    assert _teacher_forced(step, d, nsub) < 1e-9
    phys.set(engine.QPOS, d["qpos"][0][None, :]); phys.set(engine.QVEL, d["qvel"][0][None, :])
    phys.set(engine.QACC_WARMSTART, d["qacc_warmstart"][0][None, :])
    n = min(1000, d["qpos"].shape[0] - 1)
    for i in range(0, n, nsub):
        phys.set(engine.CTRL, d["ctrl"][i // nsub][None, :])
        phys.step(min(nsub, n - i))
    q = phys.qpos[0].astype(np.float64)
    rel = np.abs(q - d["qpos"][n]) / np.maximum(np.abs(d["qpos"][n]), 1e-2)
    assert rel.max() < free_tol, rel.max()


@pytest.mark.skipif(not GOLDEN, reason="no tests/golden/mujoco/config*.npz (oracle/make_golden.py needs mujoco + the menagerie)")
@pytest.mark.parametrize("path", GOLDEN)
def test_oracle_matches_mujoco(path):
    _check_oracle(path)


@pytest.mark.gpu
@pytest.mark.skipif(not GOLDEN, reason="no tests/golden/mujoco/config*.npz (oracle/make_golden.py needs mujoco + the menagerie)")
@pytest.mark.parametrize("path", GOLDEN)
def test_engine_matches_mujoco(path):
    _check_engine(path)


# ---- the same pathway on a file of the same format that CAN be made here: the recorder is the oracle on the
# stand-in scene instead of MuJoCo on the menagerie hand (this validates the loader, the importer and the replay
# logic, not the physics: for the engine it is the usual oracle-vs-HIP parity, reached through the golden format)
@pytest.fixture(scope="module")
def synthetic_golden(tmp_path_factory):
    from robopianist_amd import engine
    from robopianist_amd.tools import mjmodel_to_blob as imp
    from oracle.rp_oracle import Oracle
    si = _standin(primitive_fingertip_collisions=True)
    orc = Oracle(si.model, engine.make_blob(si.model, si.key_joint_ids))
    rng = np.random.default_rng(3)
    lo, hi = si.model.actuator_ctrlrange[:, 0], si.model.actuator_ctrlrange[:, 1]
    T, nsub = 12, 10
    ctrl = lo + rng.uniform(0.1, 0.9, (T, si.model.nu)) * (hi - lo)
    qpos, qvel, warm, ncon = [orc.qpos.copy()], [orc.qvel.copy()], [orc.qacc_warmstart.copy()], []
    for t in range(T):
        orc.ctrl[:] = ctrl[t]
        for _ in range(nsub):
            orc.step(1)
            qpos.append(orc.qpos.copy()); qvel.append(orc.qvel.copy()); warm.append(orc.qacc_warmstart.copy()); ncon.append(orc.ncon)
    path = os.path.join(tmp_path_factory.mktemp("golden"), "config_synthetic.npz")
    np.savez_compressed(path, ctrl=ctrl, qpos=np.asarray(qpos), qvel=np.asarray(qvel), qacc_warmstart=np.asarray(warm),
                        ncon=np.asarray(ncon), n_substeps=np.asarray(nsub), **imp.npz_from_model(si.model))
    return path


def test_narrow_phase_variants_can_be_bisected_on_a_recording(tmp_path):
    """oracle/bisect_golden.py: the choices DESIGN section 8 lists as not reproducible from memory (capsule-box second
    point, box-box point count, MPR stopping rule) are switches of the oracle; a recording replayed under each tells
    which rule its author followed.  Here the recorder is the oracle itself along the replay (capsule-box pairs with
    two points): the default combination reproduces it exactly, the other capsule-box rules show up as
    contact-count mismatches -- i.e. the switches are live and the tool ranks them."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_parity as tgp
    from oracle import bisect_golden, rp_oracle
    from oracle.rp_oracle import Oracle
    from robopianist_amd import engine
    from robopianist_amd.tools import mjmodel_to_blob as imp
    si = _standin(primitive_fingertip_collisions=True)
    orc = Oracle(si.model, engine.make_blob(si.model, si.key_joint_ids))
    ctrl = tgp._replay_ctrl(si)[500:620]   # (one row per mj_step: proximal links lying on palm boxes, fingers on keys)
    rec = dict(qpos=[orc.qpos.copy()], qvel=[orc.qvel.copy()], qacc_warmstart=[orc.qacc_warmstart.copy()], ncon=[])
    for c in ctrl:
        orc.ctrl[:] = c
        orc.step(1)
        rec["qpos"].append(orc.qpos.copy()); rec["qvel"].append(orc.qvel.copy())
        rec["qacc_warmstart"].append(orc.qacc_warmstart.copy()); rec["ncon"].append(orc.ncon)
    assert max(rec["ncon"]) >= 6
    d = dict(ctrl=ctrl, n_substeps=np.asarray(1), **{k: np.asarray(v) for k, v in rec.items()}, **imp.npz_from_model(si.model))
    rows = bisect_golden.bisect(d, grid=[(0, 8, "uniform"), (1, 8, "uniform"), (2, 8, "uniform"), (0, 3, "uniform")])
    by = {(r["capsule_box"], r["boxbox_max"]): r for r in rows}
    assert by[(0, 8)]["first_count_mismatch"] == -1 and by[(0, 8)]["worst_rel_dv"] == 0.0
    assert by[(1, 8)]["first_count_mismatch"] >= 0   # (one point per pair: not the recorder's rule)
    assert by[(2, 8)]["first_count_mismatch"] >= 0   # (the two ends only: neither)
    # the switches are process-wide: the tool must leave the defaults behind
    orc2 = Oracle(si.model, engine.make_blob(si.model, si.key_joint_ids))
    orc2.qpos[:] = d["qpos"][60]; orc2.qvel[:] = d["qvel"][60]; orc2.qacc_warmstart[:] = d["qacc_warmstart"][60]
    orc2.ctrl[:] = ctrl[60]; orc2.step1(); orc2.step(1)
    assert orc2.ncon == int(d["ncon"][60]) and np.array_equal(orc2.qvel, d["qvel"][61])


def test_golden_pathway_with_the_oracle_as_recorder(synthetic_golden):
    _check_oracle(synthetic_golden)


@pytest.mark.gpu
def test_golden_pathway_engine_against_the_oracle_recording(synthetic_golden):
    _check_engine(synthetic_golden)
