"""mjModel dump (`oracle/make_golden.py`: `tests/golden/mujoco/config<N>.npz`) -> `compile.Model` -> engine blob.

This is the piece that lets the REAL Shadow Hand (mujoco_menagerie, as the reference loads it:
/root/reference/robopianist/models/hands/shadow_hand_constants.py:52-53, scripts/install_deps.sh:81-88, with the
additions of models/hands/shadow_hand.py:91-311 already applied by the reference's own builder) replace this
repo's stand-in hand, and that lets the oracle and the engine be run on the very model a MuJoCo trajectory was
recorded with:

    python oracle/make_golden.py --reference /path/to/robopianist --out tests/golden/mujoco   # needs mujoco
    python -m robopianist_amd.tools.mjmodel_to_blob tests/golden/mujoco/config2.npz scene.blob
    python -m pytest tests/test_mujoco_golden.py                                              # oracle / engine vs MuJoCo

`model_from_npz` maps MuJoCo's arrays (MuJoCo field names, `model_` prefix) onto the `Model` the oracle and the
engine-table builder consume; `npz_from_model` writes the same file format from a compiled `Model`, so the
mapping is exercised here -- where MuJoCo cannot be installed -- by a round trip on the stand-in scene
(tests/test_mujoco_golden.py).  What the mapping requires of the model is asserted, not assumed: 1-dof joints
only (hinge / slide), fixed tendons (`wrap_type`), joint / tendon transmissions, fixed gain / affine bias / no
activation dynamics, no equality constraints, no explicit contact pairs, pyramidal cone + Newton + Euler, default
convex-collision tolerance through libccd MPR (not the native GJK / EPA pipeline), no tendon limits / friction /
damping / stiffness, no disable / enable flags beyond refsafe, condim 3 / priority 0 / solmix 1 / no margin on
colliding geoms; capsule / cylinder / box / mesh collision geoms (meshes enter through the vertices of their convex
hull).  `opt.impratio` IS imported (round 6: oracle and engine apply it).  A dump that violates one of these raises
ValueError in `model_from_npz`.
"""
from __future__ import annotations

import argparse
import re
from typing import Dict, Tuple

import numpy as np

from robopianist_amd.model import compile as mcompile
from robopianist_amd.model import spec

MJ_GEOM_MESH = 7
KEY_JOINT_RE = re.compile(r"(^|/)(white|black)_joint_\d+$")   # piano_mjcf.py: key joints

# Model arrays that are MuJoCo arrays of the same name and shape
_SAME = ("qpos0 body_parentid body_pos body_quat body_ipos body_iquat body_mass body_inertia body_jntadr body_jntnum "
         "body_gravcomp body_weldid jnt_type jnt_bodyid jnt_pos jnt_axis jnt_range jnt_stiffness jnt_solref jnt_solimp "
         "jnt_margin qpos_spring dof_bodyid dof_parentid dof_armature dof_damping dof_frictionloss dof_solref dof_solimp "
         "dof_invweight0 dof_M0 body_invweight0 geom_type geom_bodyid geom_pos geom_quat geom_size geom_contype "
         "geom_conaffinity geom_condim geom_friction geom_solref geom_solimp geom_solmix geom_margin geom_gap "
         "geom_priority geom_rbound site_bodyid site_pos tendon_adr tendon_num wrap_objid wrap_prm actuator_trntype "
         "actuator_ctrlrange actuator_forcerange").split()
_INT = ("body_parentid body_jntadr body_jntnum body_weldid jnt_type jnt_bodyid jnt_limited dof_bodyid dof_parentid "
        "geom_type geom_bodyid geom_contype geom_conaffinity geom_condim geom_priority site_bodyid tendon_adr tendon_num "
        "wrap_objid actuator_trntype actuator_trnid actuator_ctrllimited actuator_forcelimited geom_vertadr geom_vertnum").split()


def model_from_npz(src) -> Tuple[mcompile.Model, np.ndarray]:
    """src: path of an .npz written by oracle/make_golden.py (or by `npz_from_model`), or the dict of its arrays.
    Returns (Model, key_joint_ids)."""
    d = dict(np.load(src, allow_pickle=False)) if not isinstance(src, dict) else src
    g = lambda k: np.asarray(d["model_" + k])
    m = mcompile.Model()
    nb, njnt, ngeom = int(g("nbody")), int(g("njnt")), int(g("ngeom"))
    nv, nq, nu = int(g("nv")), int(g("nq")), int(g("nu"))
    if not (nv == nq == njnt):
        raise ValueError(f"the engine handles 1-dof joints only (nq {nq}, nv {nv}, njnt {njnt}): free / ball joints present")
    # Dynamics the engine does not model are REJECTED, never imported silently (a recording of such a model would
    # make the golden tests blame the engine).  Dumps written before these fields were recorded pass unchecked.
    def _reject(key, ok, what):
        if "model_" + key in d and not ok(np.asarray(d["model_" + key])):
            raise ValueError(f"unsupported model: {what} (model_{key} = {np.asarray(d['model_' + key]).ravel()[:8]})")
    _reject("neq", lambda v: int(v) == 0, "equality constraints")
    _reject("npair", lambda v: int(v) == 0, "explicit contact pairs")
    _reject("wrap_type", lambda v: (v == 1).all(), "tendon wrapping other than fixed (joint) tendons")             # mjWRAP_JOINT
    _reject("actuator_gaintype", lambda v: (v == 0).all(), "actuator gain types other than fixed")                 # mjGAIN_FIXED
    _reject("actuator_biastype", lambda v: np.isin(v, (0, 1)).all(), "actuator bias types other than none / affine")  # mjBIAS_NONE / AFFINE
    _reject("actuator_dyntype", lambda v: (v == 0).all(), "actuators with activation dynamics")                    # mjDYN_NONE
    _reject("opt_mpr", lambda v: abs(float(v[0]) - 1e-6) < 1e-12 and int(v[1]) == 50,
            "convex-collision tolerance / iteration cap other than MuJoCo's defaults (1e-6, 50)")
    # (round 6) what the restatement still does not model, checked wherever a dump records it (oracle/make_golden.py
    # records all of these; dumps written before a field existed pass that check unchecked):
    _reject("tendon_limited", lambda v: not np.asarray(v).any(), "tendon limits")
    _reject("tendon_frictionloss", lambda v: (np.asarray(v, float) == 0).all(), "tendon frictionloss")
    _reject("tendon_damping", lambda v: (np.asarray(v, float) == 0).all(), "tendon damping")
    _reject("tendon_stiffness", lambda v: (np.asarray(v, float) == 0).all(), "tendon stiffness")
    # opt.disableflags: only mjDSBL_REFSAFE (bit 1 << 9 [MEM]; recorded separately as model_opt_refsafe by name) may be
    # set -- and mjDSBL_NATIVECCD where it exists, which make_golden sets on purpose (below).  Every other disable bit
    # switches a pipeline stage off that the engine always runs.
    _reject("opt_disableflags_other", lambda v: int(v) == 0,
            "opt.disableflags other than refsafe / nativeccd (a pipeline stage the engine always runs is switched off)")
    # opt.enableflags: override (o_margin / o_solref / o_solimp / o_friction), multiccd, fwdinv ... -- none is modelled
    _reject("opt_enableflags", lambda v: int(v) == 0, "opt.enableflags (override / multiccd / ...: not modelled)")
    # The convex narrow phase here is a restatement of libccd's MPR (mjc_Convex before MuJoCo's native GJK / EPA became the
    # default).  A recording made WITH the native pipeline cannot be held against it at 1e-9: make_golden switches it off
    # (mjDSBL_NATIVECCD) and records what was active.
    _reject("opt_nativeccd", lambda v: int(v) == 0,
            "the recording used MuJoCo's native convex collision (GJK / EPA), not libccd MPR: re-record with "
            "oracle/make_golden.py, which sets mjDSBL_NATIVECCD")
    if "model_opt" in d:
        o_ = np.asarray(d["model_opt"], float)
        if len(o_) >= 10 and (int(o_[6]) != 0 or int(o_[8]) != 2 or int(o_[9]) != 0):
            raise ValueError(f"unsupported options: cone {int(o_[6])} (pyramidal = 0), solver {int(o_[8])} (Newton = 2), "
                             f"integrator {int(o_[9])} (Euler = 0)")
    names = {}
    for kind in ("body", "joint", "geom", "site", "actuator", "tendon"):
        key = "names_" + kind
        names[kind] = [str(s) for s in d[key]] if key in d else []
    m["names"] = names
    for k, v in (("nbody", nb), ("njnt", njnt), ("nv", nv), ("nq", nq), ("ngeom", ngeom), ("nsite", int(g("nsite"))),
                 ("ntendon", int(g("ntendon"))), ("nu", nu)):
        m[k] = v
    for k in _SAME:
        m[k] = np.array(g(k), dtype=np.int32 if k in _INT else np.float64)
    if not np.isin(m.jnt_type, (spec.JNT_SLIDE, spec.JNT_HINGE)).all():
        raise ValueError("joint types other than hinge / slide")
    # Collision geoms: capsule, cylinder (round 6), box and mesh (through its convex hull) are what the narrow phase of
    # the oracle and of the engine handle.  Anything else that can collide -- ellipsoid (4), sphere (2), plane (0), hfield
    # (1), sdf (8) -- is REJECTED here: the oracle's pair loop would skip such a pair silently and the golden tests would
    # then blame the solver for a missing contact.  Visual geoms (contype = conaffinity = 0) never collide and may be of
    # any type.
    collides = (m.geom_contype != 0) | (m.geom_conaffinity != 0)
    bad = collides & ~np.isin(m.geom_type, (spec.GEOM_CAPSULE, spec.GEOM_CYLINDER, spec.GEOM_BOX, spec.GEOM_MESH))
    if bad.any():
        mj_names = {0: "plane", 1: "hfield", 2: "sphere", 4: "ellipsoid", 8: "sdf"}
        gn = names.get("geom", [])
        items = [f"{gn[i] if i < len(gn) and gn[i] else '#%d' % i} ({mj_names.get(int(m.geom_type[i]), int(m.geom_type[i]))})"
                 for i in np.nonzero(bad)[0][:8]]
        raise ValueError("unsupported collision geom types (capsule / cylinder / box / mesh only): " + ", ".join(items)
                         + (" ..." if int(bad.sum()) > 8 else ""))
    # contact parameters the engine's tables fix (model/engine_tables.py asserts the same; here as a clean error)
    ci = np.nonzero(collides)[0]
    if len(ci):
        def _geoms(mask):
            gn = names.get("geom", [])
            return ", ".join(gn[i] if i < len(gn) and gn[i] else "#%d" % i for i in ci[mask][:8])
        for ok, what in (((m.geom_condim[ci] == 3), "condim other than 3"),
                         ((m.geom_priority[ci] == 0), "geom_priority other than 0"),
                         ((m.geom_margin[ci] == 0) & (m.geom_gap[ci] == 0), "non-zero geom margin / gap"),
                         ((m.geom_solmix[ci] == 1), "geom_solmix other than 1")):
            if not ok.all():
                raise ValueError(f"unsupported collision geoms: {what} ({_geoms(~ok)})")
    m["jnt_limited"] = np.asarray(g("jnt_limited"), np.int32).reshape(njnt)
    m["body_inertia"] = m.body_inertia.reshape(nb, 3)
    # kinematic-tree bookkeeping the oracle wants precomputed
    last = np.full(nb, -1, np.int32)
    for b in range(1, nb):
        last[b] = last[m.body_parentid[b]]
        if m.body_jntnum[b] > 0:
            last[b] = m.body_jntadr[b] + m.body_jntnum[b] - 1
    m["body_lastdof"] = last
    tree = np.zeros(nv, np.int32)
    ntree = 0
    for j in range(nv):
        p = m.dof_parentid[j]
        if p < 0:
            tree[j] = ntree; ntree += 1
        else:
            tree[j] = tree[p]
    m["dof_treeid"] = tree; m["ntree"] = ntree
    # meshes collide through the vertices of their convex hull (MuJoCo does the same); the geom's `size` becomes
    # the half extents of the hull's box in the geom frame, as compile_scene stores it
    vertadr = np.full(ngeom, -1, np.int32); vertnum = np.zeros(ngeom, np.int32)
    mesh_vert = np.zeros((0, 3))
    if (m.geom_type == MJ_GEOM_MESH).any():
        if "model_hull_vert" in d:       # synthetic dumps / dumps that already carry per-geom hull vertices
            mesh_vert = np.asarray(d["model_hull_vert"], float).reshape(-1, 3)
            vertadr = np.asarray(d["model_hull_vertadr"], np.int32); vertnum = np.asarray(d["model_hull_vertnum"], np.int32)
        else:
            did = np.asarray(g("geom_dataid"), np.int64)
            mv, madr, mnum = g("mesh_vert").reshape(-1, 3), g("mesh_vertadr"), g("mesh_vertnum")
            gadr = g("mesh_graphadr") if "model_mesh_graphadr" in d else None
            graph = g("mesh_graph") if "model_mesh_graph" in d else None
            out = []
            for gi in np.flatnonzero(m.geom_type == MJ_GEOM_MESH):
                k = int(did[gi])
                v = mv[madr[k]:madr[k] + mnum[k]]
                if gadr is not None and gadr[k] >= 0:
                    # mesh_graph: numvert, numface, vert_edgeadr[numvert], vert_globalid[numvert], ... [MJ: mjModel.mesh_graph]
                    nhv = int(graph[gadr[k]])
                    ids = graph[gadr[k] + 2 + nhv: gadr[k] + 2 + 2 * nhv]
                    v = v[np.asarray(ids, np.int64)]
                vertadr[gi] = sum(len(o) for o in out); vertnum[gi] = len(v)
                out.append(np.asarray(v, float))
            mesh_vert = np.concatenate(out, 0)
        for gi in np.flatnonzero(m.geom_type == MJ_GEOM_MESH):
            v = mesh_vert[vertadr[gi]:vertadr[gi] + vertnum[gi]]
            m.geom_size[gi] = np.abs(v).max(axis=0)
    m["geom_vertadr"] = vertadr; m["geom_vertnum"] = vertnum
    m["mesh_vert"] = mesh_vert; m["nmeshvert"] = len(mesh_vert)
    from robopianist_amd.model import hull as _hull
    _hull.attach_graphs(m)   # (large hulls -- the real hand's forearm / wrist / palm / thumb meshes: the support walk's graph)
    # touch sensors: the radius of the site's sphere where a touch sensor is attached, else 0 [mjSENS_TOUCH = 0]
    touch = np.zeros(m.nsite)
    if "model_site_touch_radius" in d:
        touch = np.asarray(d["model_site_touch_radius"], float)
    elif "model_sensor_type" in d:
        st, so, ss = g("sensor_type"), g("sensor_objid"), g("site_size").reshape(-1, 3)
        for t, o in zip(st, so):
            if int(t) == 0:
                touch[int(o)] = ss[int(o), 0]
    m["site_touch_radius"] = touch
    # actuators: `position` servos (gain kp, bias (0, -kp, -kv)) on a joint or a fixed tendon
    gain, bias, gear = g("actuator_gainprm").reshape(nu, -1), g("actuator_biasprm").reshape(nu, -1), g("actuator_gear").reshape(nu, -1)
    trnid = g("actuator_trnid").reshape(nu, -1)
    if not np.isin(m.actuator_trntype, (spec.TRN_JOINT, spec.TRN_TENDON)).all():
        raise ValueError("actuator transmissions other than joint / tendon")
    m["actuator_trnid"] = np.asarray(trnid[:, 0], np.int32)
    m["actuator_gainprm"] = np.asarray(gain[:, 0], float)
    m["actuator_biasprm"] = np.asarray(bias[:, :3], float)
    m["actuator_gear"] = np.asarray(gear[:, 0], float)
    m["actuator_ctrllimited"] = np.asarray(g("actuator_ctrllimited"), np.int32).reshape(nu)
    m["actuator_forcelimited"] = np.asarray(g("actuator_forcelimited"), np.int32).reshape(nu)
    # options
    o = g("opt")   # timestep tolerance ls_tolerance impratio iterations ls_iterations cone jacobian solver integrator
    m["opt_timestep"] = float(o[0]); m["opt_tolerance"] = float(o[1]); m["opt_ls_tolerance"] = float(o[2])
    m["opt_impratio"] = float(o[3]); m["opt_iterations"] = int(o[4]); m["opt_ls_iterations"] = int(o[5])
    if int(o[6]) != 0 or int(o[8]) != 2 or int(o[9]) != 0:
        raise ValueError(f"engine: pyramidal cone, Newton solver, Euler integrator only (cone {o[6]}, solver {o[8]}, integrator {o[9]})")
    m["opt_refsafe"] = int(d["model_opt_refsafe"]) if "model_opt_refsafe" in d else 1
    m["opt_gravity"] = np.asarray(g("opt_gravity"), float)
    m["stat_meaninertia"] = float(np.asarray(g("stat_meaninertia")).reshape(-1)[0])
    # static pair list from MuJoCo's own filters
    if "model_exclude_pairs" in d:
        excl = np.asarray(d["model_exclude_pairs"], np.int32).reshape(-1, 2)
    else:
        sig = np.asarray(g("exclude_signature"), np.int64).reshape(-1)   # (body1 << 16) + body2
        excl = np.stack([sig >> 16, sig & 0xFFFF], 1).astype(np.int32) if len(sig) else np.zeros((0, 2), np.int32)
        excl = np.sort(excl, axis=1)
    m["exclude_pairs"] = np.asarray(sorted(map(tuple, excl.tolist())), np.int32).reshape(-1, 2)
    mcompile.static_pairs(m)
    key_joint_ids = np.asarray([j for j, n in enumerate(names["joint"]) if KEY_JOINT_RE.search(n)], np.int64)
    return m, key_joint_ids


def npz_from_model(m: mcompile.Model) -> Dict[str, np.ndarray]:
    """The dump oracle/make_golden.py would write for `m`, in MuJoCo's array shapes (the inverse of
    `model_from_npz` up to what MuJoCo itself does not store); used to exercise the importer without MuJoCo."""
    nu = int(m.nu)
    out = {}
    for k in _SAME + ["jnt_limited", "actuator_ctrllimited", "actuator_forcelimited"]:
        out["model_" + k] = np.asarray(m[k])
    for k in ("nq", "nv", "nu", "nbody", "njnt", "ngeom", "nsite", "ntendon"):
        out["model_" + k] = np.asarray(int(m[k]))
    gain = np.zeros((nu, 10)); gain[:, 0] = m.actuator_gainprm
    bias = np.zeros((nu, 10)); bias[:, :3] = m.actuator_biasprm
    gear = np.zeros((nu, 6)); gear[:, 0] = m.actuator_gear
    trnid = np.full((nu, 2), -1, np.int32); trnid[:, 0] = m.actuator_trnid
    out.update(model_actuator_gainprm=gain, model_actuator_biasprm=bias, model_actuator_gear=gear, model_actuator_trnid=trnid)
    out["model_opt"] = np.array([m.opt_timestep, m.opt_tolerance, m.opt_ls_tolerance, m.opt_impratio, m.opt_iterations,
                                 m.opt_ls_iterations, 0, 0, 2, 0], float)
    out["model_opt_refsafe"] = np.asarray(int(m.opt_refsafe))
    out["model_opt_gravity"] = np.asarray(m.opt_gravity, float)
    # what the importer checks and rejects (values of a model the engine supports)
    out["model_opt_mpr"] = np.array([1e-6, 50.0])
    nt = int(m.ntendon)
    out["model_tendon_limited"] = np.zeros(nt, np.int32); out["model_tendon_frictionloss"] = np.zeros(nt)
    out["model_tendon_damping"] = np.zeros(nt); out["model_tendon_stiffness"] = np.zeros(nt)
    out["model_opt_disableflags_other"] = np.asarray(0); out["model_opt_enableflags"] = np.asarray(0)
    out["model_opt_nativeccd"] = np.asarray(0)
    out["model_neq"] = np.asarray(0); out["model_npair"] = np.asarray(0)
    out["model_wrap_type"] = np.ones(len(np.asarray(m.wrap_objid).reshape(-1)), np.int32)
    out["model_actuator_gaintype"] = np.zeros(nu, np.int32); out["model_actuator_biastype"] = np.ones(nu, np.int32)
    out["model_actuator_dyntype"] = np.zeros(nu, np.int32)
    out["model_stat_meaninertia"] = np.array([m.stat_meaninertia])
    ex = np.asarray(m.exclude_pairs, np.int64).reshape(-1, 2)
    out["model_exclude_signature"] = (ex[:, 0] << 16) + ex[:, 1]
    out["model_site_touch_radius"] = np.asarray(m.site_touch_radius, float)
    if int(m.nmeshvert):
        out["model_hull_vert"] = np.asarray(m.mesh_vert, float)
        out["model_hull_vertadr"] = np.asarray(m.geom_vertadr, np.int32); out["model_hull_vertnum"] = np.asarray(m.geom_vertnum, np.int32)
    for kind, lst in m.names.items():
        out["names_" + kind] = np.array([str(s) for s in lst])
    return out


def blob_from_npz(src) -> bytes:
    from robopianist_amd import engine
    m, keys = model_from_npz(src)
    return engine.make_blob(m, keys)


def main() -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("npz", help="tests/golden/mujoco/config<N>.npz (oracle/make_golden.py)")
    ap.add_argument("out", help="blob file for rp_create / the oracle")
    args = ap.parse_args()
    blob = blob_from_npz(args.npz)
    with open(args.out, "wb") as f:
        f.write(blob)
    m, keys = model_from_npz(args.npz)
    print(f"{args.out}: {len(blob)} bytes, nv={m.nv} nu={m.nu} ngeom={m.ngeom} keys={len(keys)}")


if __name__ == "__main__":
    main()
