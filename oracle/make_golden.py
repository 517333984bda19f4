#!/usr/bin/env python
"""Golden trajectories from REAL MuJoCo -- the pin this repo cannot produce by itself.

TEST INFRASTRUCTURE (see oracle/rp_oracle.h).  PARITY IS UNPINNED in this image: `mujoco`,
`dm_control` and the menagerie Shadow Hand are neither vendored nor installable (no network), so
`oracle/rp_oracle.c` is checked against analytic known answers only.  This script closes that gap
the moment somebody runs it on a machine that HAS the reference's dependencies:

    pip install mujoco dm_control note_seq ...   &&   bash scripts/install_deps.sh   (reference)
    python oracle/make_golden.py --reference /path/to/robopianist --out tests/golden/mujoco

It builds the reference's own environments for BASELINE configs 2-4 (suite.load, the notebook
kwargs of SURVEY.md 3.5, legacy step order), steps them with the exact action streams bench.py uses
and writes, per config, `<out>/config<N>.npz`:

    model_*    every mjModel array our compiled Model mirrors (robopianist_amd/model/compile.py
               uses MuJoCo's field names), so tests can run OUR engine and OUR oracle on the REAL
               model instead of the stand-in hand (mesh geoms excluded: primitive fingertips)
    ctrl       [T, nu]     actuator controls actually applied (after the canonical map)
    qpos,qvel,qacc_warmstart  [T*10+1, nv] state after every mj_step (row 0 = reset state)
    ncon,nefc  [T*10]       contact / constraint-row counts per mj_step
    solver_niter [T*10]

`tests/test_mujoco_golden.py` consumes these files: `robopianist_amd/tools/mjmodel_to_blob.py` turns the dumped
mjModel into the blob the oracle and the engine load (the real menagerie hand instead of the stand-in), and the
tests replay the recorded action stream teacher-forced (every mj_step restarted from MuJoCo's state: 1e-9) and
free-running (1e-4 over 1000 mj_steps) against MuJoCo's qpos / qvel / ncon.  The importer itself is exercised
without MuJoCo by a round trip on the stand-in scene.  Here, without `mujoco`, the script prints what is missing
and exits 0 (a no-op by design).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

MODEL_FIELDS = (
    "nq nv nu nbody njnt ngeom nsite ntendon "
    "qpos0 body_parentid body_pos body_quat body_ipos body_iquat body_mass body_inertia body_jntadr "
    "body_jntnum body_gravcomp body_weldid jnt_type jnt_bodyid jnt_pos jnt_axis jnt_range jnt_limited "
    "jnt_stiffness jnt_solref jnt_solimp jnt_margin qpos_spring dof_bodyid dof_jntid dof_parentid "
    "dof_armature dof_damping dof_frictionloss dof_solref dof_solimp dof_invweight0 dof_M0 "
    "body_invweight0 geom_type geom_bodyid geom_pos geom_quat geom_size geom_contype geom_conaffinity "
    "geom_condim geom_friction geom_solref geom_solimp geom_solmix geom_margin geom_gap geom_priority "
    "geom_rbound site_bodyid site_pos site_quat tendon_adr tendon_num wrap_type wrap_objid wrap_prm "
    "actuator_trntype actuator_trnid actuator_gainprm actuator_biasprm actuator_gear "
    "actuator_ctrllimited actuator_ctrlrange actuator_forcelimited actuator_forcerange "
    "exclude_signature geom_dataid mesh_vert mesh_vertadr mesh_vertnum mesh_graph mesh_graphadr "
    "site_size sensor_type sensor_objtype sensor_objid "
    # what the importer (tools/mjmodel_to_blob.py) must be able to REJECT: dynamics the engine does not model
    "neq npair actuator_gaintype actuator_biastype actuator_dyntype jnt_group "
    "tendon_limited tendon_range tendon_frictionloss tendon_damping tendon_stiffness").split()


def dump_model(m) -> dict:
    out = {}
    for f in MODEL_FIELDS:
        if hasattr(m, f):
            out["model_" + f] = np.asarray(getattr(m, f))
    o = m.opt
    out["model_opt"] = np.array([o.timestep, o.tolerance, o.ls_tolerance, o.impratio, o.iterations,
                                 o.ls_iterations, o.cone, o.jacobian, o.solver, o.integrator], float)
    out["model_opt_gravity"] = np.asarray(o.gravity)
    if hasattr(o, "mpr_tolerance"):      # (MuJoCo < 3.2; later versions call them ccd_*)
        out["model_opt_mpr"] = np.array([o.mpr_tolerance, o.mpr_iterations], float)
    elif hasattr(o, "ccd_tolerance"):
        out["model_opt_mpr"] = np.array([o.ccd_tolerance, o.ccd_iterations], float)
    out["model_stat_meaninertia"] = np.array([m.stat.meaninertia])
    import mujoco
    out["model_opt_refsafe"] = np.asarray(0 if (int(o.disableflags) & int(mujoco.mjtDisableBit.mjDSBL_REFSAFE)) else 1)
    # flags the importer rejects: any disable bit other than refsafe / nativeccd, any enable bit (override, multiccd ...)
    allowed = int(mujoco.mjtDisableBit.mjDSBL_REFSAFE)
    native = getattr(mujoco.mjtDisableBit, "mjDSBL_NATIVECCD", None)
    if native is not None:
        allowed |= int(native)
    out["model_opt_disableflags_other"] = np.asarray(int(o.disableflags) & ~allowed)
    out["model_opt_enableflags"] = np.asarray(int(o.enableflags))
    # 1 = the convex pairs of this recording went through MuJoCo's native GJK / EPA (the default of recent versions)
    # instead of libccd MPR, which oracle and engine restate: run_config() switches it off before stepping
    out["model_opt_nativeccd"] = np.asarray(1 if (native is not None and not (int(o.disableflags) & int(native))) else 0)
    out["mujoco_version"] = np.array([mujoco.__version__])
    for kind, n, enum in (("body", m.nbody, mujoco.mjtObj.mjOBJ_BODY), ("joint", m.njnt, mujoco.mjtObj.mjOBJ_JOINT),
                          ("geom", m.ngeom, mujoco.mjtObj.mjOBJ_GEOM), ("site", m.nsite, mujoco.mjtObj.mjOBJ_SITE),
                          ("actuator", m.nu, mujoco.mjtObj.mjOBJ_ACTUATOR)):
        out["names_" + kind] = np.array([mujoco.mj_id2name(m, enum, i) or "" for i in range(n)])
    return out


def action_stream(config: int, spec, n_steps: int, reference_root: str):
    """The streams of bench.py: config 2 = examples/twinkle_twinkle_actions.npy (canonical [-1, 1]
    mapped onto the spec), configs 3/4 = U(spec.min, spec.max), default_rng(12345)."""
    lo, hi = np.asarray(spec.minimum, float), np.asarray(spec.maximum, float)
    if config == 2:
        a = np.load(os.path.join(reference_root, "examples", "twinkle_twinkle_actions.npy")).astype(np.float64)
        return lo + (np.clip(a, -1, 1) + 1.0) * 0.5 * (hi - lo)
    rng = np.random.default_rng(12345)
    return lo + (rng.uniform(-1.0, 1.0, size=(n_steps, lo.shape[0])) + 1.0) * 0.5 * (hi - lo)


def run_config(config: int, out_dir: str, reference_root: str, n_steps: int):
    from robopianist import suite  # the REFERENCE package
    kw = dict(control_timestep=0.05, gravity_compensation=True, primitive_fingertip_collisions=True,
              reduced_action_space=False, n_steps_lookahead=10)
    name = {2: "RoboPianist-debug-TwinkleTwinkleRousseau-v0", 3: "RoboPianist-debug-TwinkleTwinkleRousseau-v0",
            4: "RoboPianist-debug-CMajorScaleTwoHands-v0"}[config]
    if config in (2, 3):
        kw["trim_silence"] = True
    env = suite.load(name, seed=12345, task_kwargs=kw)
    physics = env.physics
    m, d = physics.model.ptr, physics.data.ptr
    import mujoco
    # The convex pairs (hulls, cylinders) must go through libccd MPR, which oracle and engine restate: recent MuJoCo
    # versions default to their native GJK / EPA pipeline, whose results differ at the 1e-6 m level by construction.
    native = getattr(mujoco.mjtDisableBit, "mjDSBL_NATIVECCD", None)
    if native is not None:
        m.opt.disableflags |= int(native)
    acts = action_stream(config, env.action_spec(), n_steps, reference_root)
    n_sub = env.task.physics_steps_per_control_step
    env.reset()
    qpos, qvel, ncon, nefc, nit, ctrl = [d.qpos.copy()], [d.qvel.copy()], [], [], [], []
    warm = [d.qacc_warmstart.copy()]
    import mujoco
    for t in range(min(n_steps, len(acts))):
        # composer.Environment.step, unrolled so that every mj_step is recorded: before_step writes
        # ctrl, then n_sub x (mj_step2; mj_step1) in dm_control's legacy order
        env.task.before_step(physics, acts[t], env.random_state)
        ctrl.append(d.ctrl.copy())
        for _ in range(n_sub):
            mujoco.mj_step2(m, d)
            mujoco.mj_step1(m, d)
            qpos.append(d.qpos.copy()); qvel.append(d.qvel.copy()); warm.append(d.qacc_warmstart.copy())
            ncon.append(d.ncon); nefc.append(d.nefc); nit.append(int(d.solver_niter[0]) if hasattr(d.solver_niter, "__len__") else int(d.solver_niter))
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"config{config}.npz")
    np.savez_compressed(path, ctrl=np.asarray(ctrl), qpos=np.asarray(qpos), qvel=np.asarray(qvel),
                        qacc_warmstart=np.asarray(warm), n_substeps=np.asarray(n_sub),
                        ncon=np.asarray(ncon), nefc=np.asarray(nefc), solver_niter=np.asarray(nit),
                        **dump_model(m))
    print("wrote", path)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", default=os.environ.get("RP_REFERENCE", "/root/reference"))
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "mujoco"))
    ap.add_argument("--steps", type=int, default=100, help="control steps per config (100 = 1000 mj_steps)")
    ap.add_argument("--configs", type=int, nargs="+", default=[2, 3, 4])
    args = ap.parse_args()
    missing = []
    for mod in ("mujoco", "dm_control", "note_seq"):
        try:
            __import__(mod)
        except Exception:
            missing.append(mod)
    if missing:
        print("make_golden.py: nothing written -- not importable here:", ", ".join(missing),
              "(parity stays UNPINNED; run this where the reference's dependencies are installed)")
        return 0
    sys.path.insert(0, args.reference)
    for c in args.configs:
        run_config(c, args.out, args.reference, args.steps)
    return 0


if __name__ == "__main__":
    sys.exit(main())
