/*
 * rp_engine.h — C ABI of the MI355X batched physics engine for RoboPianist.
 *
 * The reference has no FFI seam for physics: tasks/entities talk to a
 * `dm_control.mjcf.Physics` Python object.  Each entry point
 * below names the part of that object protocol it replaces, with the reference
 * call sites that use it.  A maintainer binds this library with ctypes (see
 * INTEGRATION.md); `robopianist_amd/engine.py` is that binding.
 *
 * Conventions
 *   - All arrays are env-major `[n_envs][n]`, caller-owned, and may be device
 *     pointers or host pointers (copied with hipMemcpyDefault on the engine
 *     stream).  Element type is `float` for precision 32 and `double` for 64.
 *   - Return value: 0 = OK, negative = error; message via rp_last_error().
 *   - Physics divergence is not an error; it sets per-env RP_WARN_* bits
 *     (mirrors mj_checkPos/Vel -> dm_control PhysicsError).
 *   - Calls are asynchronous on the engine's HIP stream; rp_get() to host
 *     memory and rp_sync() synchronise (rp_get to device memory does not).
 *     One host thread per engine.
 */
#ifndef RP_ENGINE_H
#define RP_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rp_engine rp_engine;

typedef enum {
  RP_QPOS = 0,          /* [E][nv]    physics.bind(joints).qpos   piano.py:186, hands/base.py:78 */
  RP_QVEL = 1,          /* [E][nv]    physics.bind(joints).qvel   hands/base.py:96 */
  RP_QACC_WARMSTART = 2,/* [E][nv]    mjData.qacc_warmstart (checkpoint/resume) */
  RP_CTRL = 3,          /* [E][nu]    physics.bind(actuators).ctrl shadow_hand.py:387, piano.py:181 */
  RP_QFRC_APPLIED = 4,  /* [E][nv]    physics.bind(joints).qfrc_applied  piano_with_shadow_hands_test.py:235 */
  RP_ACT_FORCE = 5,     /* [E][nu]    `actuatorfrc` sensors       shadow_hand.py:240-245,412 */
  RP_ACT_VELOCITY = 6,  /* [E][nu]    `actuatorvel` sensors       shadow_hand.py:233-238,413 */
  RP_SITE_XPOS = 7,     /* [E][nsite][3] physics.bind(sites).xpos piano_with_shadow_hands.py:309-310,336-337 */
  RP_TIME = 8,          /* [E]        physics.data.time           midi_module.py:70 */
  RP_NCON = 9,          /* [E] int32  physics.data.ncon */
  RP_CONTACT_GEOMS = 10,/* [E][RP_MAX_CONTACTS][2] int32 model geom ids (-1 pad), for
                           collision_utils.has_collision          piano_with_shadow_hands.py:253-257 */
  RP_WARN_FLAGS = 11,   /* [E] int32 */
  RP_SOLVER_ITER = 12,  /* [E] int32  last substep: bits 0-7 Newton iterations, 8-15 rows in the
                           dense (cross-chain) block, 16-23 touched keys */
  RP_CONTACT_DIST = 13, /* [E][RP_MAX_CONTACTS] */
  RP_TREE_OFFSET = 14,  /* [E][ntree][3] per-env root-body translation (hand.shift_pose,
                           piano_with_shadow_hands.py:491-499) */
  RP_ACTIVE = 15,       /* [E] int32, write-only: envs with 0 are skipped by rp_step/rp_forward
                           (rp_set(RP_ACTIVE, NULL) re-enables all).  Needed because dm_env's
                           step-after-LAST is a reset, not a physics step. */
  RP_SENSOR_TORQUE = 16,/* [E][nv]    `torque` sensors at every hand joint's body origin projected on the
                           joint axis = the `joints_torque` observable  shadow_hand.py:209-226, hands/base.py:101-109
                           (key dofs: 0).  Needs rp_set_acc_sensors(e, 1). */
  RP_ENV_COST = 18,     /* [E] int32  shader-clock cycles >> 8 of the env's last solver-stage wave (diagnostic; the
                           predictor of rp_set_cost_ordered_launch) */
  RP_DEBUG_MASS_ROWS = 19, /* [E][rm_rows][rm_cols] (rp_dim; 52 x 10, deep builds 60 x 14) hand-over: row i = M[link i][its ancestors by depth] (tests) */
  RP_DEBUG_HANDOVER_HDR = 20, /* [E][8] int32 hand-over header of the last position stage: contacts, touched keys,
                           dirty-row mask lo / hi, contact Jacobian entries, max entries per contact (diagnostics) */
  RP_SENSOR_TOUCH = 17  /* [E][nsite] `touch` sensors (sum of the normal forces of the contacts whose force ray
                           hits the site's sphere), non-zero for the fingertip sites = the `fingertip_force`
                           observable  shadow_hand.py:248-270,425-432.  Needs rp_set_acc_sensors(e, 1). */
} rp_field;

#define RP_MAX_CONTACTS 64

#define RP_WARN_BADSTATE 1     /* NaN / |q|>1e10 in qpos or qvel */
#define RP_WARN_CONTACT_FULL 2 /* more than 64 simultaneous contacts or 640 contact Jacobian entries (rpk: RpCaps; the
                                  fp32 build: 32 / 240) in one env; the deepest contacts are kept */
#define RP_WARN_HESSIAN 4      /* non-positive pivot in the Newton Hessian */
#define RP_WARN_KEYSLOT_FULL 8 /* more simultaneously touched keys than solver slots */
#define RP_WARN_WORK_FULL 16   /* candidate list overflow: more geom-key candidates than the key list holds */
#define RP_WARN_SPLIT_FULL 64  /* split position stage only: more than 256 candidates / 384 result records in one env and mj_step (the
                                  excess is dropped for that mj_step; the engine's own schedule rule stops using the split stage from the
                                  next step on -- the one-kernel stage never overflows) */
#define RP_WARN_DENSE_FULL 32  /* more than 57 (fp32 build: 53; deep builds: 52 / 47) cross-coupled rows; cross terms dropped */

/* Builds an engine for `n_envs` copies of the model in `model_blob`
 * (robopianist_amd.model.compile.to_blob + engine tables) on HIP device
 * `device_id`.  precision: 32 or 64.  Replaces: mjcf.Physics.from_mjcf_model /
 * composer.Environment construction (suite/__init__.py:87-93). */
int rp_create(const void* model_blob, size_t blob_bytes, int n_envs, int device_id,
              int precision, rp_engine** out);
int rp_destroy(rp_engine* e);

/* physics.reset(): qpos<-qpos0, qvel<-0, ctrl<-0, warmstart<-0, qfrc_applied<-0,
 * time<-0 for envs with mask[e]!=0 (mask==NULL: all).  mask is a host or
 * device uint8 array; a device mask keeps the call fully asynchronous. */
int rp_reset(rp_engine* e, const uint8_t* mask);

/* Generic field write/read (RP_QPOS, RP_QVEL, RP_QACC_WARMSTART, RP_CTRL,
 * RP_QFRC_APPLIED, RP_TREE_OFFSET, RP_TIME are writable). */
int rp_set(rp_engine* e, rp_field f, const void* src);
int rp_get(rp_engine* e, rp_field f, void* dst);

/* `n_substeps` x physics.step() in dm_control legacy order (mj_step2; mj_step1),
 * i.e. the body of composer.Environment.step's substep loop
 * (suite/tasks/base.py:28,31,68-70).  If key_trace != NULL it receives, for
 * every substep, the bitmask of activated keys (Piano._update_key_state,
 * piano.py:178-192) as [E][n_substeps][4] uint32. */
int rp_step(rp_engine* e, int n_substeps, uint32_t* key_trace);

/* Recomputes position/velocity-stage outputs (site_xpos, act_velocity, contacts)
 * for the current state without stepping: physics.forward(). */
int rp_forward(rp_engine* e);

/* One call for what composer.Environment.step does around its substep loop when episodes of a batch end at
 * different times (dm_control composer/environment.py: a step after LAST resets instead of stepping):
 * rp_reset(reset_mask) of the flagged envs, physics.forward() of those (their position / velocity stage: the
 * FIRST observation reads site positions / contacts of the reset state), and rp_step of the envs the RP_ACTIVE mask
 * selects -- the caller clears the flagged envs there (rp_task_prestep writes both masks).  reset_mask: [E] bytes,
 * device memory; NULL = rp_step.  Same results as rp_reset + rp_set(RP_ACTIVE) + rp_forward + rp_set(RP_ACTIVE) +
 * rp_step, with one leading position / velocity stage instead of two. */
int rp_step_masked(rp_engine* e, int n_substeps, uint32_t* key_trace, const uint8_t* reset_mask);

/* Solver iteration caps (defaults: model opt.iterations / opt.ls_iterations). */
int rp_set_solver_limits(rp_engine* e, int max_newton_iter, int max_ls_iter);

/* Newton termination tolerance / line-search tolerance (defaults: model opt.tolerance,
 * opt.ls_tolerance; the fp32 build raises a default tolerance below 1e-6 to 1e-6, the resolution
 * of single precision; values <= 0 leave the current setting). */
int rp_set_solver_tolerance(rp_engine* e, double tolerance, double ls_tolerance);
/* Stopping tolerance of the convex narrow phase (hull / cylinder pairs: Minkowski Portal Refinement, MuJoCo's
 * mjc_Convex; reached from physics.step() like everything else, /root/reference/robopianist/suite/__init__.py:87-93).
 * Default: MuJoCo's opt.mpr_tolerance = 1e-6 for every pair.  `polytope_tolerance` > 0 gives pairs of two polytopes
 * (box / hull on both sides) their own: at 1e-10 the refinement has converged to a face of the Minkowski difference and
 * the result no longer depends on rounding-sized tie-breaks between support vertices -- the setting of step-by-step
 * comparisons against another implementation (the oracle has the same switch); NOT MuJoCo's rule.  Values <= 0 leave
 * `tolerance` as it is / set the polytope tolerance back to `tolerance`.  fp32 engines clamp at 1e-6. */
int rp_set_mpr_tolerance(rp_engine* e, double tolerance, double polytope_tolerance);
/* rp_step starts with a position/velocity stage for the incoming state (the caller may have
 * changed it).  With the lazy mode on, that leading stage is skipped for every env whose stage data
 * is still the one of its current state: computed by the previous rp_step / rp_forward, and no
 * rp_reset / rp_set(RP_QPOS | RP_QVEL | RP_TREE_OFFSET) since.  The engine cannot see writes through
 * rp_field_ptr views: a caller that writes state through a view must call rp_forward (or rp_set)
 * before the next rp_step.  Default: off.  Results are bit-identical either way. */
int rp_set_lazy_position_stage(rp_engine* e, int on);
/* dm_control's `legacy_step` (robopianist/suite/__init__.py:55,91 -> composer.Environment(legacy_step=...) ->
 * mjcf.Physics.step).  on (the default, and dm_control's): every physics.step() is `mj_step2; mj_step1`, i.e. after
 * rp_step the position-dependent outputs (RP_SITE_XPOS, RP_NCON / RP_CONTACT_GEOMS / RP_CONTACT_DIST, RP_ACT_VELOCITY)
 * belong to the NEW state.  off: physics.step() is `mj_step` = `mj_step1; mj_step2` -- the state trajectory is the same,
 * but those outputs are the ones of the state BEFORE the last integration of the call, exactly what MuJoCo's mjData
 * holds after mj_step.  (The engine still runs the stage for the new state -- the next rp_step needs its hand-over --
 * and only keeps its outputs back; the leading stage of the next rp_step publishes them, so the lazy mode is ignored.) */
int rp_set_legacy_step(rp_engine* e, int on);
/* Cost-ordered launch: every stage kernel processes the envs in descending order of what their last
 * solver stage needed (Newton iterations x coupled rows, contact count), re-sorted on the device after
 * every solver stage (longest-processing-time-first).  One wave steps one env and a SIMD runs its
 * envs one after the other, so in a heterogeneous batch (random policies, envs at different episode
 * times) a heavy env that happens to start last is the tail of the whole launch.  Results are
 * bit-identical either way (envs are independent); default: off. */
int rp_set_cost_ordered_launch(rp_engine* e, int on);
/* Capacity classes of the solver stage (on by default for the fp64 engine on scenes with <= 2 forearm dofs per
 * hand; RP_LEAN=0 in the environment turns the default off).  on: envs whose constraint system of the substep
 * fits the light class (rpk::LeanCaps: <= 24 contacts, <= 184 contact Jacobian entries, <= 36 cross-coupled rows,
 * <= 12 touched keys) are stepped by the lean build of the solver stage (two waves per SIMD), the others by the full-capacity
 * build; results do not depend on the class to more than rounding.  No counterpart in the reference (MuJoCo
 * allocates its constraint arrays per step).  on > 1 (tests): as 1, with the light class further restricted to
 * envs with at most `on` contact Jacobian entries, so that small scenes exercise both classes. */
int rp_set_lean_solver(rp_engine* e, int on);

/* Fused substeps (fp64 default builds, i.e. wherever the lean solver stage applies): rp_step runs all its
 * substeps in ONE launch -- a wave keeps its env through n_sub x (solver stage; position stage) -- followed by one
 * small clean-up launch for the envs that left the light capacity class on the way.  on: 0 = never (one launch
 * per stage and substep), 1 = always, 2 = automatic (the default): with rp_set_stream_slices(e, 0) the engine
 * times this schedule against the per-stage ones on a few steps of every 128 and runs the fastest -- fused wins
 * when the batch is at most one round of resident waves (<= 2048 envs) or the envs are alike, two slices win on
 * 4096 envs at different episode times.  Same results to rounding (the same stage code in the same order per
 * env).  rp_get_fused_substeps: 0 = not in use, 1 = the schedule rp_step currently runs, 2 = a candidate the
 * automatic choice currently rejects.  No counterpart in the reference (mj_step is one CPU call per substep). */
int rp_set_fused_substeps(rp_engine* e, int on);
int rp_get_fused_substeps(rp_engine* e);

/* Split position stage (fp64 default-depth builds).  The position / velocity stage of a substep (mj_step1: kinematics,
 * CRB, collision, constraint rows) runs as three launches instead of one -- front part (kinematics, composite inertias,
 * broad phase, fp32 prefilters; one wave per env, 13 KB of LDS / 155 registers: three waves per SIMD), POOLED narrow phase
 * (one wave per 64 candidate pairs of one geom-type pair, whatever envs they belong to: capsule-capsule, capsule-box,
 * box-box, hull pairs through MPR in buckets of equal vertex sets), back part (constraint rows, contact Jacobians,
 * velocity stage; one wave per env).  Bit-identical results (the same routines on the same inputs, contacts collected
 * in the one-kernel stage's order); candidates beyond 256 per env and mj_step are dropped and flagged
 * RP_WARN_WORK_FULL.  on: 0 = never, 1 = in every per-stage schedule, 2 = automatic (the default where the builds
 * exist; RP_SPLIT_POS=0/1/2 in the environment sets the default): with rp_set_stream_slices(e, 0) the schedule "three
 * slices, split position stage, every slice's launches on one stream" is one more candidate of the engine's own
 * timing-based choice (batches of >= 3072 envs).  What it replaces: mj_collision's narrow phase inside physics.step()
 * (robopianist/suite/tasks/base.py:28,31,68-70).  rp_get_split_position_stage: 0 = not in use, 1 = the schedule rp_step
 * currently runs, 2 = a candidate the automatic choice currently rejects. */
int rp_set_split_position_stage(rp_engine* e, int on);
int rp_get_split_position_stage(rp_engine* e);

/* Stream slices (0, 1, 2, 3 or 4; default 1): with n > 1, rp_step runs n slices of the batch as n
 * independent kernel chains (the caller's stream and internal ones, forked / joined with events
 * inside the call), so that the tail of one slice's launch overlaps another slice's next kernel:
 * faster for heterogeneous batches (two slices: +3 %; four: slower again), slower for uniform ones
 * (-6 %).  0 = automatic: the engine times one and two slices on 8 of every 64 steps (its own HIP
 * events, never waited on) and runs the rest with the faster.  Same results in every mode; batches
 * of fewer than 1024 envs and stream captures run as one slice. */
int rp_set_stream_slices(rp_engine* e, int n);
/* Acceleration-stage sensors (mj_sensorAcc: `torque`, `touch`).  When on, the last substep of every
 * rp_step is followed by one extra launch (the sensor stage: position / velocity stage of the state
 * before that substep's Euler step + mj_rnePostConstraint), which fills RP_SENSOR_TORQUE and
 * RP_SENSOR_TOUCH with the values MuJoCo's sensordata holds after the step.  Default: off (the task's
 * default observables and rewards do not read them). */
int rp_set_acc_sensors(rp_engine* e, int on);
int rp_sync(rp_engine* e);
int rp_get_stream(rp_engine* e, void** hip_stream);
/* Makes the engine enqueue on a caller-owned HIP stream (e.g. PyTorch's current stream),
 * so engine kernels and the caller's own kernels are ordered without host syncs. */
int rp_set_stream(rp_engine* e, void* hip_stream);
/* Device address and size of a field's backing array (zero-copy views: the
 * `physics.bind(...)` arrays of the reference become tensors aliasing engine memory).
 * Accesses must be ordered on the engine's stream. */
int rp_field_ptr(rp_engine* e, rp_field f, void** ptr, size_t* bytes);
int rp_n_envs(const rp_engine* e);
int rp_dim(const rp_engine* e, const char* name); /* "nv","nu","nsite","ntree","nkey","nlink","maxdepth","precision"; "rm_rows","rm_cols" = per-env shape of RP_DEBUG_MASS_ROWS */
/* Average device time (ms) of one rp_step launch sequence (1 + 2*n_substeps kernels)
 * since the last call, measured with HIP events on the engine stream; also returns
 * the number of sequences timed. */
int rp_kernel_time(rp_engine* e, double* avg_ms, int* n_launches);
/* Average device time (ms) of ONE launch of the dominant kernel, HIP events on the engine stream: the solver
 * stage of one substep (rp_lean_solver_kernel + the full-capacity rp_stage_kernel<T,1> next to it; the probed
 * substep rotates with the call count), or -- fused schedule -- rp_fused_steps_kernel + its clean-up launch (all
 * substeps of the step).  Steps the automatic schedule choice runs as trials carry no probe; when both
 * schedules were sampled the one with more samples is reported.  Call before rp_kernel_time if both are wanted
 * for the same interval. */
int rp_solver_kernel_time(rp_engine* e, double* avg_ms, int* n_launches);
/* Average number of env-substeps one timed launch covered, over the launches the last rp_solver_kernel_time
 * call reported: the whole batch or one slice of it (rp_set_stream_slices); fused schedule: batch x substeps. */
int rp_solver_kernel_envs(rp_engine* e, double* avg_envs);
/* 1 if the launches the last rp_solver_kernel_time call reported were fused-substeps launches, 0 if they were
 * per-stage solver launches (bench.py names the timed kernel from this, not from a heuristic). */
int rp_solver_kernel_fused(rp_engine* e);
/* Debug aid: per-phase shader-clock counters of env 0 (see rp_kernels.hpp PROF).
 * Reads and clears the counters (out may be NULL), then enables/disables them. */
int rp_profile(rp_engine* e, long long* out, int n, int enable);
const char* rp_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
