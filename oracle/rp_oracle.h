/*
 * rp_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A generic, single-environment, fp64 restatement of the physics step that the
 * reference reaches through `composer_utils.Environment(...)`
 * (/root/reference/robopianist/suite/__init__.py:87-93) -> dm_control
 * `Physics.step()` -> MuJoCo `mj_step` (mujoco>=3.1.1, setup.py:39).
 *
 * PARITY UNPINNED: MuJoCo is a third-party dependency that is absent from
 * /root/reference and from this image (no wheel, no source, no network), and
 * the reference's own tests hold no numeric physics vectors.
 * This file restates MuJoCo's *published* computation pipeline (docs chapter
 * "Computation"; stage list in SURVEY.md Appendix B) from memory.  It is
 * anchored on (i) the reference's one physical inequality
 * (piano_with_shadow_hands_test.py:228-242), (ii) analytic known answers
 * (tests/test_oracle_*.py), not on MuJoCo outputs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or load this library.
 */
#ifndef RP_ORACLE_H
#define RP_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rpo_model rpo_model;
typedef struct rpo_data rpo_data;

/* fields for rpo_get_ptr */
enum {
  RPO_QPOS = 0, RPO_QVEL, RPO_QACC, RPO_QACC_WARMSTART, RPO_CTRL, RPO_QFRC_APPLIED,
  RPO_ACTUATOR_FORCE, RPO_ACTUATOR_VELOCITY, RPO_ACTUATOR_LENGTH,
  RPO_XPOS, RPO_XMAT, RPO_GEOM_XPOS, RPO_GEOM_XMAT, RPO_SITE_XPOS,
  RPO_QM /* dense nv*nv */, RPO_QFRC_BIAS, RPO_QFRC_PASSIVE, RPO_QFRC_ACTUATOR,
  RPO_QFRC_SMOOTH, RPO_QACC_SMOOTH, RPO_QFRC_CONSTRAINT,
  RPO_EFC_FORCE, RPO_EFC_AREF, RPO_EFC_D, RPO_EFC_POS, RPO_EFC_J,
  RPO_CONTACT /* ncon * 16: dist,pos[3],frame[9],geom1,geom2,mu */,
  RPO_TIME, RPO_BODY_POS /* model, writable: nbody*3 */,
  RPO_SENSOR_TORQUE /* nv: joint-axis projection of the torque sensor at each joint's body origin */,
  RPO_SENSOR_TOUCH /* nsite: touch sensor reading of the sites with site_touch_radius > 0 */,
  RPO_CFRC_INT /* nbody*6 */, RPO_SUBTREE_COM /* nbody*3 */
};

rpo_model* rpo_model_load(const void* blob, size_t nbytes);
void rpo_model_free(rpo_model* m);
int rpo_model_int(const rpo_model* m, const char* name);

rpo_data* rpo_data_new(const rpo_model* m);
void rpo_data_free(rpo_data* d);

/* qpos<-qpos0, qvel<-0, ctrl<-0, warmstart<-0, time<-0, then a full forward pass
 * (what dm_control's Physics.reset()+forward() leave behind). */
void rpo_reset(const rpo_model* m, rpo_data* d);
/* position+velocity stage, then acceleration stage WITHOUT integrating. */
void rpo_forward(const rpo_model* m, rpo_data* d);
/* One physics.step() in dm_control legacy order: mj_step2 then mj_step1. */
void rpo_step(const rpo_model* m, rpo_data* d);

double* rpo_get_ptr(const rpo_model* m, rpo_data* d, int field);
int rpo_ncon(const rpo_data* d);
int rpo_nefc(const rpo_data* d);
int rpo_solver_iter(const rpo_data* d);
int rpo_warnings(const rpo_data* d);

/* Batched CPU baseline: steps `nenv` independent copies `nstep` times with the
 * given per-env ctrl [nenv][nu] (held constant), OpenMP over envs. Returns
 * seconds of wall time. */
double rpo_bench(const rpo_model* m, int nenv, int nstep, const double* ctrl,
                 int nthreads, double* qpos_out /* [nenv][nq] or NULL */);
/* every env replays ctrl_seq [T][nu]: row (start[e] + s / hold) mod T at mj_step s */
double rpo_bench_seq(const rpo_model* m, int nenv, int nstep, const double* ctrl_seq, int T, int hold,
                     const int* start, int nthreads, double* qpos_out);

/* Test hook: the line search of the Newton solver (PrimalSearch as restated in rp_oracle.c) on a
 * hand-made one-dimensional problem  phi(alpha) = q0 + q1 alpha + q2 alpha^2 + sum_i row_i(jar_i + alpha jv_i)
 * with n rows of type[i] (0 friction-loss: D, floss, R used; 1 / 2 limit / contact: quadratic D x^2 / 2 on
 * x < 0).  Returns alpha; *evals = number of evaluations of phi it used. */
void rpo_step1(const rpo_model* m, rpo_data* d);   /* mj_step1 of the current state only (teacher-forced replays) */
/* bisection knobs of the narrow phase (oracle/rp_oracle.py: set_narrow_phase_variant; process-wide, defaults = the engine's rules) */
void rpo_debug_set_capsule_box(int variant);
void rpo_debug_set_boxbox_max(int n);
void rpo_debug_set_mpr(double tol, int discrete);   /* uniform stopping tolerance (<= 0: MuJoCo's 1e-6); resets the polytope tolerance */
void rpo_debug_set_mpr_poly(double tol_poly);         /* >= 0: polytope pairs (box / hull on both sides) refine to this instead; < 0: off */
void rpo_debug_newton_stats(long* out, int reset);     /* out[0] Newton directions computed, out[1] of them with the Hessian of the direction before (single-env runs only) */
double rpo_debug_line_search(int n, const int* type, const double* jar, const double* jv, const double* D,
                             const double* floss, const double* R, const double quad[3], double gtol,
                             int ls_iterations, int* evals);

#ifdef __cplusplus
}
#endif
#endif
