/*
 * rp_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE). See rp_oracle.h for scope and
 * for the "PARITY UNPINNED" statement.
 *
 * Generic (any tree of hinge/slide joints), single-env, fp64, dense-J
 * restatement of MuJoCo's mj_step pipeline as the reference exercises it
 * (SURVEY.md Appendix B).  Stage <-> function map ("[MJ: x]" names the MuJoCo
 * routine whose published behaviour is restated from memory):
 *
 *   kinematics()      [MJ: mj_kinematics]          body/geom/site poses
 *   com_pos()         [MJ: mj_comPos]              subtree com, cinert, cdof
 *   crb()             [MJ: mj_crb]                 dense joint-space inertia
 *   factor_ld()       [MJ: mj_factorM]             tree-sparse L'DL
 *   collision()       [MJ: mj_collision]           static pair list + sphere cull
 *   make_constraint() [MJ: mj_makeConstraint,mj_makeImpedance] friction/limit/contact rows
 *   com_vel(), rne()  [MJ: mj_comVel, mj_rne]      bias forces
 *   passive()         [MJ: mj_passive]             spring/damper/gravcomp
 *   actuation()       [MJ: mj_transmission, mj_fwdActuation]
 *   solve_newton()    [MJ: mj_solNewton]           exact-Hessian primal Newton
 *   euler()           [MJ: mj_Euler, eulerdamp]    implicit-damping semi-implicit Euler
 *
 * Reference-side anchors: timestep 0.005 (suite/tasks/base.py:28), key solref
 * (base.py:66), legacy step order step2->step1 (suite/__init__.py:57,92).
 */
#include "rp_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999
#define MAXCON 512
#define JNT_SLIDE 2
#define JNT_HINGE 3
#define GEOM_CAPSULE 3
#define GEOM_CYLINDER 5   /* size = (radius, half height) along the geom's z axis [mjGEOM_CYLINDER] */
#define GEOM_BOX 6
#define GEOM_MESH 7
#define TRN_JOINT 0
#define TRN_TENDON 3

enum { EFC_FRICTION = 0, EFC_LIMIT = 1, EFC_CONTACT = 2 };

/* ------------------------------------------------------------------ blob */
typedef struct {
  char name[40];
  int32_t dtype, ndim;
  int64_t count, offset;
} blob_entry;

struct rpo_model {
  unsigned char* blob;
  size_t nbytes;
  int nentries;
  const blob_entry* entries;
  int nbody, njnt, nv, ngeom, nsite, ntendon, nu, npair;
  double timestep, tolerance, ls_tolerance, meaninertia;
  double impratio;   /* opt.impratio: ratio of frictional to normal constraint impedance [MJ: mj_makeImpedance] */
  int iterations, ls_iterations, refsafe;
  const double* gravity;
  const int32_t *body_parentid, *body_jntadr, *body_jntnum, *body_weldid;
  double* body_pos; /* writable copy (rp_set_body_pos analogue) */
  const double *body_quat, *body_ipos, *body_iquat, *body_mass, *body_inertia,
      *body_gravcomp, *body_invweight0;
  const int32_t *jnt_type, *jnt_bodyid, *jnt_limited;
  const double *jnt_pos, *jnt_axis, *jnt_range, *jnt_stiffness, *qpos_spring, *qpos0,
      *jnt_solref, *jnt_solimp, *jnt_margin;
  const int32_t* dof_parentid;
  const double *dof_armature, *dof_damping, *dof_frictionloss, *dof_invweight0,
      *dof_solref, *dof_solimp;
  const int32_t *geom_type, *geom_bodyid, *geom_condim, *geom_priority;
  const double *geom_pos, *geom_quat, *geom_size, *geom_rbound, *geom_friction,
      *geom_solref, *geom_solimp, *geom_solmix, *geom_margin, *geom_gap;
  const int32_t *geom_vertadr, *geom_vertnum;   /* GEOM_MESH: hull vertices (geom frame) */
  const double* mesh_vert;
  const int32_t *geom_vertgraph, *mesh_graph;   /* large hulls: 1 = walk the vertex graph; rows [degree, neighbours...] (model/hull.py) */
  const int32_t* site_bodyid;
  const double* site_pos;
  const double* site_touch_radius; /* > 0: the site is the zone of a touch sensor (sphere); may be NULL */
  const int32_t *tendon_adr, *tendon_num, *wrap_objid;
  const double* wrap_prm;
  const int32_t *actuator_trntype, *actuator_trnid, *actuator_ctrllimited,
      *actuator_forcelimited;
  const double *actuator_gainprm, *actuator_biasprm, *actuator_gear,
      *actuator_ctrlrange, *actuator_forcerange;
  const int32_t* pair_geom;
  int32_t* body_rootid;
  int has_damping;
};

static const blob_entry* blob_find(const rpo_model* m, const char* name) {
  for (int i = 0; i < m->nentries; i++)
    if (strncmp(m->entries[i].name, name, 40) == 0) return &m->entries[i];
  return NULL;
}
static const void* blob_ptr(const rpo_model* m, const char* name, int dtype) {
  const blob_entry* e = blob_find(m, name);
  if (!e) { fprintf(stderr, "rp_oracle: blob entry '%s' missing\n", name); abort(); }
  if (e->dtype != dtype) { fprintf(stderr, "rp_oracle: blob entry '%s' dtype\n", name); abort(); }
  return m->blob + e->offset;
}
#define BF(name) ((const double*)blob_ptr(m, name, 0))
#define BI(name) ((const int32_t*)blob_ptr(m, name, 1))

int rpo_model_int(const rpo_model* m, const char* name) {
  const blob_entry* e = blob_find(m, name);
  if (!e || e->dtype != 1) return -1;
  return *(const int32_t*)(m->blob + e->offset);
}

rpo_model* rpo_model_load(const void* blob, size_t nbytes) {
  rpo_model* m = (rpo_model*)calloc(1, sizeof(rpo_model));
  m->blob = (unsigned char*)malloc(nbytes);
  memcpy(m->blob, blob, nbytes);
  m->nbytes = nbytes;
  const uint32_t* h = (const uint32_t*)m->blob;
  if (h[0] != 0x52504D42u) { free(m->blob); free(m); return NULL; }
  m->nentries = (int)h[2];
  m->entries = (const blob_entry*)(m->blob + 12);
  m->nbody = *BI("nbody"); m->njnt = *BI("njnt"); m->nv = *BI("nv");
  m->ngeom = *BI("ngeom"); m->nsite = *BI("nsite"); m->ntendon = *BI("ntendon");
  m->nu = *BI("nu"); m->npair = *BI("npair");
  m->timestep = *BF("opt_timestep"); m->tolerance = *BF("opt_tolerance");
  m->ls_tolerance = *BF("opt_ls_tolerance"); m->meaninertia = *BF("stat_meaninertia");
  m->impratio = *BF("opt_impratio");
  m->iterations = *BI("opt_iterations"); m->ls_iterations = *BI("opt_ls_iterations");
  m->refsafe = *BI("opt_refsafe");
  m->gravity = BF("opt_gravity");
  m->body_parentid = BI("body_parentid"); m->body_jntadr = BI("body_jntadr");
  m->body_jntnum = BI("body_jntnum"); m->body_weldid = BI("body_weldid");
  m->body_pos = (double*)malloc(sizeof(double) * 3 * m->nbody);
  memcpy(m->body_pos, BF("body_pos"), sizeof(double) * 3 * m->nbody);
  m->body_quat = BF("body_quat"); m->body_ipos = BF("body_ipos");
  m->body_iquat = BF("body_iquat"); m->body_mass = BF("body_mass");
  m->body_inertia = BF("body_inertia"); m->body_gravcomp = BF("body_gravcomp");
  m->body_invweight0 = BF("body_invweight0");
  m->jnt_type = BI("jnt_type"); m->jnt_bodyid = BI("jnt_bodyid");
  m->jnt_limited = BI("jnt_limited"); m->jnt_pos = BF("jnt_pos");
  m->jnt_axis = BF("jnt_axis"); m->jnt_range = BF("jnt_range");
  m->jnt_stiffness = BF("jnt_stiffness"); m->qpos_spring = BF("qpos_spring");
  m->qpos0 = BF("qpos0"); m->jnt_solref = BF("jnt_solref");
  m->jnt_solimp = BF("jnt_solimp"); m->jnt_margin = BF("jnt_margin");
  m->dof_parentid = BI("dof_parentid"); m->dof_armature = BF("dof_armature");
  m->dof_damping = BF("dof_damping"); m->dof_frictionloss = BF("dof_frictionloss");
  m->dof_invweight0 = BF("dof_invweight0"); m->dof_solref = BF("dof_solref");
  m->dof_solimp = BF("dof_solimp");
  if (m->ngeom) {
    m->geom_type = BI("geom_type"); m->geom_bodyid = BI("geom_bodyid");
    m->geom_condim = BI("geom_condim"); m->geom_priority = BI("geom_priority");
    m->geom_pos = BF("geom_pos"); m->geom_quat = BF("geom_quat");
    m->geom_size = BF("geom_size"); m->geom_rbound = BF("geom_rbound");
    m->geom_friction = BF("geom_friction"); m->geom_solref = BF("geom_solref");
    m->geom_solimp = BF("geom_solimp"); m->geom_solmix = BF("geom_solmix");
    m->geom_margin = BF("geom_margin"); m->geom_gap = BF("geom_gap");
    if (blob_find(m, "geom_vertadr")) {
      m->geom_vertadr = BI("geom_vertadr"); m->geom_vertnum = BI("geom_vertnum");
      m->mesh_vert = blob_find(m, "mesh_vert") ? BF("mesh_vert") : NULL;
      m->geom_vertgraph = blob_find(m, "geom_vertgraph") ? BI("geom_vertgraph") : NULL;
      m->mesh_graph = blob_find(m, "mesh_graph") ? BI("mesh_graph") : NULL;
    }
  }
  if (m->nsite) {
    m->site_bodyid = BI("site_bodyid"); m->site_pos = BF("site_pos");
    m->site_touch_radius = blob_find(m, "site_touch_radius") ? BF("site_touch_radius") : NULL;
  }
  if (m->ntendon) {
    m->tendon_adr = BI("tendon_adr"); m->tendon_num = BI("tendon_num");
    m->wrap_objid = BI("wrap_objid"); m->wrap_prm = BF("wrap_prm");
  }
  if (m->nu) {
    m->actuator_trntype = BI("actuator_trntype"); m->actuator_trnid = BI("actuator_trnid");
    m->actuator_ctrllimited = BI("actuator_ctrllimited");
    m->actuator_forcelimited = BI("actuator_forcelimited");
    m->actuator_gainprm = BF("actuator_gainprm"); m->actuator_biasprm = BF("actuator_biasprm");
    m->actuator_gear = BF("actuator_gear"); m->actuator_ctrlrange = BF("actuator_ctrlrange");
    m->actuator_forcerange = BF("actuator_forcerange");
  }
  if (m->npair) m->pair_geom = BI("pair_geom");
  m->body_rootid = (int32_t*)calloc(m->nbody, sizeof(int32_t));
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b];
    m->body_rootid[b] = (p == 0) ? b : m->body_rootid[p];
  }
  m->has_damping = 0;
  for (int i = 0; i < m->nv; i++) if (m->dof_damping[i] > 0) m->has_damping = 1;
  return m;
}

void rpo_model_free(rpo_model* m) {
  if (!m) return;
  free(m->blob); free(m->body_pos); free(m->body_rootid); free(m);
}

/* ------------------------------------------------------------------ data */
typedef struct {
  double dist, pos[3], frame[9];
  int geom1, geom2;
  double friction[5], solref[2], solimp[5], includemargin;
} contact_t;

struct rpo_data {
  int nv, nu, nbody, ngeom, nsite, njnt, ntendon;
  double time;
  double *qpos, *qvel, *qacc, *qacc_warmstart, *ctrl, *qfrc_applied;
  double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis;
  double *geom_xpos, *geom_xmat, *site_xpos;
  double *subtree_com, *subtree_mass, *cinert, *crbi, *cdof, *cdof_dot, *cvel, *cacc, *cfrc;
  double *qM, *qLD, *qLDe;
  double *ten_length, *ten_velocity, *act_length, *act_velocity, *act_force, *act_moment;
  double *qfrc_passive, *qfrc_bias, *qfrc_actuator, *qfrc_smooth, *qfrc_constraint,
      *qacc_smooth;
  int ncon;
  contact_t* contact;
  double* contact_out;
  int nefc, nefc_max;
  double *efc_J, *efc_pos, *efc_margin, *efc_floss, *efc_diagApprox, *efc_K, *efc_B,
      *efc_imp, *efc_D, *efc_R, *efc_vel, *efc_aref, *efc_force, *efc_jar, *efc_jv;
  int *efc_type, *efc_state;
  /* solver scratch */
  double *Ma, *grad, *Mgrad, *search, *Mv, *H, *tmp;
  int* act_idx;
  int solver_iter, warnings;
  /* acceleration-stage sensors [MJ: mj_rnePostConstraint, mj_sensorAcc] */
  int first_con_row;
  double *cfrc_ext, *cfrc_int, *cacc_post, *sens_torque, *sens_touch;
};

static double* dalloc(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

rpo_data* rpo_data_new(const rpo_model* m) {
  rpo_data* d = (rpo_data*)calloc(1, sizeof(rpo_data));
  int nv = m->nv, nb = m->nbody, nu = m->nu;
  d->nv = nv; d->nu = nu; d->nbody = nb; d->ngeom = m->ngeom; d->nsite = m->nsite;
  d->njnt = m->njnt; d->ntendon = m->ntendon;
  d->qpos = dalloc(nv); d->qvel = dalloc(nv); d->qacc = dalloc(nv);
  d->qacc_warmstart = dalloc(nv); d->ctrl = dalloc(nu); d->qfrc_applied = dalloc(nv);
  d->xpos = dalloc(3 * nb); d->xquat = dalloc(4 * nb); d->xmat = dalloc(9 * nb);
  d->xipos = dalloc(3 * nb); d->ximat = dalloc(9 * nb);
  d->xanchor = dalloc(3 * m->njnt); d->xaxis = dalloc(3 * m->njnt);
  d->geom_xpos = dalloc(3 * m->ngeom); d->geom_xmat = dalloc(9 * m->ngeom);
  d->site_xpos = dalloc(3 * m->nsite);
  d->subtree_com = dalloc(3 * nb); d->subtree_mass = dalloc(nb);
  d->cinert = dalloc(10 * nb); d->crbi = dalloc(10 * nb);
  d->cdof = dalloc(6 * nv); d->cdof_dot = dalloc(6 * nv);
  d->cvel = dalloc(6 * nb); d->cacc = dalloc(6 * nb); d->cfrc = dalloc(6 * nb);
  d->qM = dalloc((size_t)nv * nv); d->qLD = dalloc((size_t)nv * nv);
  d->qLDe = dalloc((size_t)nv * nv);
  d->ten_length = dalloc(m->ntendon); d->ten_velocity = dalloc(m->ntendon);
  d->act_length = dalloc(nu); d->act_velocity = dalloc(nu); d->act_force = dalloc(nu);
  d->act_moment = dalloc((size_t)nu * nv);
  d->qfrc_passive = dalloc(nv); d->qfrc_bias = dalloc(nv); d->qfrc_actuator = dalloc(nv);
  d->qfrc_smooth = dalloc(nv); d->qfrc_constraint = dalloc(nv); d->qacc_smooth = dalloc(nv);
  d->contact = (contact_t*)calloc(MAXCON, sizeof(contact_t));
  d->contact_out = dalloc(MAXCON * 16);
  d->nefc_max = 3 * nv + 4 * MAXCON;
  int ne = d->nefc_max;
  d->efc_J = dalloc((size_t)ne * nv);
  d->efc_pos = dalloc(ne); d->efc_margin = dalloc(ne); d->efc_floss = dalloc(ne);
  d->efc_diagApprox = dalloc(ne); d->efc_K = dalloc(ne); d->efc_B = dalloc(ne);
  d->efc_imp = dalloc(ne); d->efc_D = dalloc(ne); d->efc_R = dalloc(ne);
  d->efc_vel = dalloc(ne); d->efc_aref = dalloc(ne); d->efc_force = dalloc(ne);
  d->efc_jar = dalloc(ne); d->efc_jv = dalloc(ne);
  d->efc_type = (int*)calloc(ne, sizeof(int)); d->efc_state = (int*)calloc(ne, sizeof(int));
  d->Ma = dalloc(nv); d->grad = dalloc(nv); d->Mgrad = dalloc(nv); d->search = dalloc(nv);
  d->Mv = dalloc(nv); d->H = dalloc((size_t)nv * nv); d->tmp = dalloc(nv);
  d->act_idx = (int*)calloc(nv ? nv : 1, sizeof(int));
  d->cfrc_ext = dalloc(6 * nb); d->cfrc_int = dalloc(6 * nb); d->cacc_post = dalloc(6 * nb);
  d->sens_torque = dalloc(nv); d->sens_touch = dalloc(m->nsite);
  return d;
}

void rpo_data_free(rpo_data* d) {
  if (!d) return;
  double** ps[] = {&d->qpos, &d->qvel, &d->qacc, &d->qacc_warmstart, &d->ctrl, &d->qfrc_applied,
    &d->xpos, &d->xquat, &d->xmat, &d->xipos, &d->ximat, &d->xanchor, &d->xaxis, &d->geom_xpos,
    &d->geom_xmat, &d->site_xpos, &d->subtree_com, &d->subtree_mass, &d->cinert, &d->crbi,
    &d->cdof, &d->cdof_dot, &d->cvel, &d->cacc, &d->cfrc, &d->qM, &d->qLD, &d->qLDe,
    &d->ten_length, &d->ten_velocity, &d->act_length, &d->act_velocity, &d->act_force,
    &d->act_moment, &d->qfrc_passive, &d->qfrc_bias, &d->qfrc_actuator, &d->qfrc_smooth,
    &d->qfrc_constraint, &d->qacc_smooth, &d->contact_out, &d->efc_J, &d->efc_pos,
    &d->efc_margin, &d->efc_floss, &d->efc_diagApprox, &d->efc_K, &d->efc_B, &d->efc_imp,
    &d->efc_D, &d->efc_R, &d->efc_vel, &d->efc_aref, &d->efc_force, &d->efc_jar, &d->efc_jv,
    &d->Ma, &d->grad, &d->Mgrad, &d->search, &d->Mv, &d->H, &d->tmp,
    &d->cfrc_ext, &d->cfrc_int, &d->cacc_post, &d->sens_torque, &d->sens_touch};
  for (size_t i = 0; i < sizeof(ps) / sizeof(ps[0]); i++) free(*ps[i]);
  free(d->contact); free(d->efc_type); free(d->efc_state); free(d->act_idx);
  free(d);
}

/* ------------------------------------------------------------ small math */
static inline double dot3(const double* a, const double* b) { return a[0]*b[0]+a[1]*b[1]+a[2]*b[2]; }
static inline void cross3(double* r, const double* a, const double* b) {
  double x = a[1]*b[2]-a[2]*b[1], y = a[2]*b[0]-a[0]*b[2], z = a[0]*b[1]-a[1]*b[0];
  r[0]=x; r[1]=y; r[2]=z;
}
static inline double norm3(const double* a) { return sqrt(dot3(a, a)); }
static void quat_mul(double* r, const double* a, const double* b) {
  double w = a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3];
  double x = a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2];
  double y = a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1];
  double z = a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0];
  r[0]=w; r[1]=x; r[2]=y; r[3]=z;
}
static void quat_norm(double* q) {
  double n = sqrt(q[0]*q[0]+q[1]*q[1]+q[2]*q[2]+q[3]*q[3]);
  for (int i = 0; i < 4; i++) q[i] /= n;
}
static void quat2mat(double* m, const double* q) {
  double w=q[0], x=q[1], y=q[2], z=q[3];
  m[0]=1-2*(y*y+z*z); m[1]=2*(x*y-w*z); m[2]=2*(x*z+w*y);
  m[3]=2*(x*y+w*z); m[4]=1-2*(x*x+z*z); m[5]=2*(y*z-w*x);
  m[6]=2*(x*z-w*y); m[7]=2*(y*z+w*x); m[8]=1-2*(x*x+y*y);
}
static inline void mat_vec(double* r, const double* m, const double* v) { /* r = M v */
  double x = m[0]*v[0]+m[1]*v[1]+m[2]*v[2], y = m[3]*v[0]+m[4]*v[1]+m[5]*v[2],
         z = m[6]*v[0]+m[7]*v[1]+m[8]*v[2];
  r[0]=x; r[1]=y; r[2]=z;
}
static inline void matT_vec(double* r, const double* m, const double* v) { /* r = M^T v */
  double x = m[0]*v[0]+m[3]*v[1]+m[6]*v[2], y = m[1]*v[0]+m[4]*v[1]+m[7]*v[2],
         z = m[2]*v[0]+m[5]*v[1]+m[8]*v[2];
  r[0]=x; r[1]=y; r[2]=z;
}
static void mat_mul(double* r, const double* a, const double* b) {
  double t[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
    t[3*i+j] = a[3*i]*b[j] + a[3*i+1]*b[3+j] + a[3*i+2]*b[6+j];
  memcpy(r, t, sizeof t);
}

/* ------------------------------------------------------------ kinematics */
static void kinematics(const rpo_model* m, rpo_data* d) {
  d->xquat[0] = 1;
  quat2mat(d->xmat, d->xquat);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b];
    double pos[3], quat[4], mat[9], t[3];
    mat_vec(t, d->xmat + 9*p, m->body_pos + 3*b);
    for (int k = 0; k < 3; k++) pos[k] = d->xpos[3*p+k] + t[k];
    quat_mul(quat, d->xquat + 4*p, m->body_quat + 4*b);
    for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
      quat2mat(mat, quat);
      double axis[3], anchor[3];
      mat_vec(axis, mat, m->jnt_axis + 3*j);
      mat_vec(t, mat, m->jnt_pos + 3*j);
      for (int k = 0; k < 3; k++) anchor[k] = pos[k] + t[k];
      double q = d->qpos[j] - m->qpos0[j];
      if (m->jnt_type[j] == JNT_SLIDE) {
        for (int k = 0; k < 3; k++) pos[k] += axis[k] * q;
      } else {
        double s = sin(0.5*q), dq[4] = {cos(0.5*q), m->jnt_axis[3*j]*s, m->jnt_axis[3*j+1]*s,
                                        m->jnt_axis[3*j+2]*s}, nq[4];
        quat_mul(nq, quat, dq);
        memcpy(quat, nq, sizeof nq);
        quat2mat(mat, quat);
        mat_vec(t, mat, m->jnt_pos + 3*j);
        for (int k = 0; k < 3; k++) pos[k] = anchor[k] - t[k];
      }
      memcpy(d->xanchor + 3*j, anchor, sizeof anchor);
      memcpy(d->xaxis + 3*j, axis, sizeof axis);
    }
    quat_norm(quat);
    memcpy(d->xpos + 3*b, pos, sizeof pos);
    memcpy(d->xquat + 4*b, quat, sizeof quat);
    quat2mat(d->xmat + 9*b, quat);
    mat_vec(t, d->xmat + 9*b, m->body_ipos + 3*b);
    for (int k = 0; k < 3; k++) d->xipos[3*b+k] = pos[k] + t[k];
    double imat[9];
    quat2mat(imat, m->body_iquat + 4*b);
    mat_mul(d->ximat + 9*b, d->xmat + 9*b, imat);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    double t[3], gm[9];
    mat_vec(t, d->xmat + 9*b, m->geom_pos + 3*g);
    for (int k = 0; k < 3; k++) d->geom_xpos[3*g+k] = d->xpos[3*b+k] + t[k];
    quat2mat(gm, m->geom_quat + 4*g);
    mat_mul(d->geom_xmat + 9*g, d->xmat + 9*b, gm);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    double t[3];
    mat_vec(t, d->xmat + 9*b, m->site_pos + 3*s);
    for (int k = 0; k < 3; k++) d->site_xpos[3*s+k] = d->xpos[3*b+k] + t[k];
  }
}

/* cinert layout: Ixx Iyy Izz Ixy Ixz Iyz  mdx mdy mdz  m  (about the tree's
 * reference point, world orientation). */
static void com_pos(const rpo_model* m, rpo_data* d) {
  int nb = m->nbody;
  for (int b = 0; b < nb; b++) {
    d->subtree_mass[b] = m->body_mass[b];
    for (int k = 0; k < 3; k++) d->subtree_com[3*b+k] = m->body_mass[b] * d->xipos[3*b+k];
  }
  for (int b = nb - 1; b >= 1; b--) {
    int p = m->body_parentid[b];
    d->subtree_mass[p] += d->subtree_mass[b];
    for (int k = 0; k < 3; k++) d->subtree_com[3*p+k] += d->subtree_com[3*b+k];
  }
  for (int b = 0; b < nb; b++) {
    if (d->subtree_mass[b] < MINVAL) for (int k = 0; k < 3; k++) d->subtree_com[3*b+k] = d->xipos[3*b+k];
    else for (int k = 0; k < 3; k++) d->subtree_com[3*b+k] /= d->subtree_mass[b];
  }
  for (int b = 1; b < nb; b++) {
    const double* c = d->subtree_com + 3*m->body_rootid[b];
    const double* R = d->ximat + 9*b;
    const double* I = m->body_inertia + 3*b;
    double mass = m->body_mass[b], dd[3];
    for (int k = 0; k < 3; k++) dd[k] = d->xipos[3*b+k] - c[k];
    double Iw[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
      Iw[3*i+j] = R[3*i]*I[0]*R[3*j] + R[3*i+1]*I[1]*R[3*j+1] + R[3*i+2]*I[2]*R[3*j+2];
    double d2 = dot3(dd, dd);
    double* ci = d->cinert + 10*b;
    ci[0] = Iw[0] + mass*(d2 - dd[0]*dd[0]);
    ci[1] = Iw[4] + mass*(d2 - dd[1]*dd[1]);
    ci[2] = Iw[8] + mass*(d2 - dd[2]*dd[2]);
    ci[3] = Iw[1] - mass*dd[0]*dd[1];
    ci[4] = Iw[2] - mass*dd[0]*dd[2];
    ci[5] = Iw[5] - mass*dd[1]*dd[2];
    ci[6] = mass*dd[0]; ci[7] = mass*dd[1]; ci[8] = mass*dd[2]; ci[9] = mass;
  }
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j];
    const double* c = d->subtree_com + 3*m->body_rootid[b];
    double* cd = d->cdof + 6*j;
    if (m->jnt_type[j] == JNT_SLIDE) {
      cd[0]=cd[1]=cd[2]=0;
      for (int k = 0; k < 3; k++) cd[3+k] = d->xaxis[3*j+k];
    } else {
      double off[3];
      for (int k = 0; k < 3; k++) { cd[k] = d->xaxis[3*j+k]; off[k] = c[k] - d->xanchor[3*j+k]; }
      cross3(cd + 3, d->xaxis + 3*j, off);
    }
  }
}

/* res = I * v  (spatial inertia times motion vector -> force vector) */
static void mul_inert(double* res, const double* I, const double* v) {
  double t[3];
  res[0] = I[0]*v[0] + I[3]*v[1] + I[4]*v[2];
  res[1] = I[3]*v[0] + I[1]*v[1] + I[5]*v[2];
  res[2] = I[4]*v[0] + I[5]*v[1] + I[2]*v[2];
  cross3(t, I + 6, v + 3);
  res[0] += t[0]; res[1] += t[1]; res[2] += t[2];
  cross3(t, I + 6, v);
  res[3] = I[9]*v[3] - t[0]; res[4] = I[9]*v[4] - t[1]; res[5] = I[9]*v[5] - t[2];
}
static double dot6(const double* a, const double* b) {
  return a[0]*b[0]+a[1]*b[1]+a[2]*b[2]+a[3]*b[3]+a[4]*b[4]+a[5]*b[5];
}

static void crb(const rpo_model* m, rpo_data* d) {
  int nv = m->nv, nb = m->nbody;
  memcpy(d->crbi, d->cinert, sizeof(double) * 10 * nb);
  for (int b = nb - 1; b >= 1; b--) {
    int p = m->body_parentid[b];
    if (p > 0) for (int k = 0; k < 10; k++) d->crbi[10*p+k] += d->crbi[10*b+k];
  }
  memset(d->qM, 0, sizeof(double) * nv * nv);
  for (int i = 0; i < nv; i++) {
    double buf[6];
    mul_inert(buf, d->crbi + 10*m->jnt_bodyid[i], d->cdof + 6*i);
    d->qM[i*nv+i] = dot6(d->cdof + 6*i, buf) + m->dof_armature[i];
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) {
      double v = dot6(d->cdof + 6*j, buf);
      d->qM[i*nv+j] = v; d->qM[j*nv+i] = v;
    }
  }
}

/* Tree-sparse L'DL in dense storage: LD[k][i] (i ancestor of k) = L, LD[k][k] = D. */
static void factor_ld(const rpo_model* m, double* LD, const double* M, const double* diag_add, double h) {
  int nv = m->nv;
  memcpy(LD, M, sizeof(double) * nv * nv);
  if (diag_add) for (int i = 0; i < nv; i++) LD[i*nv+i] += h * diag_add[i];
  for (int k = nv - 1; k >= 0; k--) {
    double Mkk = LD[k*nv+k];
    for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) {
      double tmp = LD[k*nv+i] / Mkk;
      for (int j = i; j >= 0; j = m->dof_parentid[j]) LD[i*nv+j] -= LD[k*nv+j] * tmp;
      LD[k*nv+i] = tmp;
    }
  }
}
static void solve_ld(const rpo_model* m, const double* LD, double* x) {
  int nv = m->nv;
  for (int k = nv - 1; k >= 0; k--)
    for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) x[i] -= LD[k*nv+i] * x[k];
  for (int k = 0; k < nv; k++) x[k] /= LD[k*nv+k];
  for (int k = 0; k < nv; k++)
    for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) x[k] -= LD[k*nv+i] * x[i];
}
static void mul_M(const rpo_model* m, const rpo_data* d, double* res, const double* v) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) {
    double s = d->qM[i*nv+i] * v[i];
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) s += d->qM[i*nv+j] * v[j];
    res[i] = s;
  }
  for (int i = 0; i < nv; i++)
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) res[j] += d->qM[i*nv+j] * v[i];
}

/* ------------------------------------------------------------- collision */
static void make_frame(double* frame) { /* frame[0:3] given & normalised */
  double* x = frame; double* y = frame + 3; double* z = frame + 6;
  if (fabs(x[1]) < 0.5) { y[0]=0; y[1]=1; y[2]=0; } else { y[0]=0; y[1]=0; y[2]=1; }
  double dp = dot3(x, y);
  for (int k = 0; k < 3; k++) y[k] -= dp * x[k];
  double n = norm3(y);
  for (int k = 0; k < 3; k++) y[k] /= n;
  cross3(z, x, y);
}

typedef struct { double dist, pos[3], normal[3]; } rawcon;

/* sphere(c1,r1) vs sphere(c2,r2); normal points 1 -> 2 */
static int sphere_sphere(rawcon* c, const double* c1, double r1, const double* c2, double r2, double margin) {
  double v[3] = {c2[0]-c1[0], c2[1]-c1[1], c2[2]-c1[2]};
  double len = norm3(v), dist = len - r1 - r2;
  if (dist > margin) return 0;
  if (len < MINVAL) { v[0]=1; v[1]=0; v[2]=0; } else { for (int k = 0; k < 3; k++) v[k] /= len; }
  c->dist = dist;
  for (int k = 0; k < 3; k++) { c->normal[k] = v[k]; c->pos[k] = c1[k] + v[k]*(r1 + 0.5*dist); }
  return 1;
}

/* capsule-capsule: closest points of the two axis segments, then sphere-sphere.
 * Parallel axes give up to two contacts at the ends of the overlap interval. */
static int capsule_capsule(rawcon* out, const double* p1, const double* m1, const double* s1,
                           const double* p2, const double* m2, const double* s2, double margin) {
  double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]};
  double r1 = s1[0], l1 = s1[1], r2 = s2[0], l2 = s2[1];
  double dif[3] = {p1[0]-p2[0], p1[1]-p2[1], p1[2]-p2[2]};
  double b = dot3(a1, a2), u = -dot3(a1, dif), v = dot3(a2, dif);
  double det = 1.0 - b*b;
  int n = 0;
  if (det > 1e-10) {
    /* x1,x2 minimise |p1 + a1 x1 - p2 - a2 x2|^2 */
    double x1 = (u + b*v) / det, x2 = (v + b*u) / det;
    if (x1 > l1) { x1 = l1; x2 = v + b*x1; } else if (x1 < -l1) { x1 = -l1; x2 = v + b*x1; }
    if (x2 > l2) { x2 = l2; x1 = u + b*x2; if (x1 > l1) x1 = l1; else if (x1 < -l1) x1 = -l1; }
    else if (x2 < -l2) { x2 = -l2; x1 = u + b*x2; if (x1 > l1) x1 = l1; else if (x1 < -l1) x1 = -l1; }
    double c1[3], c2[3];
    for (int k = 0; k < 3; k++) { c1[k] = p1[k] + a1[k]*x1; c2[k] = p2[k] + a2[k]*x2; }
    n += sphere_sphere(out + n, c1, r1, c2, r2, margin);
  } else {
    /* parallel: segment 2 in capsule-1 axis coordinates spans [mid-l2, mid+l2] */
    double sgn = b >= 0 ? 1.0 : -1.0;
    double mid = u; /* a1.(p2-p1) */
    double lo = fmax(-l1, mid - l2), hi = fmin(l1, mid + l2);
    if (lo <= hi) {
      double xs[2] = {lo, hi};
      int cnt = (hi - lo > 1e-12) ? 2 : 1;
      for (int q = 0; q < cnt; q++) {
        double x1 = xs[q], x2 = sgn * (x1 - mid), c1[3], c2[3];
        for (int k = 0; k < 3; k++) { c1[k] = p1[k] + a1[k]*x1; c2[k] = p2[k] + a2[k]*x2; }
        n += sphere_sphere(out + n, c1, r1, c2, r2, margin);
      }
    } else {
      double x1 = mid > 0 ? l1 : -l1;
      double x2 = sgn * (x1 - mid);
      if (x2 > l2) x2 = l2; else if (x2 < -l2) x2 = -l2;
      double c1[3], c2[3];
      for (int k = 0; k < 3; k++) { c1[k] = p1[k] + a1[k]*x1; c2[k] = p2[k] + a2[k]*x2; }
      n += sphere_sphere(out + n, c1, r1, c2, r2, margin);
    }
  }
  return n;
}

/* derivative (half) of squared exterior distance of c + t a to the box h */
static double seg_box_g(const double* c, const double* a, const double* h, double t) {
  double g = 0;
  for (int k = 0; k < 3; k++) {
    double p = c[k] + t*a[k];
    if (p > h[k]) g += a[k] * (p - h[k]); else if (p < -h[k]) g += a[k] * (p + h[k]);
  }
  return g;
}

/* sphere (local centre p, radius r) vs box (half sizes h), all in box frame.
 * normal points sphere -> box (geom1 = capsule, geom2 = box). */
static int sphere_box_local(rawcon* c, const double* p, double r, const double* h, double margin) {
  double q[3], v[3], d2 = 0;
  for (int k = 0; k < 3; k++) {
    q[k] = p[k] > h[k] ? h[k] : (p[k] < -h[k] ? -h[k] : p[k]);
    v[k] = p[k] - q[k]; d2 += v[k]*v[k];
  }
  double nbs[3], dist;
  if (d2 > 0) {
    double dd = sqrt(d2);
    dist = dd - r;
    for (int k = 0; k < 3; k++) nbs[k] = v[k] / dd;
  } else {
    int ax = 0; double best = -1e300;
    for (int k = 0; k < 3; k++) { double pen = fabs(p[k]) - h[k]; if (pen > best) { best = pen; ax = k; } }
    nbs[0]=nbs[1]=nbs[2]=0; nbs[ax] = p[ax] >= 0 ? 1.0 : -1.0;
    q[ax] = nbs[ax] * h[ax];
    dist = best - r;
  }
  if (dist > margin) return 0;
  c->dist = dist;
  for (int k = 0; k < 3; k++) { c->pos[k] = q[k] + nbs[k]*0.5*dist; c->normal[k] = -nbs[k]; }
  return 1;
}

/* capsule (geom1) vs box (geom2).  Contract of MuJoCo's mjc_CapsuleBox [MEM: engine_collision_box.c;
 * the routine itself is not available here]: at most TWO contacts, both sphere-vs-box tests at
 * points of the capsule's axis -- the first at the axis point closest to the box, the second further
 * along the axis where the capsule still bears on the box.  Restated as: first contact at the exact
 * minimiser t* of the segment-box distance (root of the piecewise-linear derivative), second contact
 * at the segment END that lies deeper in / closer to the box (skipped when it coincides with t*).
 * The way MuJoCo picks its second point in edge / corner configurations is NOT reproduced. */
/* Bisection knob (rpo_debug_set_capsule_box; oracle/bisect_golden.py): which points of the capsule's axis get a
 * sphere-box contact.  0 (default, what the engine computes) = the point closest to the box, then the deeper END;
 * 1 = the closest point only; 2 = the two ends only (no interior point); 3 = closest point + BOTH ends (up to three).
 * The choice among these is what DESIGN section 8 lists as "not reproducible from memory" of mjc_CapsuleBox: a first real
 * recording that disagrees with the default can be replayed under each variant to see which rule MuJoCo follows. */
static int g_capbox_variant = 0;
void rpo_debug_set_capsule_box(int variant) { g_capbox_variant = variant; }
static int capsule_box(rawcon* out, const double* cp, const double* cm, const double* cs,
                       const double* bp, const double* bm, const double* bs, double margin) {
  double r = cs[0], l = cs[1];
  double ax[3] = {cm[2], cm[5], cm[8]}, rel[3] = {cp[0]-bp[0], cp[1]-bp[1], cp[2]-bp[2]};
  double c[3], a[3];
  matT_vec(c, bm, rel);
  matT_vec(a, bm, ax);
  for (int k = 0; k < 3; k++) a[k] *= l;
  /* candidate parameters: ends + breakpoints where a coordinate crosses +-h */
  double ts[8]; int nt = 0;
  ts[nt++] = -1; ts[nt++] = 1;
  for (int k = 0; k < 3; k++) {
    if (fabs(a[k]) > MINVAL) {
      double t1 = (bs[k] - c[k]) / a[k], t2 = (-bs[k] - c[k]) / a[k];
      if (t1 > -1 && t1 < 1) ts[nt++] = t1;
      if (t2 > -1 && t2 < 1) ts[nt++] = t2;
    }
  }
  double tstar;
  double gm = seg_box_g(c, a, bs, -1), gp = seg_box_g(c, a, bs, 1);
  if (gm >= 0) tstar = -1;
  else if (gp <= 0) tstar = 1;
  else {
    double tl = -1, gl = gm, tr = 1, gr = gp;
    for (int i = 2; i < nt; i++) {
      double g = seg_box_g(c, a, bs, ts[i]);
      if (g <= 0 && ts[i] > tl) { tl = ts[i]; gl = g; }
      if (g >= 0 && ts[i] < tr) { tr = ts[i]; gr = g; }
    }
    if (tr <= tl) tstar = tl;
    else if (gr - gl > 0) tstar = tl + (tr - tl) * (-gl) / (gr - gl);
    else tstar = tl;
  }
  /* the deeper end: compare the two ends' sphere-box distances (ties: the -1 end) */
  double dend[2];
  for (int e = 0; e < 2; e++) {
    double te = e ? 1.0 : -1.0, p[3] = {c[0]+te*a[0], c[1]+te*a[1], c[2]+te*a[2]};
    rawcon rc;
    dend[e] = sphere_box_local(&rc, p, r, bs, 1e300) ? rc.dist : 1e300;
  }
  double tend = dend[0] <= dend[1] ? -1.0 : 1.0;
  if (fabs(tend - tstar) < 1e-9) tend = -tend;
  double cand[3] = {tstar, tend, -tend};
  int ncand = 2;
  if (g_capbox_variant == 1) ncand = 1;
  else if (g_capbox_variant == 2) { cand[0] = -1.0; cand[1] = 1.0; }
  else if (g_capbox_variant == 3) ncand = 3;
  int n = 0;
  for (int i = 0; i < ncand; i++) {
    if (i > 0 && fabs(cand[i] - cand[0]) < 1e-9) continue;
    if (i > 1 && fabs(cand[i] - cand[1]) < 1e-9) continue;
    double p[3] = {c[0]+cand[i]*a[0], c[1]+cand[i]*a[1], c[2]+cand[i]*a[2]};
    rawcon rc;
    if (sphere_box_local(&rc, p, r, bs, margin)) {
      double w[3];
      mat_vec(w, bm, rc.pos);
      for (int k = 0; k < 3; k++) out[n].pos[k] = bp[k] + w[k];
      mat_vec(out[n].normal, bm, rc.normal);
      out[n].dist = rc.dist;
      n++;
    }
  }
  return n;
}


/* box (geom1) vs box (geom2).  Contract of MuJoCo's mjc_BoxBox [MEM; source not available]: separating-
 * axis test over the 15 axes; if the least-penetration axis is a face normal, the contact points are
 * the part of the other box's incident face inside the reference face's prism (vertices, reference
 * corners, edge crossings: up to 8 points), all with that face normal; if it is an edge-edge axis,
 * one contact at the closest points of the two edges.  Contact position = midway between the
 * surfaces, dist = -penetration, normal from geom1 to geom2.  A quadrilateral clipped by a rectangle
 * has at most eight corners: all candidates are kept in emission order, up to RPO_BOXBOX_MAX = 8
 * (MuJoCo's maximum for the pair; rounds 1-3 kept the three deepest). */
static int g_boxbox_max = 8;   /* experiment knob (scratch/r4/contact_stats.py) */
void rpo_debug_set_boxbox_max(int n) { g_boxbox_max = n; }
#define RPO_BOXBOX_MAX g_boxbox_max
static void bb_keep(rawcon* out, int* n, const double* pos, const double* nrm, double dist) {
  if (*n >= RPO_BOXBOX_MAX) return;
  out[*n].dist = dist;
  memcpy(out[*n].pos, pos, 3*sizeof(double)); memcpy(out[*n].normal, nrm, 3*sizeof(double));
  (*n)++;
}

static int box_box(rawcon* out, const double* p1, const double* m1, const double* s1,
                   const double* p2, const double* m2, const double* s2, double margin) {
  /* R[i][j] = A_i . B_j, t = centre of B in A's frame */
  double R[3][3], Q[3][3], d[3] = {p2[0]-p1[0], p2[1]-p1[1], p2[2]-p1[2]}, t[3];
  for (int i = 0; i < 3; i++) {
    t[i] = m1[i]*d[0] + m1[3+i]*d[1] + m1[6+i]*d[2];
    for (int j = 0; j < 3; j++) {
      R[i][j] = m1[i]*m2[j] + m1[3+i]*m2[3+j] + m1[6+i]*m2[6+j];
      Q[i][j] = fabs(R[i][j]) + 1e-12;
    }
  }
  double tb[3];
  for (int j = 0; j < 3; j++) tb[j] = t[0]*R[0][j] + t[1]*R[1][j] + t[2]*R[2][j];   /* same in B's frame */
  double best = -1e300; int code = -1; double sgn = 1;
  /* face axes of A (codes 0-2) and of B (3-5): separation = |t.L| - rA - rB */
  for (int i = 0; i < 3; i++) {
    double sep = fabs(t[i]) - (s1[i] + s2[0]*Q[i][0] + s2[1]*Q[i][1] + s2[2]*Q[i][2]);
    if (sep > margin) return 0;
    if (sep > best) { best = sep; code = i; sgn = t[i] >= 0 ? 1 : -1; }
  }
  for (int j = 0; j < 3; j++) {
    double sep = fabs(tb[j]) - (s2[j] + s1[0]*Q[0][j] + s1[1]*Q[1][j] + s1[2]*Q[2][j]);
    if (sep > margin) return 0;
    /* parallel faces tie exactly: geom1's face stays the reference unless geom2's is clearly better */
    if (sep > best + 1e-10) { best = sep; code = 3 + j; sgn = tb[j] >= 0 ? 1 : -1; }
  }
  /* edge axes A_i x B_j (codes 6..14); normalised; a face axis wins unless the edge axis is
   * clearly better (5 % + 1e-9 bias), as is customary */
  double ebest = -1e300; int ecode = -1; double esgn = 1, eaxis[3] = {0, 0, 0};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    int i1 = (i+1)%3, i2 = (i+2)%3, j1 = (j+1)%3, j2 = (j+2)%3;
    /* L = A_i x B_j in A's frame: components along A_i1, A_i2 are -R[i2][j], R[i1][j] */
    double l2 = R[i1][j]*R[i1][j] + R[i2][j]*R[i2][j];
    if (l2 < 1e-12) continue;   /* parallel edges: covered by the face axes */
    double inv = 1.0 / sqrt(l2);
    double tl = (t[i2]*R[i1][j] - t[i1]*R[i2][j]) * inv;
    double ra = (s1[i1]*Q[i2][j] + s1[i2]*Q[i1][j]) * inv;
    double rb = (s2[j1]*Q[i][j2] + s2[j2]*Q[i][j1]) * inv;
    double sep = fabs(tl) - ra - rb;
    if (sep > margin) return 0;
    if (sep > ebest) {
      ebest = sep; ecode = 6 + 3*i + j; esgn = tl >= 0 ? 1 : -1;
      /* world axis = (A_i x B_j) / |.| */
      double Ai[3] = {m1[i], m1[3+i], m1[6+i]}, Bj[3] = {m2[j], m2[3+j], m2[6+j]};
      cross3(eaxis, Ai, Bj);
      for (int k = 0; k < 3; k++) eaxis[k] *= inv;
    }
  }
  int n = 0;
  if (ecode >= 0 && ebest > best + 1e-9 + 0.05*fabs(best)) {
    /* ---- edge-edge: closest points of the edge of A (parallel to A_i) and of B (parallel to B_j)
     * that face each other along the axis */
    int i = (ecode - 6) / 3, j = (ecode - 6) % 3;
    double nrm[3] = {esgn*eaxis[0], esgn*eaxis[1], esgn*eaxis[2]};   /* from A to B */
    double pa[3] = {p1[0], p1[1], p1[2]}, pb[3] = {p2[0], p2[1], p2[2]};
    for (int k = 0; k < 3; k++) {
      if (k != i) {
        double ak[3] = {m1[k], m1[3+k], m1[6+k]};
        double sg = dot3(nrm, ak) >= 0 ? 1 : -1;
        for (int c = 0; c < 3; c++) pa[c] += sg * s1[k] * ak[c];
      }
      if (k != j) {
        double bk[3] = {m2[k], m2[3+k], m2[6+k]};
        double sg = dot3(nrm, bk) >= 0 ? -1 : 1;
        for (int c = 0; c < 3; c++) pb[c] += sg * s2[k] * bk[c];
      }
    }
    double ua[3] = {m1[i], m1[3+i], m1[6+i]}, ub[3] = {m2[j], m2[3+j], m2[6+j]};
    double w[3] = {pb[0]-pa[0], pb[1]-pa[1], pb[2]-pa[2]};
    double uaub = dot3(ua, ub), q1 = dot3(ua, w), q2 = -dot3(ub, w), den = 1 - uaub*uaub;
    double alpha = 0, beta = 0;
    if (den > 1e-12) { alpha = (q1 + uaub*q2) / den; beta = (uaub*q1 + q2) / den; }
    alpha = fmin(s1[i], fmax(-s1[i], alpha)); beta = fmin(s2[j], fmax(-s2[j], beta));
    double pos[3];
    for (int c = 0; c < 3; c++) pos[c] = 0.5 * ((pa[c] + alpha*ua[c]) + (pb[c] + beta*ub[c]));
    bb_keep(out, &n, pos, nrm, ebest);
    return n;
  }
  /* ---- face contact: reference box = owner of the axis */
  int refA = code < 3, ax = code % 3;
  const double *pr = refA ? p1 : p2, *mr = refA ? m1 : m2, *sr = refA ? s1 : s2;
  const double *pi = refA ? p2 : p1, *mi = refA ? m2 : m1, *si = refA ? s2 : s1;
  /* outward normal of the reference face: towards the other box */
  double fs = refA ? sgn : -sgn;               /* sign of the reference axis pointing at the incident box */
  int u = (ax+1)%3, v = (ax+2)%3;
  /* incident box in the reference frame: centre ci, axes E[k] (column k of Rri) */
  double ci[3], E[3][3], dd[3] = {pi[0]-pr[0], pi[1]-pr[1], pi[2]-pr[2]};
  for (int a = 0; a < 3; a++) {
    ci[a] = mr[a]*dd[0] + mr[3+a]*dd[1] + mr[6+a]*dd[2];
    for (int k = 0; k < 3; k++) E[k][a] = mr[a]*mi[k] + mr[3+a]*mi[3+k] + mr[6+a]*mi[6+k];
  }
  /* incident face: the face of the incident box most anti-parallel to the reference normal */
  int ia = 0; double bestd = -1;
  for (int k = 0; k < 3; k++) { double c = fabs(E[k][ax]); if (c > bestd) { bestd = c; ia = k; } }
  double isg = (E[ia][ax] * fs > 0) ? -1 : 1;   /* the face whose outward normal opposes fs * axis */
  int iu = (ia+1)%3, iv = (ia+2)%3;
  double fc[3];
  for (int a = 0; a < 3; a++) fc[a] = ci[a] + isg * si[ia] * E[ia][a];      /* face centre */
  double q[4][3];
  for (int c = 0; c < 4; c++) {
    double su = (c == 0 || c == 3) ? -1 : 1, sv = (c < 2) ? -1 : 1;          /* (-,-) (+,-) (+,+) (-,+) */
    for (int a = 0; a < 3; a++) q[c][a] = fc[a] + su*si[iu]*E[iu][a] + sv*si[iv]*E[iv][a];
  }
  double h = sr[ax], hu = sr[u], hv = sr[v];
  double nrm_ref[3] = {fs*mr[ax], fs*mr[3+ax], fs*mr[6+ax]};                 /* world, reference -> incident */
  double nrm[3];
  for (int k = 0; k < 3; k++) nrm[k] = refA ? nrm_ref[k] : -nrm_ref[k];      /* geom1 -> geom2 */
  /* a candidate at reference-plane coordinates (cu, cv) with signed height z along fs*axis */
#define BB_EMIT(cu, cv, zz) do {                                                          \
    double depth_ = h - (zz);                                                             \
    if (-depth_ <= margin) {                                                              \
      double loc_[3]; loc_[ax] = fs * ((zz) + 0.5*depth_); loc_[u] = (cu); loc_[v] = (cv); \
      double pos_[3];                                                                     \
      for (int k_ = 0; k_ < 3; k_++) pos_[k_] = pr[k_] + mr[3*k_+0]*loc_[0] + mr[3*k_+1]*loc_[1] + mr[3*k_+2]*loc_[2]; \
      bb_keep(out, &n, pos_, nrm, -depth_);                                               \
    } } while (0)
  /* (a) incident vertices inside the reference face's prism */
  for (int c = 0; c < 4; c++)
    if (fabs(q[c][u]) <= hu && fabs(q[c][v]) <= hv) BB_EMIT(q[c][u], q[c][v], fs*q[c][ax]);
  /* incident face as a parallelogram in (u, v): P = f0 + a e1 + b e2, a,b in [-1,1] */
  double f0[2] = {fc[u], fc[v]}, e1[2] = {si[iu]*E[iu][u], si[iu]*E[iu][v]}, e2[2] = {si[iv]*E[iv][u], si[iv]*E[iv][v]};
  double det = e1[0]*e2[1] - e1[1]*e2[0];
  double z0 = fs*fc[ax], z1 = fs*si[iu]*E[iu][ax], z2 = fs*si[iv]*E[iv][ax];
  if (fabs(det) > 1e-14) {
    /* (b) reference corners inside the incident parallelogram */
    for (int c = 0; c < 4; c++) {
      double cu = ((c == 0 || c == 3) ? -hu : hu), cv = (c < 2 ? -hv : hv);
      double ru = cu - f0[0], rv = cv - f0[1];
      double a = (ru*e2[1] - rv*e2[0]) / det, b = (e1[0]*rv - e1[1]*ru) / det;
      if (fabs(a) <= 1 && fabs(b) <= 1) BB_EMIT(cu, cv, z0 + a*z1 + b*z2);
    }
  }
  /* (c) crossings of the incident face's edges with the reference rectangle's edges */
  for (int c = 0; c < 4; c++) {
    const double *qa = q[c], *qb = q[(c+1)%4];
    for (int side = 0; side < 4; side++) {
      int cax = side < 2 ? u : v, oax = side < 2 ? v : u;
      double lim = (side & 1) ? (side < 2 ? hu : hv) : -(side < 2 ? hu : hv);
      double ho = side < 2 ? hv : hu;
      double da = qa[cax] - lim, db = qb[cax] - lim;
      if ((da < 0) == (db < 0) || da == db) continue;
      double tt = da / (da - db);
      double oc = qa[oax] + tt*(qb[oax] - qa[oax]);
      if (fabs(oc) > ho) continue;
      double zz = fs * (qa[ax] + tt*(qb[ax] - qa[ax]));
      if (side < 2) BB_EMIT(lim, oc, zz); else BB_EMIT(oc, lim, zz);
    }
  }
#undef BB_EMIT
  return n;
}

/* ------------------------------------------------------------------------------------------------
 * Convex pairs that involve a hull (GEOM_MESH).  MuJoCo sends these to libccd's Minkowski Portal
 * Refinement [MEM: mjc_Convex -> ccdMPRPenetration; neither source is available here]: ONE contact per
 * pair -- penetration depth, direction and a position from the portal's witness points.  Restated as a
 * plain MPR (XenoCollide): portal discovery from the centre-to-centre ray, refinement until the
 * support point in the portal's direction is within `CCD_TOL` of the portal, then
 * depth = distance of the origin to the portal plane along its normal, normal = that direction (from
 * geom1 to geom2), position = midpoint of the two witness points combined with the barycentric
 * weights of the origin's ray through the portal.  Tolerance / iteration cap: MuJoCo's defaults
 * (1e-6, 50). */
#define CCD_TOL 1e-6
#define CCD_ITER 50
/* The stopping rule is MuJoCo's UNIFORM one by default: every pair refines until the support point is within
 * g_mpr_tol (1e-6) of the portal.  Two experiment knobs (process-wide; oracle/rp_oracle.py: set_mpr_experiment):
 *   * g_mpr_tol_poly >= 0: POLYTOPE pairs (both sides a box or a hull) refine to that tolerance instead.  For two
 *     polytopes the refinement ends on a face of B - A after finitely many steps (reach = 0 to rounding); with the
 *     1e-6 rule it may ALSO end one step earlier, on a portal up to 1e-6 short of that face -- and whether it does
 *     depends on rounding-sized differences of the search direction (which of several tied support vertices comes
 *     first).  Two implementations then return depths up to 1e-6 apart: at mj_step 1058 of the hull replay the HIP
 *     engine stops 1.7e-7 short of this oracle, 1.6 % of that step's velocity change (1579 of 1580 mj_steps agree to
 *     1e-11).  At 1e-10 the result no longer depends on the path: the setting of the tests that compare two
 *     implementations step by step (the engine has the same switch, rp_set_mpr_tolerance).  NOT MuJoCo's rule.
 *   * g_mpr_discrete: polytope pairs stop when the support vertex already is a portal vertex. */
static double g_mpr_tol = CCD_TOL;
static double g_mpr_tol_poly = -1.0;   /* < 0: polytope pairs follow g_mpr_tol (the uniform rule) */
static int g_mpr_discrete = 0;
void rpo_debug_set_mpr(double tol, int discrete) { g_mpr_tol = tol > 0 ? tol : CCD_TOL; g_mpr_tol_poly = -1.0; g_mpr_discrete = discrete; }
void rpo_debug_set_mpr_poly(double tol_poly) { g_mpr_tol_poly = tol_poly; }
#define HULL_GRAPH_ROW 24   /* ints per vertex of mesh_graph: degree + neighbours (model/hull.py: GRAPH_ROW) */
typedef struct { int type; const double *pos, *mat, *size; const double* vert; int nvert; const int32_t* graph; } cgeom;
typedef struct { double v[3], p1[3], p2[3]; int id; } mpoint;   /* point of B - A, its witnesses on A and B; id = (vertex of A, vertex of B) for polytopes */

static int geom_support(const cgeom* g, const double* d, double* out) {   /* d: unit, world; returns the vertex id (polytopes) */
  if (g->type == GEOM_CAPSULE) {
    double ax[3] = {g->mat[2], g->mat[5], g->mat[8]}, sg = dot3(ax, d) >= 0 ? 1.0 : -1.0;
    for (int k = 0; k < 3; k++) out[k] = g->pos[k] + sg*g->size[1]*ax[k] + g->size[0]*d[k];
    return 0;
  } else if (g->type == GEOM_CYLINDER) {
    /* [MJ: mjc_support, mjGEOM_CYLINDER] in the geom frame: the direction's xy part scaled to the radius (nothing when
     * it vanishes), sign(z) * half height along the axis (mju_sign: 0 for 0) */
    double dl[3], r[3]; matT_vec(dl, g->mat, d);
    double t = sqrt(dl[0]*dl[0] + dl[1]*dl[1]);
    if (t > MINVAL) { r[0] = dl[0] / t * g->size[0]; r[1] = dl[1] / t * g->size[0]; } else r[0] = r[1] = 0;
    r[2] = dl[2] > 0 ? g->size[1] : (dl[2] < 0 ? -g->size[1] : 0.0);
    double w[3]; mat_vec(w, g->mat, r);
    for (int k = 0; k < 3; k++) out[k] = g->pos[k] + w[k];
    return 0;
  } else if (g->type == GEOM_BOX) {
    int id = 0;
    for (int k = 0; k < 3; k++) out[k] = g->pos[k];
    for (int a = 0; a < 3; a++) {
      double ax[3] = {g->mat[a], g->mat[3+a], g->mat[6+a]}, sg = dot3(ax, d) >= 0 ? 1.0 : -1.0;
      if (sg > 0) id |= 1 << a;
      for (int k = 0; k < 3; k++) out[k] += sg*g->size[a]*ax[k];
    }
    return id;
  } else {
    double dl[3]; matT_vec(dl, g->mat, d);
    int best = 0; double bv = -1e300;
    if (g->graph) {
      /* large hull: walk its vertex graph from vertex 0 -- to the neighbour with the largest dot product (first
       * one on ties) while that is strictly larger than the current value [MJ: hill climbing over mesh_graph] */
      bv = dot3(dl, g->vert);
      for (;;) {
        const int32_t* row = g->graph + (size_t)HULL_GRAPH_ROW * best;
        int next = -1; double nv_ = bv;
        for (int j = 0; j < row[0]; j++) { double v = dot3(dl, g->vert + 3*row[1+j]); if (v > nv_) { nv_ = v; next = row[1+j]; } }
        if (next < 0) break;
        best = next; bv = nv_;
      }
    } else
    for (int i = 0; i < g->nvert; i++) { double v = dot3(dl, g->vert + 3*i); if (v > bv) { bv = v; best = i; } }
    double w[3]; mat_vec(w, g->mat, g->vert + 3*best);
    for (int k = 0; k < 3; k++) out[k] = g->pos[k] + w[k];
    return best;
  }
}
static void mpr_support(const cgeom* A, const cgeom* B, const double* d, mpoint* o) {
  double nd[3] = {-d[0], -d[1], -d[2]};
  int ia = geom_support(A, nd, o->p1), ib = geom_support(B, d, o->p2);
  o->id = ia | (ib << 16);
  for (int k = 0; k < 3; k++) o->v[k] = o->p2[k] - o->p1[k];
}
static int normalize3(double* v) { double n = norm3(v); if (n < 1e-14) return 0; v[0] /= n; v[1] /= n; v[2] /= n; return 1; }

/* returns 1 and fills (dist <= 0, pos, normal) if A and B overlap */
static int mpr_penetration(const cgeom* A, const cgeom* B, rawcon* out) {
  mpoint v0, v1, v2, v3, v4;
  double dir[3], t1[3], t2[3];
  for (int k = 0; k < 3; k++) { v0.p1[k] = A->pos[k]; v0.p2[k] = B->pos[k]; v0.v[k] = B->pos[k] - A->pos[k]; }
  if (norm3(v0.v) < 1e-10) v0.v[0] = 1e-5;
  /* ---- portal discovery */
  for (int k = 0; k < 3; k++) dir[k] = -v0.v[k];
  normalize3(dir);
  mpr_support(A, B, dir, &v1);
  if (dot3(v1.v, dir) <= 0) return 0;
  cross3(dir, v0.v, v1.v);
  if (!normalize3(dir)) {
    /* the origin lies on the ray v0 -> v1: penetration along that ray */
    double n[3] = {v1.v[0] - v0.v[0], v1.v[1] - v0.v[1], v1.v[2] - v0.v[2]};
    normalize3(n);
    out->dist = -dot3(v1.v, n);
    for (int k = 0; k < 3; k++) { out->normal[k] = -n[k]; out->pos[k] = 0.5*(v1.p1[k] + v1.p2[k]); }
    return out->dist <= 0;
  }
  mpr_support(A, B, dir, &v2);
  if (dot3(v2.v, dir) <= 0) return 0;
  for (int k = 0; k < 3; k++) { t1[k] = v1.v[k] - v0.v[k]; t2[k] = v2.v[k] - v0.v[k]; }
  cross3(dir, t1, t2); normalize3(dir);
  if (dot3(dir, v0.v) > 0) { mpoint tmp = v1; v1 = v2; v2 = tmp; for (int k = 0; k < 3; k++) dir[k] = -dir[k]; }
  for (int it = 0; ; it++) {
    if (it > CCD_ITER) return 0;
    mpr_support(A, B, dir, &v3);
    if (dot3(v3.v, dir) <= 0) return 0;
    cross3(t1, v1.v, v3.v);
    if (dot3(t1, v0.v) < 0) {           /* origin outside (v1, v0, v3): replace v2 */
      v2 = v3;
      for (int k = 0; k < 3; k++) { t1[k] = v1.v[k] - v0.v[k]; t2[k] = v3.v[k] - v0.v[k]; }
      cross3(dir, t1, t2); normalize3(dir);
      continue;
    }
    cross3(t1, v3.v, v2.v);
    if (dot3(t1, v0.v) < 0) {           /* origin outside (v3, v0, v2): replace v1 */
      v1 = v3;
      for (int k = 0; k < 3; k++) { t1[k] = v3.v[k] - v0.v[k]; t2[k] = v2.v[k] - v0.v[k]; }
      cross3(dir, t1, t2); normalize3(dir);
      continue;
    }
    break;
  }
  /* ---- portal refinement */
  int hit = 0;
  for (int it = 0; it <= CCD_ITER; it++) {
    for (int k = 0; k < 3; k++) { t1[k] = v2.v[k] - v1.v[k]; t2[k] = v3.v[k] - v1.v[k]; }
    cross3(dir, t1, t2);
    if (!normalize3(dir)) return 0;
    if (dot3(dir, v1.v) >= 0) hit = 1;   /* the origin is inside the portal */
    mpr_support(A, B, dir, &v4);
    double reach = dot3(v4.v, dir) - dot3(v1.v, dir);
    if (!hit && dot3(v4.v, dir) < 0) return 0;             /* the origin lies beyond the support plane */
    const int poly = (A->type == GEOM_BOX || A->type == GEOM_MESH) && (B->type == GEOM_BOX || B->type == GEOM_MESH);
    int stop = reach <= ((poly && g_mpr_tol_poly >= 0) ? g_mpr_tol_poly : g_mpr_tol);
    if (g_mpr_discrete && poly)
      stop = v4.id == v1.id || v4.id == v2.id || v4.id == v3.id || reach <= 1e-10;
    if (stop || it == CCD_ITER) {
      if (!hit) return 0;
      /* penetration: depth along the portal normal, witnesses weighted by the origin's ray */
      double depth = dot3(v1.v, dir);
      double b[4], c[3];
      cross3(c, v1.v, v2.v); b[0] = dot3(c, v3.v);
      cross3(c, v3.v, v2.v); b[1] = dot3(c, v0.v);
      cross3(c, v0.v, v1.v); b[2] = dot3(c, v3.v);
      cross3(c, v2.v, v1.v); b[3] = dot3(c, v0.v);
      double sum = b[0] + b[1] + b[2] + b[3];
      if (sum <= 0) {
        b[0] = 0;
        cross3(c, v2.v, v3.v); b[1] = dot3(c, dir);
        cross3(c, v3.v, v1.v); b[2] = dot3(c, dir);
        cross3(c, v1.v, v2.v); b[3] = dot3(c, dir);
        sum = b[1] + b[2] + b[3];
      }
      const mpoint* P[4] = {&v0, &v1, &v2, &v3};
      double inv = 1.0 / sum;
      for (int k = 0; k < 3; k++) {
        double acc = 0;
        for (int i = 0; i < 4; i++) acc += b[i] * 0.5 * (P[i]->p1[k] + P[i]->p2[k]);
        out->pos[k] = acc * inv; out->normal[k] = -dir[k];   /* (the portal faces away from B - A's centre: geom1 -> geom2 is -dir) */
      }
      out->dist = -depth;
      return 1;
    }
    /* expand the portal with v4 */
    cross3(t1, v4.v, v0.v);
    if (dot3(v1.v, t1) > 0) { if (dot3(v2.v, t1) > 0) v1 = v4; else v3 = v4; }
    else { if (dot3(v3.v, t1) > 0) v2 = v4; else v1 = v4; }
  }
  return 0;
}

static void collision(const rpo_model* m, rpo_data* d) {
  d->ncon = 0;
  for (int ip = 0; ip < m->npair; ip++) {
    int g1 = m->pair_geom[2*ip], g2 = m->pair_geom[2*ip+1];
    double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
    double gap = fmax(m->geom_gap[g1], m->geom_gap[g2]);
    const double *p1 = d->geom_xpos + 3*g1, *p2 = d->geom_xpos + 3*g2;
    double dv[3] = {p1[0]-p2[0], p1[1]-p2[1], p1[2]-p2[2]};
    double bound = m->geom_rbound[g1] + m->geom_rbound[g2] + margin;
    if (dot3(dv, dv) > bound*bound) continue;
    rawcon rc[8]; int n = 0;
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    if (t1 == GEOM_CAPSULE && t2 == GEOM_CAPSULE)
      n = capsule_capsule(rc, p1, d->geom_xmat + 9*g1, m->geom_size + 3*g1, p2,
                          d->geom_xmat + 9*g2, m->geom_size + 3*g2, margin);
    else if (t1 == GEOM_CAPSULE && t2 == GEOM_BOX)
      n = capsule_box(rc, p1, d->geom_xmat + 9*g1, m->geom_size + 3*g1, p2,
                      d->geom_xmat + 9*g2, m->geom_size + 3*g2, margin);
    else if (t1 == GEOM_BOX && t2 == GEOM_BOX)
      n = box_box(rc, p1, d->geom_xmat + 9*g1, m->geom_size + 3*g1, p2,
                  d->geom_xmat + 9*g2, m->geom_size + 3*g2, margin);
    else if ((t2 == GEOM_MESH && m->mesh_vert) || t1 == GEOM_CYLINDER || t2 == GEOM_CYLINDER) {
      /* [MJ: mjCOLLISIONFUNC] every pair with a hull or a cylinder goes to mjc_Convex, in geom-type order
       * (capsule 3 < cylinder 5 < box 6 < mesh 7): (capsule, cylinder), (cylinder, cylinder / box / mesh) */
#define HULL_GRAPH_OF(g_) ((m->geom_vertgraph && m->mesh_graph && m->geom_vertgraph[g_]) ? m->mesh_graph + (size_t)HULL_GRAPH_ROW * m->geom_vertadr[g_] : NULL)
      cgeom A = {t1, p1, d->geom_xmat + 9*g1, m->geom_size + 3*g1,
                 t1 == GEOM_MESH ? m->mesh_vert + 3*m->geom_vertadr[g1] : NULL, t1 == GEOM_MESH ? m->geom_vertnum[g1] : 0,
                 t1 == GEOM_MESH ? HULL_GRAPH_OF(g1) : NULL};
      cgeom B = {t2, p2, d->geom_xmat + 9*g2, m->geom_size + 3*g2,
                 t2 == GEOM_MESH ? m->mesh_vert + 3*m->geom_vertadr[g2] : NULL, t2 == GEOM_MESH ? m->geom_vertnum[g2] : 0,
                 t2 == GEOM_MESH ? HULL_GRAPH_OF(g2) : NULL};
#undef HULL_GRAPH_OF
      n = mpr_penetration(&A, &B, rc);
    }
    else continue;
    for (int i = 0; i < n; i++) {
      if (d->ncon >= MAXCON) { d->warnings |= 2; return; }
      contact_t* c = d->contact + d->ncon++;
      c->dist = rc[i].dist;
      memcpy(c->pos, rc[i].pos, sizeof c->pos);
      memcpy(c->frame, rc[i].normal, 3 * sizeof(double));
      make_frame(c->frame);
      c->geom1 = g1; c->geom2 = g2;
      c->includemargin = margin - gap;
      /* [MJ: mj_contactParam] equal priority: solmix-weighted solref/solimp, max friction */
      double mix;
      double s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2];
      if (m->geom_priority[g1] != m->geom_priority[g2]) mix = m->geom_priority[g1] > m->geom_priority[g2] ? 1 : 0;
      else if (s1 >= MINVAL && s2 >= MINVAL) mix = s1 / (s1 + s2);
      else if (s1 < MINVAL && s2 < MINVAL) mix = 0.5;
      else mix = s1 < MINVAL ? 0.0 : 1.0;
      for (int k = 0; k < 2; k++) c->solref[k] = mix*m->geom_solref[2*g1+k] + (1-mix)*m->geom_solref[2*g2+k];
      for (int k = 0; k < 5; k++) c->solimp[k] = mix*m->geom_solimp[5*g1+k] + (1-mix)*m->geom_solimp[5*g2+k];
      double f[3];
      for (int k = 0; k < 3; k++) f[k] = fmax(m->geom_friction[3*g1+k], m->geom_friction[3*g2+k]);
      c->friction[0] = c->friction[1] = f[0]; c->friction[2] = f[1];
      c->friction[3] = c->friction[4] = f[2];
    }
  }
}

/* ------------------------------------------------------------ constraint */
static void get_impedance(const double* solimp, double pos, double margin, double* imp) {
  double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  dmin = fmin(MAXIMP, fmax(MINIMP, dmin));
  dmax = fmin(MAXIMP, fmax(MINIMP, dmax));
  width = fmax(MINVAL, width);
  mid = fmin(MAXIMP, fmax(MINIMP, mid));
  power = fmax(1, power);
  if (dmin == dmax || width <= MINVAL) { *imp = 0.5*(dmin + dmax); return; }
  double x = fabs(pos - margin) / width;
  if (x >= 1) { *imp = dmax; return; }
  if (x == 0) { *imp = dmin; return; }
  double y;
  if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  *imp = dmin + y * (dmax - dmin);
}

static void row_kbip(const rpo_model* m, rpo_data* d, int i, const double* solref_in,
                     const double* solimp, double pos, double margin) {
  double solref[2] = {solref_in[0], solref_in[1]};
  if (m->refsafe && solref[0] > 0) solref[0] = fmax(solref[0], 2 * m->timestep);
  double imp;
  get_impedance(solimp, pos, margin, &imp);
  double dmax = fmin(MAXIMP, fmax(MINIMP, solimp[1]));
  d->efc_K[i] = 1.0 / fmax(MINVAL, dmax*dmax*solref[0]*solref[0]*solref[1]*solref[1]);
  d->efc_B[i] = 2.0 / fmax(MINVAL, dmax*solref[0]);
  d->efc_imp[i] = imp;
}

/* translational Jacobian of world point `p` on body `b`, added with sign into row[nv][3] */
static void jac_point(const rpo_model* m, const rpo_data* d, double* jac3, int b, const double* p, double sign) {
  int nv = m->nv;
  while (b > 0) {
    for (int j = m->body_jntadr[b] + m->body_jntnum[b] - 1; j >= m->body_jntadr[b]; j--) {
      double col[3];
      if (m->jnt_type[j] == JNT_SLIDE) memcpy(col, d->xaxis + 3*j, sizeof col);
      else {
        double r[3] = {p[0]-d->xanchor[3*j], p[1]-d->xanchor[3*j+1], p[2]-d->xanchor[3*j+2]};
        cross3(col, d->xaxis + 3*j, r);
      }
      for (int k = 0; k < 3; k++) jac3[k*nv + j] += sign * col[k];
    }
    b = m->body_parentid[b];
  }
}

static void make_constraint(const rpo_model* m, rpo_data* d) {
  int nv = m->nv, ne = 0;
  /* friction loss rows */
  for (int i = 0; i < nv; i++) {
    if (m->dof_frictionloss[i] <= 0) continue;
    memset(d->efc_J + (size_t)ne*nv, 0, sizeof(double)*nv);
    d->efc_J[(size_t)ne*nv + i] = 1;
    d->efc_type[ne] = EFC_FRICTION; d->efc_pos[ne] = 0; d->efc_margin[ne] = 0;
    d->efc_floss[ne] = m->dof_frictionloss[i];
    d->efc_diagApprox[ne] = m->dof_invweight0[i];
    row_kbip(m, d, ne, m->dof_solref + 2*i, m->dof_solimp + 5*i, 0, 0);
    ne++;
  }
  /* joint limits */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j]) continue;
    double margin = m->jnt_margin[j];
    for (int side = 0; side < 2; side++) {
      double dist = side == 0 ? d->qpos[j] - m->jnt_range[2*j] : m->jnt_range[2*j+1] - d->qpos[j];
      if (dist >= margin) continue;
      memset(d->efc_J + (size_t)ne*nv, 0, sizeof(double)*nv);
      d->efc_J[(size_t)ne*nv + j] = side == 0 ? 1 : -1;
      d->efc_type[ne] = EFC_LIMIT; d->efc_pos[ne] = dist; d->efc_margin[ne] = margin;
      d->efc_floss[ne] = 0;
      d->efc_diagApprox[ne] = m->dof_invweight0[j];
      row_kbip(m, d, ne, m->jnt_solref + 2*j, m->jnt_solimp + 5*j, dist, margin);
      ne++;
    }
  }
  /* pyramidal contacts, condim 3 */
  double* jd = (double*)malloc(sizeof(double) * 3 * nv);
  int first_con_row = ne;
  for (int ic = 0; ic < d->ncon; ic++) {
    contact_t* c = d->contact + ic;
    int b1 = m->geom_bodyid[c->geom1], b2 = m->geom_bodyid[c->geom2];
    memset(jd, 0, sizeof(double)*3*nv);
    jac_point(m, d, jd, b2, c->pos, 1.0);
    jac_point(m, d, jd, b1, c->pos, -1.0);
    double tran = m->body_invweight0[2*b1] + m->body_invweight0[2*b2];
    for (int r = 0; r < 4; r++) {
      double mu = c->friction[r/2];
      const double* n = c->frame; const double* t = c->frame + 3*(1 + r/2);
      double sg = (r & 1) ? -1.0 : 1.0;
      double* row = d->efc_J + (size_t)ne*nv;
      for (int j = 0; j < nv; j++) {
        double jn = n[0]*jd[j] + n[1]*jd[nv+j] + n[2]*jd[2*nv+j];
        double jt = t[0]*jd[j] + t[1]*jd[nv+j] + t[2]*jd[2*nv+j];
        row[j] = jn + sg*mu*jt;
      }
      d->efc_type[ne] = EFC_CONTACT; d->efc_pos[ne] = c->dist; d->efc_margin[ne] = c->includemargin;
      d->efc_floss[ne] = 0;
      d->efc_diagApprox[ne] = tran + mu*mu*tran;
      row_kbip(m, d, ne, c->solref, c->solimp, c->dist, c->includemargin);
      ne++;
    }
  }
  free(jd);
  d->first_con_row = first_con_row;
  d->nefc = ne;
  /* R, D  [MJ: mj_makeImpedance]; pyramidal rows share Rpy = 2 mu^2 R(first row) */
  for (int i = 0; i < ne; i++) {
    d->efc_R[i] = fmax(MINVAL, (1 - d->efc_imp[i]) * d->efc_diagApprox[i] / d->efc_imp[i]);
  }
  /* [MJ: mj_makeImpedance, pyramidal cone] the friction dimensions are regularised by R / impratio: the contact's
   * regularised friction coefficient is mu = friction[0] * sqrt(1 / impratio), and every pyramid edge gets
   * Rpy = 2 mu^2 R (the Jacobian rows above keep friction[] itself: impratio > 1 makes slip "harder" without
   * widening the cone).  The reference's hand XML sets impratio = 10; mjcf.from_path carries the
   * option into the scene (robopianist/models/hands/shadow_hand.py:122). */
  const double mu_scale = sqrt(1.0 / fmax(MINVAL, m->impratio));
  for (int ic = 0; ic < d->ncon; ic++) {
    int base = first_con_row + 4*ic;
    double mu = d->contact[ic].friction[0] * mu_scale;
    double Rpy = 2 * mu * mu * d->efc_R[base];
    for (int r = 0; r < 4; r++) d->efc_R[base + r] = Rpy;
  }
  for (int i = 0; i < ne; i++) d->efc_D[i] = 1.0 / d->efc_R[i];
}

/* ------------------------------------------------- velocity-stage pieces */
static void cross_motion(double* r, const double* v, const double* s) {
  double t1[3], t2[3];
  cross3(r, v, s);
  cross3(t1, v, s + 3); cross3(t2, v + 3, s);
  r[3] = t1[0]+t2[0]; r[4] = t1[1]+t2[1]; r[5] = t1[2]+t2[2];
}
static void cross_force(double* r, const double* v, const double* f) {
  double t1[3], t2[3];
  cross3(t1, v, f); cross3(t2, v + 3, f + 3);
  r[0] = t1[0]+t2[0]; r[1] = t1[1]+t2[1]; r[2] = t1[2]+t2[2];
  cross3(r + 3, v, f + 3);
}

static void com_vel(const rpo_model* m, rpo_data* d) {
  memset(d->cvel, 0, sizeof(double) * 6);
  for (int b = 1; b < m->nbody; b++) {
    double cv[6];
    memcpy(cv, d->cvel + 6*m->body_parentid[b], sizeof cv);
    for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
      cross_motion(d->cdof_dot + 6*j, cv, d->cdof + 6*j);
      for (int k = 0; k < 6; k++) cv[k] += d->cdof[6*j+k] * d->qvel[j];
    }
    memcpy(d->cvel + 6*b, cv, sizeof cv);
  }
}

static void rne(const rpo_model* m, rpo_data* d) {
  int nb = m->nbody;
  for (int k = 0; k < 3; k++) { d->cacc[k] = 0; d->cacc[3+k] = -m->gravity[k]; }
  for (int b = 1; b < nb; b++) {
    double ca[6];
    memcpy(ca, d->cacc + 6*m->body_parentid[b], sizeof ca);
    for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++)
      for (int k = 0; k < 6; k++) ca[k] += d->cdof_dot[6*j+k] * d->qvel[j];
    memcpy(d->cacc + 6*b, ca, sizeof ca);
    double f1[6], iv[6], f2[6];
    mul_inert(f1, d->cinert + 10*b, ca);
    mul_inert(iv, d->cinert + 10*b, d->cvel + 6*b);
    cross_force(f2, d->cvel + 6*b, iv);
    for (int k = 0; k < 6; k++) d->cfrc[6*b+k] = f1[k] + f2[k];
  }
  memset(d->cfrc, 0, sizeof(double)*6);
  for (int b = nb - 1; b >= 1; b--) {
    int p = m->body_parentid[b];
    if (p > 0) for (int k = 0; k < 6; k++) d->cfrc[6*p+k] += d->cfrc[6*b+k];
  }
  for (int j = 0; j < m->nv; j++) d->qfrc_bias[j] = dot6(d->cdof + 6*j, d->cfrc + 6*m->jnt_bodyid[j]);
}

static void passive(const rpo_model* m, rpo_data* d) {
  int nv = m->nv;
  for (int j = 0; j < nv; j++)
    d->qfrc_passive[j] = -m->jnt_stiffness[j] * (d->qpos[j] - m->qpos_spring[j])
                         - m->dof_damping[j] * d->qvel[j];
  /* gravity compensation: force -g*m*gravcomp applied at the body com */
  double* jd = NULL;
  for (int b = 1; b < m->nbody; b++) {
    if (m->body_gravcomp[b] == 0 || m->body_mass[b] == 0) continue;
    if (!jd) jd = (double*)malloc(sizeof(double)*3*nv);
    memset(jd, 0, sizeof(double)*3*nv);
    jac_point(m, d, jd, b, d->xipos + 3*b, 1.0);
    double f[3];
    for (int k = 0; k < 3; k++) f[k] = -m->gravity[k] * m->body_mass[b] * m->body_gravcomp[b];
    for (int j = 0; j < nv; j++) d->qfrc_passive[j] += jd[j]*f[0] + jd[nv+j]*f[1] + jd[2*nv+j]*f[2];
  }
  free(jd);
}

static void transmission(const rpo_model* m, rpo_data* d) {
  int nv = m->nv;
  for (int t = 0; t < m->ntendon; t++) {
    double L = 0, V = 0;
    for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) {
      L += m->wrap_prm[w] * d->qpos[m->wrap_objid[w]];
      V += m->wrap_prm[w] * d->qvel[m->wrap_objid[w]];
    }
    d->ten_length[t] = L; d->ten_velocity[t] = V;
  }
  memset(d->act_moment, 0, sizeof(double) * m->nu * nv);
  for (int i = 0; i < m->nu; i++) {
    double gear = m->actuator_gear[i];
    int id = m->actuator_trnid[i];
    if (m->actuator_trntype[i] == TRN_JOINT) {
      d->act_length[i] = gear * d->qpos[id];
      d->act_moment[(size_t)i*nv + id] = gear;
    } else {
      d->act_length[i] = gear * d->ten_length[id];
      for (int w = m->tendon_adr[id]; w < m->tendon_adr[id] + m->tendon_num[id]; w++)
        d->act_moment[(size_t)i*nv + m->wrap_objid[w]] += gear * m->wrap_prm[w];
    }
    double v = 0;
    for (int j = 0; j < nv; j++) v += d->act_moment[(size_t)i*nv + j] * d->qvel[j];
    d->act_velocity[i] = v;
  }
}

static void actuation(const rpo_model* m, rpo_data* d) {
  int nv = m->nv;
  memset(d->qfrc_actuator, 0, sizeof(double)*nv);
  for (int i = 0; i < m->nu; i++) {
    double c = d->ctrl[i];
    if (m->actuator_ctrllimited[i])
      c = fmin(m->actuator_ctrlrange[2*i+1], fmax(m->actuator_ctrlrange[2*i], c));
    const double* bp = m->actuator_biasprm + 3*i;
    double f = m->actuator_gainprm[i]*c + bp[0] + bp[1]*d->act_length[i] + bp[2]*d->act_velocity[i];
    if (m->actuator_forcelimited[i])
      f = fmin(m->actuator_forcerange[2*i+1], fmax(m->actuator_forcerange[2*i], f));
    d->act_force[i] = f;
    for (int j = 0; j < nv; j++) d->qfrc_actuator[j] += d->act_moment[(size_t)i*nv + j] * f;
  }
}

/* --------------------------------------------------------------- solver */
/* updates efc_force/state from efc_jar; returns constraint cost; fills qfrc_constraint */
static double update_constraint(const rpo_model* m, rpo_data* d) {
  int nv = m->nv, ne = d->nefc;
  double cost = 0;
  for (int i = 0; i < ne; i++) {
    double jar = d->efc_jar[i], D = d->efc_D[i];
    if (d->efc_type[i] == EFC_FRICTION) {
      double f = d->efc_floss[i], R = d->efc_R[i];
      if (jar <= -R*f) { d->efc_force[i] = f; d->efc_state[i] = 2; cost += -0.5*R*f*f - f*jar; }
      else if (jar >= R*f) { d->efc_force[i] = -f; d->efc_state[i] = 3; cost += -0.5*R*f*f + f*jar; }
      else { d->efc_force[i] = -D*jar; d->efc_state[i] = 1; cost += 0.5*D*jar*jar; }
    } else {
      if (jar >= 0) { d->efc_force[i] = 0; d->efc_state[i] = 0; }
      else { d->efc_force[i] = -D*jar; d->efc_state[i] = 1; cost += 0.5*D*jar*jar; }
    }
  }
  memset(d->qfrc_constraint, 0, sizeof(double)*nv);
  for (int i = 0; i < ne; i++) {
    double f = d->efc_force[i];
    if (f == 0) continue;
    const double* row = d->efc_J + (size_t)i*nv;
    for (int j = 0; j < nv; j++) d->qfrc_constraint[j] += row[j] * f;
  }
  return cost;
}

static double gauss_cost(const rpo_model* m, const rpo_data* d) {
  double c = 0;
  for (int j = 0; j < m->nv; j++) c += (d->Ma[j] - d->qfrc_smooth[j]) * (d->qacc[j] - d->qacc_smooth[j]);
  return 0.5 * c;
}

/* derivatives of the line cost at alpha: returns phi, sets d1, d2 */
static double ls_eval(const rpo_data* d, const double* quadGauss, double alpha, double* d1, double* d2) {
  double cost = alpha*alpha*quadGauss[2] + alpha*quadGauss[1] + quadGauss[0];
  double g1 = 2*alpha*quadGauss[2] + quadGauss[1], g2 = 2*quadGauss[2];
  for (int i = 0; i < d->nefc; i++) {
    double x = d->efc_jar[i] + alpha * d->efc_jv[i], D = d->efc_D[i], jv = d->efc_jv[i];
    if (d->efc_type[i] == EFC_FRICTION) {
      double f = d->efc_floss[i], R = d->efc_R[i];
      if (x <= -R*f) { cost += -0.5*R*f*f - f*x; g1 += -f*jv; }
      else if (x >= R*f) { cost += -0.5*R*f*f + f*x; g1 += f*jv; }
      else { cost += 0.5*D*x*x; g1 += D*x*jv; g2 += D*jv*jv; }
    } else if (x < 0) { cost += 0.5*D*x*x; g1 += D*x*jv; g2 += D*jv*jv; }
  }
  *d1 = g1; *d2 = g2;
  return cost;
}

/* One point of the line search [MJ: mjPrimalPnt]: phi and its first two derivatives at alpha. */
typedef struct { double alpha, cost, d0, d1; } ls_pnt;

static ls_pnt ls_point(const rpo_data* d, const double* quadGauss, double alpha, int* evals) {
  ls_pnt p;
  p.alpha = alpha;
  p.cost = ls_eval(d, quadGauss, alpha, &p.d0, &p.d1);
  (*evals)++;
  return p;
}

/* [MJ: updateBracket] move one end of the bracket to the candidate on the same side of the root
 * whose slope is closest to zero; if it moved, its Newton successor is evaluated into *pnext. */
static int ls_update_bracket(const rpo_data* d, const double* quadGauss, ls_pnt* p, const ls_pnt cand[3],
                             ls_pnt* pnext, int* evals) {
  int flag = 0;
  for (int i = 0; i < 3; i++) {
    if (p->d0 < 0 && cand[i].d0 < 0 && p->d0 < cand[i].d0) { *p = cand[i]; flag = 1; }
    else if (p->d0 > 0 && cand[i].d0 > 0 && p->d0 > cand[i].d0) { *p = cand[i]; flag = 2; }
  }
  if (flag) *pnext = ls_point(d, quadGauss, p->alpha - p->d0 / p->d1, evals);
  return flag;
}

/* Exact line search along `search` [MJ: engine_solver.c PrimalSearch, restated]: a Newton step from
 * alpha = 0; Newton steps on phi' while the slope keeps its sign (one-sided phase); once the root
 * of phi' is bracketed, each round evaluates the midpoint and the Newton successors of both
 * bracket ends, returns the best candidate whose |phi'| < gtol, otherwise tightens the bracket.
 * Returns alpha (0 = no improvement). */
static double primal_search_n(int ls_iterations, const rpo_data* d, const double* quadGauss, double gtol, int* pev);
static double primal_search(const rpo_model* m, const rpo_data* d, const double* quadGauss, double gtol) {
  int evals = 0;
  return primal_search_n(m->ls_iterations, d, quadGauss, gtol, &evals);
}
static double primal_search_n(int ls_iterations, const rpo_data* d, const double* quadGauss, double gtol, int* pev) {
  ls_pnt p0 = ls_point(d, quadGauss, 0, pev);
  if (!(p0.d1 > 0)) return 0;
  ls_pnt p1 = ls_point(d, quadGauss, p0.alpha - p0.d0 / p0.d1, pev);   /* always one Newton step */
  if (p0.cost < p1.cost) p1 = p0;
  if (fabs(p1.d0) < gtol) return p1.alpha;
  int dir = p1.d0 < 0 ? 1 : -1;
  ls_pnt p2 = p1;
  int p2update = 0;
  while (p1.d0 * dir <= -gtol && (*pev) < ls_iterations) {               /* one-sided search */
    p2 = p1; p2update = 1;
    p1 = ls_point(d, quadGauss, p1.alpha - p1.d0 / p1.d1, pev);
    if (fabs(p1.d0) < gtol) return p1.alpha;
  }
  if ((*pev) >= ls_iterations || !p2update) return p1.alpha;             /* failed to bracket */
  ls_pnt p2next = p1;
  ls_pnt p1next = ls_point(d, quadGauss, p1.alpha - p1.d0 / p1.d1, pev);
  while ((*pev) < ls_iterations) {                                        /* bracketed search */
    ls_pnt pmid = ls_point(d, quadGauss, 0.5 * (p1.alpha + p2.alpha), pev);
    ls_pnt cand[3] = {p1next, p2next, pmid};
    int best = -1;
    for (int i = 0; i < 3; i++)
      if (fabs(cand[i].d0) < gtol && (best < 0 || cand[i].cost < cand[best].cost)) best = i;
    if (best >= 0) return cand[best].alpha;
    int b1 = ls_update_bracket(d, quadGauss, &p1, cand, &p1next, pev);
    int b2 = ls_update_bracket(d, quadGauss, &p2, cand, &p2next, pev);
    if (!b1 && !b2) return pmid.alpha;                                      /* numerical accuracy reached */
  }
  if (p1.cost <= p2.cost && p1.cost < p0.cost) return p1.alpha;
  if (p2.cost <= p1.cost && p2.cost < p0.cost) return p2.alpha;
  return 0;
}

static void mul_J(const rpo_model* m, const rpo_data* d, double* res, const double* v) {
  int nv = m->nv;
  for (int i = 0; i < d->nefc; i++) {
    const double* row = d->efc_J + (size_t)i*nv;
    double s = 0;
    for (int j = 0; j < nv; j++) s += row[j] * v[j];
    res[i] = s;
  }
}

/* search = -H^{-1} grad with H = M + J' diag(D_active) J.  H is block diagonal:
 * dofs without off-diagonal entries are scalars, the rest (the hand trees and
 * any key they touch) form one dense block. */
/* Debug statistics (not thread safe; single-env experiments only): Newton directions computed, and how many of them had
 * the SAME set of quadratic rows as the direction before in the same solve, i.e. an unchanged Hessian. */
static long g_dbg_newton_dirs = 0, g_dbg_newton_same_hessian = 0;
static int g_dbg_prev_quad[4096], g_dbg_prev_n = -1;
void rpo_debug_newton_stats(long* out, int reset) {
  out[0] = g_dbg_newton_dirs; out[1] = g_dbg_newton_same_hessian;
  if (reset) g_dbg_newton_dirs = g_dbg_newton_same_hessian = 0;
}
static void newton_direction(const rpo_model* m, rpo_data* d) {
  int nv = m->nv, ne = d->nefc;
  double* H = d->H;
  if (ne <= 4096) {
    int same = g_dbg_prev_n == ne;
    for (int i = 0; i < ne; i++) {
      int q = d->efc_state[i] == 1;
      if (same && g_dbg_prev_quad[i] != q) same = 0;
      g_dbg_prev_quad[i] = q;
    }
    g_dbg_prev_n = ne;
    g_dbg_newton_dirs++; g_dbg_newton_same_hessian += same;
  }
  memcpy(H, d->qM, sizeof(double)*nv*nv);
  for (int i = 0; i < ne; i++) {
    if (d->efc_state[i] != 1) continue;
    const double* row = d->efc_J + (size_t)i*nv;
    double D = d->efc_D[i];
    int nz[64], nnz = 0;
    for (int j = 0; j < nv && nnz < 64; j++) if (row[j] != 0) nz[nnz++] = j;
    for (int a = 0; a < nnz; a++) for (int b = 0; b < nnz; b++)
      H[nz[a]*nv + nz[b]] += D * row[nz[a]] * row[nz[b]];
  }
  int ns = 0;
  for (int i = 0; i < nv; i++) {
    int coupled = 0;
    for (int j = 0; j < nv; j++) if (j != i && H[i*nv+j] != 0) { coupled = 1; break; }
    if (coupled) d->act_idx[ns++] = i;
    else d->search[i] = -d->grad[i] / H[i*nv+i];
  }
  if (!ns) return;
  double* A = (double*)malloc(sizeof(double)*ns*ns);
  double* x = (double*)malloc(sizeof(double)*ns);
  for (int a = 0; a < ns; a++) {
    x[a] = d->grad[d->act_idx[a]];
    for (int b = 0; b < ns; b++) A[a*ns+b] = H[d->act_idx[a]*nv + d->act_idx[b]];
  }
  for (int k = 0; k < ns; k++) {
    double s = A[k*ns+k];
    for (int p = 0; p < k; p++) s -= A[k*ns+p]*A[k*ns+p];
    if (s < MINVAL) { s = MINVAL; d->warnings |= 4; }
    double lkk = sqrt(s);
    A[k*ns+k] = lkk;
    for (int i = k+1; i < ns; i++) {
      double v = A[i*ns+k];
      for (int p = 0; p < k; p++) v -= A[i*ns+p]*A[k*ns+p];
      A[i*ns+k] = v / lkk;
    }
  }
  for (int i = 0; i < ns; i++) {
    double s = x[i];
    for (int p = 0; p < i; p++) s -= A[i*ns+p]*x[p];
    x[i] = s / A[i*ns+i];
  }
  for (int i = ns-1; i >= 0; i--) {
    double s = x[i];
    for (int p = i+1; p < ns; p++) s -= A[p*ns+i]*x[p];
    x[i] = s / A[i*ns+i];
  }
  for (int a = 0; a < ns; a++) d->search[d->act_idx[a]] = -x[a];
  free(A); free(x);
}

static void solve_newton(const rpo_model* m, rpo_data* d) {
  int nv = m->nv, ne = d->nefc;
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  d->solver_iter = 0;
  if (ne == 0) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double)*nv);
    memset(d->qfrc_constraint, 0, sizeof(double)*nv);
    return;
  }
  /* warmstart [MJ: warmstart()]: keep qacc_warmstart only if it costs less than qacc_smooth */
  mul_J(m, d, d->efc_jar, d->qacc_smooth);
  for (int i = 0; i < ne; i++) d->efc_jar[i] -= d->efc_aref[i];
  double cost_smooth = update_constraint(m, d); /* Gauss term is zero at qacc_smooth */
  memcpy(d->qacc, d->qacc_warmstart, sizeof(double)*nv);
  mul_M(m, d, d->Ma, d->qacc);
  mul_J(m, d, d->efc_jar, d->qacc);
  for (int i = 0; i < ne; i++) d->efc_jar[i] -= d->efc_aref[i];
  double cost = update_constraint(m, d) + gauss_cost(m, d);
  if (cost > cost_smooth) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double)*nv);
    mul_M(m, d, d->Ma, d->qacc);
    mul_J(m, d, d->efc_jar, d->qacc);
    for (int i = 0; i < ne; i++) d->efc_jar[i] -= d->efc_aref[i];
    cost = update_constraint(m, d) + gauss_cost(m, d);
  }
  for (int j = 0; j < nv; j++) d->grad[j] = d->Ma[j] - d->qfrc_smooth[j] - d->qfrc_constraint[j];

  g_dbg_prev_n = -1;
  for (int iter = 0; iter < m->iterations; iter++) {
    newton_direction(m, d);
    double snorm = 0;
    for (int j = 0; j < nv; j++) snorm += d->search[j]*d->search[j];
    snorm = sqrt(snorm);
    if (snorm < MINVAL) break;
    mul_M(m, d, d->Mv, d->search);
    mul_J(m, d, d->efc_jv, d->search);
    double quadGauss[3] = {gauss_cost(m, d), 0, 0};
    for (int j = 0; j < nv; j++) {
      quadGauss[1] += d->search[j] * (d->Ma[j] - d->qfrc_smooth[j]);
      quadGauss[2] += 0.5 * d->search[j] * d->Mv[j];
    }
    /* exact line search [MJ: PrimalSearch]; gtol = tolerance * ls_tolerance * |search| * meaninertia * max(1, nv) */
    double gtol = m->tolerance * m->ls_tolerance * snorm / scale;
    double alpha = primal_search(m, d, quadGauss, gtol);
    if (!(alpha > 0)) break;   /* no improvement */
    for (int j = 0; j < nv; j++) { d->qacc[j] += alpha*d->search[j]; d->Ma[j] += alpha*d->Mv[j]; }
    for (int i = 0; i < ne; i++) d->efc_jar[i] += alpha * d->efc_jv[i];
    double oldcost = cost;
    cost = update_constraint(m, d) + gauss_cost(m, d);
    d->solver_iter = iter + 1;
    double gn = 0;
    for (int j = 0; j < nv; j++) {
      d->grad[j] = d->Ma[j] - d->qfrc_smooth[j] - d->qfrc_constraint[j];
      gn += d->grad[j]*d->grad[j];
    }
    if (scale * (oldcost - cost) < m->tolerance || scale * sqrt(gn) < m->tolerance) break;
  }
}

/* ---------------------------------------------------------------- stages */
static void check_state(const rpo_model* m, rpo_data* d) {
  for (int j = 0; j < m->nv; j++)
    if (!(fabs(d->qpos[j]) < 1e10) || !(fabs(d->qvel[j]) < 1e10)) d->warnings |= 1;
}

static void step1(const rpo_model* m, rpo_data* d) { /* position + velocity */
  check_state(m, d);
  kinematics(m, d);
  com_pos(m, d);
  crb(m, d);
  factor_ld(m, d->qLD, d->qM, NULL, 0);
  collision(m, d);
  make_constraint(m, d);
  transmission(m, d);
  com_vel(m, d);
  passive(m, d);
  /* reference acceleration [MJ: mj_referenceConstraint] */
  int nv = m->nv;
  for (int i = 0; i < d->nefc; i++) {
    const double* row = d->efc_J + (size_t)i*nv;
    double v = 0;
    for (int j = 0; j < nv; j++) v += row[j] * d->qvel[j];
    d->efc_vel[i] = v;
    d->efc_aref[i] = -d->efc_B[i]*v - d->efc_K[i]*d->efc_imp[i]*(d->efc_pos[i] - d->efc_margin[i]);
  }
  rne(m, d);
}

/* ------------------------------------------------ acceleration-stage sensors
 * [MJ: mj_rnePostConstraint + mj_sensorAcc, the two sensor types the hands declare at
 * robopianist/models/hands/shadow_hand.py:209-226 (torque) and :248-270 (touch)].
 *   cfrc_ext : contact forces on each body, as spatial force about the tree's subtree_com;
 *   cfrc_int : interaction force between a body and its parent = sum over the subtree of
 *              (I cacc + cvel x* I cvel - cfrc_ext), cacc from the CONSTRAINED qacc;
 *   torque sensor of joint j (a site at the origin of the joint's body, shadow_hand.py:211-219):
 *              cfrc_int of that body moved to the body origin, rotated into the body frame; the
 *              observable projects it on the joint axis (hands/base.py:101-109), so the oracle
 *              reports sens_torque[j] = (xmat_b jnt_axis_j) . torque directly;
 *   touch sensor of a site: sum of the normal forces of the contacts of the site's body whose
 *              force ray from the contact point hits the site's sphere. */
static void sensor_acc(const rpo_model* m, rpo_data* d) {
  int nb = m->nbody, nv = m->nv;
  memset(d->cfrc_ext, 0, sizeof(double) * 6 * nb);
  memset(d->sens_touch, 0, sizeof(double) * (m->nsite ? m->nsite : 1));
  for (int ic = 0; ic < d->ncon; ic++) {
    const contact_t* c = d->contact + ic;
    const double* f = d->efc_force + d->first_con_row + 4*ic;
    /* [MJ: mju_decodePyramid] */
    double fn = f[0] + f[1] + f[2] + f[3];
    double f1 = c->friction[0] * (f[0] - f[1]), f2 = c->friction[1] * (f[2] - f[3]);
    double F[3];
    for (int k = 0; k < 3; k++) F[k] = fn*c->frame[k] + f1*c->frame[3+k] + f2*c->frame[6+k];
    int body[2] = {m->geom_bodyid[c->geom1], m->geom_bodyid[c->geom2]};
    for (int side = 0; side < 2; side++) {  /* the force acts on geom2's body, its reaction on geom1's */
      int b = body[side];
      if (b == 0) continue;
      double sg = side ? 1.0 : -1.0;
      const double* ref = d->subtree_com + 3*m->body_rootid[b];
      double r[3] = {c->pos[0]-ref[0], c->pos[1]-ref[1], c->pos[2]-ref[2]}, t[3];
      cross3(t, r, F);
      for (int k = 0; k < 3; k++) { d->cfrc_ext[6*b+k] += sg*t[k]; d->cfrc_ext[6*b+3+k] += sg*F[k]; }
    }
    if (fn <= 0 || !m->site_touch_radius) continue;
    for (int s = 0; s < m->nsite; s++) {
      double rad = m->site_touch_radius[s];
      int sb = m->site_bodyid[s];
      if (rad <= 0 || (sb != body[0] && sb != body[1])) continue;
      /* ray from the contact point along the normal force (flipped when the sensor is on body 2)
       * against the sphere [MJ: mju_rayGeom, sphere] */
      double sg = (sb == body[1]) ? -1.0 : 1.0;
      double o[3] = {c->pos[0]-d->site_xpos[3*s], c->pos[1]-d->site_xpos[3*s+1], c->pos[2]-d->site_xpos[3*s+2]};
      double bq = sg * dot3(o, c->frame), cq = dot3(o, o) - rad*rad;
      double det = bq*bq - cq;
      if (det < 1e-15) continue;
      if (-bq + sqrt(det) >= 0) d->sens_touch[s] += fn;
    }
  }
  for (int k = 0; k < 3; k++) { d->cacc_post[k] = 0; d->cacc_post[3+k] = -m->gravity[k]; }
  memset(d->cfrc_int, 0, sizeof(double) * 6);
  for (int b = 1; b < nb; b++) {
    double ca[6];
    memcpy(ca, d->cacc_post + 6*m->body_parentid[b], sizeof ca);
    for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++)
      for (int k = 0; k < 6; k++) ca[k] += d->cdof_dot[6*j+k] * d->qvel[j] + d->cdof[6*j+k] * d->qacc[j];
    memcpy(d->cacc_post + 6*b, ca, sizeof ca);
    double f1[6], iv[6], f2[6];
    mul_inert(f1, d->cinert + 10*b, ca);
    mul_inert(iv, d->cinert + 10*b, d->cvel + 6*b);
    cross_force(f2, d->cvel + 6*b, iv);
    for (int k = 0; k < 6; k++) d->cfrc_int[6*b+k] = f1[k] + f2[k] - d->cfrc_ext[6*b+k];
  }
  for (int b = nb - 1; b >= 1; b--) {
    int p = m->body_parentid[b];
    if (p > 0) for (int k = 0; k < 6; k++) d->cfrc_int[6*p+k] += d->cfrc_int[6*b+k];
  }
  for (int j = 0; j < nv; j++) {
    int b = m->jnt_bodyid[j];
    const double* ref = d->subtree_com + 3*m->body_rootid[b];
    const double* fi = d->cfrc_int + 6*b;
    /* moment about the body origin: M_p = M_ref + (ref - p) x F  [MJ: mju_transformSpatial, force] */
    double r[3] = {ref[0]-d->xpos[3*b], ref[1]-d->xpos[3*b+1], ref[2]-d->xpos[3*b+2]}, t[3], ax[3];
    cross3(t, r, fi + 3);
    mat_vec(ax, d->xmat + 9*b, m->jnt_axis + 3*j);
    d->sens_torque[j] = (fi[0]+t[0])*ax[0] + (fi[1]+t[1])*ax[1] + (fi[2]+t[2])*ax[2];
  }
}

static void acceleration_stage(const rpo_model* m, rpo_data* d) {
  int nv = m->nv;
  actuation(m, d);
  for (int j = 0; j < nv; j++)
    d->qfrc_smooth[j] = d->qfrc_passive[j] - d->qfrc_bias[j] + d->qfrc_applied[j] + d->qfrc_actuator[j];
  memcpy(d->qacc_smooth, d->qfrc_smooth, sizeof(double)*nv);
  solve_ld(m, d->qLD, d->qacc_smooth);
  solve_newton(m, d);
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double)*nv);
  sensor_acc(m, d);
}

static void euler(const rpo_model* m, rpo_data* d) {
  int nv = m->nv;
  double h = m->timestep;
  double* qacc = d->tmp;
  if (m->has_damping) {
    for (int j = 0; j < nv; j++) qacc[j] = d->qfrc_smooth[j] + d->qfrc_constraint[j];
    factor_ld(m, d->qLDe, d->qM, m->dof_damping, h);
    solve_ld(m, d->qLDe, qacc);
  } else memcpy(qacc, d->qacc, sizeof(double)*nv);
  for (int j = 0; j < nv; j++) { d->qvel[j] += h * qacc[j]; d->qpos[j] += h * d->qvel[j]; }
  d->time += h;
}

void rpo_forward(const rpo_model* m, rpo_data* d) { step1(m, d); acceleration_stage(m, d); }
/* mj_step1 alone: the position / velocity stage of the CURRENT qpos / qvel, nothing else touched (in particular not
 * qacc_warmstart, which rpo_forward's acceleration stage overwrites).  What a teacher-forced replay of a recording
 * needs between writing a recorded state and rpo_step (= mj_step2 on the stage data in place; mj_step1 of the result). */
void rpo_step1(const rpo_model* m, rpo_data* d) { step1(m, d); }

void rpo_reset(const rpo_model* m, rpo_data* d) {
  int nv = m->nv;
  memcpy(d->qpos, m->qpos0, sizeof(double)*nv);
  memset(d->qvel, 0, sizeof(double)*nv);
  memset(d->qacc, 0, sizeof(double)*nv);
  memset(d->qacc_warmstart, 0, sizeof(double)*nv);
  memset(d->qfrc_applied, 0, sizeof(double)*nv);
  memset(d->ctrl, 0, sizeof(double)*m->nu);
  d->time = 0; d->warnings = 0;
  rpo_forward(m, d);
}

void rpo_step(const rpo_model* m, rpo_data* d) {
  acceleration_stage(m, d); /* mj_step2: uses the position/velocity stage of the current state */
  euler(m, d);
  step1(m, d);              /* mj_step1 for the new state */
}

int rpo_ncon(const rpo_data* d) { return d->ncon; }
int rpo_nefc(const rpo_data* d) { return d->nefc; }
int rpo_solver_iter(const rpo_data* d) { return d->solver_iter; }
int rpo_warnings(const rpo_data* d) { return d->warnings; }

double* rpo_get_ptr(const rpo_model* m, rpo_data* d, int field) {
  switch (field) {
    case RPO_QPOS: return d->qpos; case RPO_QVEL: return d->qvel; case RPO_QACC: return d->qacc;
    case RPO_QACC_WARMSTART: return d->qacc_warmstart; case RPO_CTRL: return d->ctrl;
    case RPO_QFRC_APPLIED: return d->qfrc_applied; case RPO_ACTUATOR_FORCE: return d->act_force;
    case RPO_ACTUATOR_VELOCITY: return d->act_velocity; case RPO_ACTUATOR_LENGTH: return d->act_length;
    case RPO_XPOS: return d->xpos; case RPO_XMAT: return d->xmat; case RPO_GEOM_XPOS: return d->geom_xpos;
    case RPO_GEOM_XMAT: return d->geom_xmat; case RPO_SITE_XPOS: return d->site_xpos;
    case RPO_QM: return d->qM; case RPO_QFRC_BIAS: return d->qfrc_bias;
    case RPO_QFRC_PASSIVE: return d->qfrc_passive; case RPO_QFRC_ACTUATOR: return d->qfrc_actuator;
    case RPO_QFRC_SMOOTH: return d->qfrc_smooth; case RPO_QACC_SMOOTH: return d->qacc_smooth;
    case RPO_QFRC_CONSTRAINT: return d->qfrc_constraint; case RPO_EFC_FORCE: return d->efc_force;
    case RPO_EFC_AREF: return d->efc_aref; case RPO_EFC_D: return d->efc_D;
    case RPO_EFC_POS: return d->efc_pos; case RPO_EFC_J: return d->efc_J;
    case RPO_CONTACT:
      for (int i = 0; i < d->ncon; i++) {
        double* o = d->contact_out + 16*i; const contact_t* c = d->contact + i;
        o[0] = c->dist; memcpy(o+1, c->pos, 3*sizeof(double)); memcpy(o+4, c->frame, 9*sizeof(double));
        o[13] = c->geom1; o[14] = c->geom2; o[15] = c->friction[0];
      }
      return d->contact_out;
    case RPO_TIME: return &d->time;
    case RPO_BODY_POS: return ((rpo_model*)m)->body_pos;
    case RPO_SENSOR_TORQUE: return d->sens_torque;
    case RPO_SENSOR_TOUCH: return d->sens_touch;
    case RPO_CFRC_INT: return d->cfrc_int;
    case RPO_SUBTREE_COM: return d->subtree_com;
  }
  return NULL;
}

static double now_s(void) {
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

double rpo_bench(const rpo_model* m, int nenv, int nstep, const double* ctrl, int nthreads,
                 double* qpos_out) {
  rpo_data** ds = (rpo_data**)calloc(nenv, sizeof(rpo_data*));
  for (int e = 0; e < nenv; e++) {
    ds[e] = rpo_data_new(m);
    rpo_reset(m, ds[e]);
    if (ctrl) memcpy(ds[e]->ctrl, ctrl + (size_t)e*m->nu, sizeof(double)*m->nu);
  }
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
  double t0 = now_s();
#pragma omp parallel for schedule(dynamic, 1)
  for (int e = 0; e < nenv; e++)
    for (int s = 0; s < nstep; s++) rpo_step(m, ds[e]);
  double t1 = now_s();
  for (int e = 0; e < nenv; e++) {
    if (qpos_out) memcpy(qpos_out + (size_t)e*m->nv, ds[e]->qpos, sizeof(double)*m->nv);
    rpo_data_free(ds[e]);
  }
  free(ds);
  return t1 - t0;
}

/* Like rpo_bench, but every env REPLAYS an action stream: env e applies row (start[e] + s / hold) mod T of
 * ctrl_seq [T][nu] at mj_step s (hold = mj_steps per control step), i.e. the benchmark's own workload -- rows
 * advancing every control step, every env at its own episode phase -- instead of one constant row. */
double rpo_bench_seq(const rpo_model* m, int nenv, int nstep, const double* ctrl_seq, int T, int hold,
                     const int* start, int nthreads, double* qpos_out) {
  rpo_data** ds = (rpo_data**)calloc(nenv, sizeof(rpo_data*));
  for (int e = 0; e < nenv; e++) { ds[e] = rpo_data_new(m); rpo_reset(m, ds[e]); }
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
  double t0 = now_s();
#pragma omp parallel for schedule(dynamic, 1)
  for (int e = 0; e < nenv; e++)
    for (int s = 0; s < nstep; s++) {
      if (s % hold == 0)
        memcpy(ds[e]->ctrl, ctrl_seq + (size_t)((start[e] + s / hold) % T) * m->nu, sizeof(double) * m->nu);
      rpo_step(m, ds[e]);
    }
  double t1 = now_s();
  for (int e = 0; e < nenv; e++) {
    if (qpos_out) memcpy(qpos_out + (size_t)e*m->nv, ds[e]->qpos, sizeof(double)*m->nv);
    rpo_data_free(ds[e]);
  }
  free(ds);
  return t1 - t0;
}

/* ---- test hook (rp_oracle.h): PrimalSearch on a hand-made one-dimensional problem */
double rpo_debug_line_search(int n, const int* type, const double* jar, const double* jv, const double* D,
                             const double* floss, const double* R, const double quad[3], double gtol,
                             int ls_iterations, int* evals) {
  rpo_data d;
  memset(&d, 0, sizeof d);
  d.nefc = n;
  d.efc_type = (int*)type; d.efc_jar = (double*)jar; d.efc_jv = (double*)jv; d.efc_D = (double*)D;
  d.efc_floss = (double*)floss; d.efc_R = (double*)R;
  int ev = 0;
  const double a = primal_search_n(ls_iterations, &d, quad, gtol, &ev);
  if (evals) *evals = ev;
  return a;
}
