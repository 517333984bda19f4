// rp_kernels.hpp — the two stage kernels of the batched step (position/velocity stage and
// constraint-solver stage), their LDS layouts, and the reset kernel.  Lane roles, tables and
// hand-over buffers: rp_model.hpp.
#pragma once
#include <type_traits>
#include "rp_model.hpp"
#include "rp_wave.hpp"
#include "rp_narrow.hpp"
#include "rp_collide.hpp"
#include "rp_dense.hpp"
#include "rp_solver2.hpp"

#ifdef RPK_NO_BOXBOX   // perf experiment: what the box-box routine costs the position kernel
#define RPK_BOXBOX(...) 0
#else
// (the box-box routine is a real call: its arguments and result live in memory, so they are copies --
// the capsule paths' own arrays stay in registers)
// (up to eight points: the first three go on in registers like every other pair's, points four to eight stay in
// the memory-resident array `bx_` and are emitted from there)
#define RPK_BOXBOX(rc_, bx_, pA_, mA_, sA_, pB_, mB_, sB_) [&]() -> int {                       \
    T a_[15], b_[15]; RawCon<T> o_[3];                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < 3; i_++) { a_[i_] = (pA_)[i_]; b_[i_] = (pB_)[i_]; a_[12 + i_] = (sA_)[i_]; b_[12 + i_] = (sB_)[i_]; } \
    _Pragma("unroll") for (int i_ = 0; i_ < 9; i_++) { a_[3 + i_] = (mA_)[i_]; b_[3 + i_] = (mB_)[i_]; }                     \
    const int n_ = box_box(o_, bx_, a_, a_ + 3, a_ + 12, b_, b_ + 3, b_ + 12);                  \
    (rc_)[0] = o_[0]; (rc_)[1] = o_[1]; (rc_)[2] = o_[2];                                       \
    return n_; }()
#endif
namespace rpk {
// ----------------------------------------------------------------- shared memory
// LDS budget drives occupancy (fp64: 40.5 KB -> 4 workgroups per CU).  Scratch of the
// position/velocity stage and scratch of the solver are never live together, so they
// share storage; only what crosses from one stage into the other is persistent.
template <typename T>
struct SmemShared {  // used by both stages
  union {
    T vec[2][RPK_WAVE];
    short work[RPK_WORK][2];  // position stage, collision only: candidate pairs
  };
  unsigned long long slotmask[16];
  short slotkey[16];
  short slotlink[16];
  signed char keyslot[RPK_NKEYS];
  unsigned prof[RPK_NPROF_STAGE];   // per-launch phase cycle counts of env 0 (debug aid)
#ifdef RPK_OCC_TEST  // occupancy experiment: pad the LDS footprint to force one workgroup per SIMD
  char occ_pad[RPK_OCC_TEST];
#endif
};
template <typename T, int MODE, int MD = RPK_MAXD> struct Smem;
// ---- position / velocity stage (mj_step1)
template <typename T, int MD>
struct Smem<T, 0, MD> : SmemShared<T> {
  T xpos[RPK_NLX(MD)][3];
  T xmat[RPK_NLX(MD)][9];
  T xaxis[RPK_NLX(MD)][3];
  T xanchor[RPK_NLX(MD)][3];
  union {
    T cdof[RPK_NLX(MD)][6];  // until the mass-matrix rows are built
    T vel[RPK_NLX(MD)][6];   // velocity stage: spatial velocities / accelerations
    struct {
      float gbox[RPK_NBOXF][12];  // in between (collision): world frame + half sizes of the boxes
      unsigned short glist[RPK_GLIST];   // ... and the compacted geom-geom candidates of the drain rounds (owner << 6 | partner)
      unsigned short klist[RPK_KLIST];   // ... and the geom-key candidates (geom << 7 | key)
    };
  };
  union {
    T acc[RPK_NLX(MD)][10];  // composite inertias, then subtree forces
    struct {            // in between (collision .. contact Jacobians):
      float gax[RPK_WAVE][4];  // fp32 capsule axes for the candidate prefilter: world axis, half-length
      float grr[RPK_WAVE];     // radius (bounding radius for boxes)
      union {
        struct {
          T cpos[RPK_NCL][3];      // contact points, normals, distances of the first RPK_NCL contacts (the rest: RpStage::covf)
          T cn[RPK_NCL][3];
          T cdist[RPK_NCL];
          T cpar[RPK_NCL][4];      // mu, kterm (K*imp*dist), B, D
        };
        int clist[RPK_NCAND];      // split stage, front part: the candidates that passed the prefilters (ga | gb << 16)
      };
    };
  };
  T gpos[RPK_WAVE][3];
  T kq[RPK_NKEYS];
  T keyvec[1][RPK_NKEYS];
  int cA[RPK_NCL], cB[RPK_NCL], cgA[RPK_NCL], cgB[RPK_NCL];
};
// ---- front part of the split position stage (PART 1): kinematics, composite inertias, broad phase, prefilters.  Only
// what those need: 13 312 B, so that TWELVE workgroups fit a CU (three waves per SIMD; the part needs 164 registers)
// where the whole stage's 19.7 KB allow eight.
template <typename T, int MD>
struct SmemFront {
  unsigned prof[RPK_NPROF_STAGE];
  T xpos[RPK_NLX(MD)][3];
  T xmat[RPK_NLX(MD)][9];
  union {
    T cdof[RPK_NLX(MD)][6];
    struct {
      float gbox[RPK_NBOXF][12];
      unsigned short glist[RPK_GLIST];
      unsigned short klist[RPK_KLIST];
    };
  };
  union {
    T acc[RPK_NLX(MD)][10];
    struct {
      float gax[RPK_WAVE][4];
      float grr[RPK_WAVE];
      int clist[RPK_NCAND];      // the candidates that passed the prefilters (ga | gb << 16)
    };
  };
  T gpos[RPK_WAVE][3];
};
// ---- sensor stage (MODE 2): the position / velocity stage of the state BEFORE the last Euler
// step, plus what the acceleration-stage sensors need (mj_rnePostConstraint, mj_sensorAcc)
template <typename T, int MD>
struct Smem<T, 2, MD> : Smem<T, 0, MD> {
  T fext[RPK_NLX(MD)][6];   // contact forces on each link: spatial force about the tree reference point
  T touch[RPK_WAVE];   // touch sensor sums per engine site
};
// ---- acceleration stage (mj_step2: constraint solver + Euler)
template <typename T, int MD>
struct Smem<T, 1, MD> : SmemShared<T> {
  T R[RPK_WAVE][MD + 1];  // tree factor rows (L), incl. key-leaf rows
  T Dg[RPK_WAVE];               // tree factor diagonal
  T xs[RPK_WAVE];               // solve staging
  T H[(RpCaps<T>::HMAX + 1) * (RpCaps<T>::HMAX + 2) / 2];  // dense block of the cross-coupled rows + rhs row
  T RM[RPK_NLX(MD)][MD + 1];   // mass-matrix rows: RM[i][e] = M[i][anc_e(i)]
  T keyvec[2][RPK_NKEYS];
  T entJ[RpCaps<T>::NE][3];            // contact Jacobian entries (see RpStage)
  int entM[RpCaps<T>::NE][2];
  T cC[RpCaps<T>::NC][6];       // per-contact 3x3 weight of the current Newton iteration
  T cv[RpCaps<T>::NC][3];       // per-contact 3-vector staging (J x, or the contact force)
  T jt[RPK_WAVE];               // J^T f staging, one value per solver row
  T actf[RPK_WAVE];
};

// rows owned by one lane: friction-loss row of its hand dof, one limit row per dof
// slot (hand, key, key+64), four pyramidal rows of its contact.
template <typename T> struct Rows { T fr, lim[3], con[4]; };
}  // namespace rpk

// physics.reset() for the envs selected by a device-side mask (null = all).
template <typename T>
__global__ void rp_reset_kernel(RpState<T> S, const T* qpos0, const unsigned char* mask, int nv, int nu, int nsite,
                                unsigned char* stage_valid) {
  const int env = blockIdx.x;
  if (mask && !mask[env]) return;
  if (threadIdx.x == 0) stage_valid[env] = 0;  // the hand-over of this env no longer matches its state
  for (int i = threadIdx.x; i < nv; i += blockDim.x) {
    S.qpos[(size_t)env * nv + i] = qpos0[i];
    S.qvel[(size_t)env * nv + i] = 0;
    S.warm[(size_t)env * nv + i] = 0;
    S.qfrc_applied[(size_t)env * nv + i] = 0;
  }
  for (int i = threadIdx.x; i < nu; i += blockDim.x) S.ctrl[(size_t)env * nu + i] = 0;
  if (threadIdx.x == 0) { S.time[env] = 0; S.warn[env] = 0; }
  // acceleration-stage sensors: no reading of the previous episode survives a reset (the first observation of the
  // new episode reports zero until the first step has run the sensor stage)
  if (S.sens_torque) for (int i = threadIdx.x; i < nv; i += blockDim.x) S.sens_torque[(size_t)env * nv + i] = 0;
  if (S.sens_touch) for (int i = threadIdx.x; i < nsite; i += blockDim.x) S.sens_touch[(size_t)env * nsite + i] = 0;
}

// ============================================================================
// The stage kernels.  MODE 0: position/velocity stage (mj_step1: kinematics, CRB, collision,
// constraint rows; also physics.forward()).  MODE 1: acceleration stage (mj_step2: Newton
// solver + Euler).  FIXED_TL > 0 specialises the solver for trunks of exactly that many links.
// The host launches  pos, then n_substeps x (sol, pos);  RpStage carries the hand-over.
// ============================================================================
#ifndef RPK_SOL64_WAVES
#define RPK_SOL64_WAVES 1
#endif
// PART (MODE 0 only): 0 = the whole position / velocity stage; 1 = its front part (kinematics, CRB, broad phase, fp32
// prefilters: leaves link / geom frames and the candidate list, RpStage::frames .. tlist); 2 = its back part (collects the
// pooled narrow phase's results in the whole stage's emission order, then constraint rows, Jacobians, velocity stage).
template <typename T, int MODE, int FIXED_TL = 0, int MD = RPK_MAXD, int MESH = 0, bool EXT = false, int PART = 0>
__device__ __forceinline__ void rp_stage_body(const RpModel<T>& M, const RpState<T>& S, const RpStage<T>& B, const int substep,
                                              const int nsub, const int env, void* ext, const int lane) {
  using namespace rpk;
  using N = Num<T>;
  if constexpr (MODE == 1) {
    if (S.lean && B.hdr[env * 8 + 6] == 1) return;   // a light env: rp_lean_solver_kernel steps it
  }
  if constexpr (MODE == 0) {
    if constexpr (PART == 2) { if (B.ncand[env] < 0) return; }   // (the front part did not run for this env: masked, or heavy)
    else if (S.skip_heavy && S.listed[env]) return;   // (its position stage follows its solve on the companion stream)
  }
  const int env_active = S.active ? S.active[env] : 1;  // tested after the prologue loads are in flight
  using SM_ = std::conditional_t<(MODE == 0 && PART == 1), SmemFront<T, MD>, Smem<T, MODE, MD>>;
  SM_& sm = rp_smem<SM_, EXT>(ext);
  constexpr int TC = MD > 9 ? 8 : 4;            // trunk links the chain-blocked solver holds
  constexpr int NT = TC * (TC + 1) / 2, NREC = NT + TC;  // packed trunk block / per-chain record
  // the chain records of the tree elimination live at the end of the dense block's LDS (sm.H); the
  // packed block (rows + its rhs row) may grow up to there
  constexpr int HSIZE = (RpCaps<T>::HMAX + 1) * (RpCaps<T>::HMAX + 2) / 2;
#ifdef RPK_POISON_LDS  // debug build: nothing may depend on what a previous workgroup left in LDS
  {
    unsigned* w_ = reinterpret_cast<unsigned*>(&sm);
    for (int i = threadIdx.x; i < (int)(sizeof(sm) / 4); i += 64) w_[i] = 0xFFF4DEADu;
    WSYNC();
  }
#endif
  int warn = 0;
  if (S.prof && env == 0 && lane < RPK_NPROF_STAGE) sm.prof[lane] = 0;
  long long prof_t = (long long)__builtin_readcyclecounter();
  const long long kernel_t0 = prof_t;
  const int nl = M.nlink, nk = M.nkey, nv = M.nv, nu = M.nu;
  const int HREC0 = HSIZE - M.ntree * 5 * NREC;   // first chain record in sm.H = capacity of the packed dense block
  const T h = M.timestep;

  // ------------------------------------------------------------ lane constants
  const bool isl = lane < nl;
  const int L = isl ? lane : 0;
  // one 64-byte topology record per link lane (engine_tables.py: eng_lane_topo), so that
  // the prologue is a single batch of independent loads instead of a chain of lookups
  int tp[16];
  // (the key tables are requested with the topology record -- clamped indices, selected below -- and the state with
  // one more batch: read under their predicates they were seven dependent trips to L2 in front of the first useful instruction)
  int kdof_raw[2], kact_raw[2];
  {
    const int4* rec = (const int4*)(M.lane_topo() + 16 * L);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int4 v = rec[q];
      tp[4 * q] = v.x; tp[4 * q + 1] = v.y; tp[4 * q + 2] = v.z; tp[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int K = (lane + 64 * s) < M.nkey ? lane + 64 * s : 0;
      kdof_raw[s] = M.key_dof()[K]; kact_raw[s] = M.key_act()[K];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // (not const: the position stage re-reads the record after the collision phase instead of carrying
  // fourteen integers through it -- it is register-bound there)
  int parent = isl ? tp[0] : -1;
  int depth = isl ? tp[1] : -1;
  int jtype = isl ? tp[2] : 0;
  int sibrank = isl ? tp[3] : 0;
  int ltree = isl ? tp[4] : 0;
  int ldof = isl ? tp[5] : 0;
  // chain structure (lanes are in preorder: trunk chain, then up to 5 leaf chains)
  int tbase = isl ? tp[6] : 0, TL = isl ? tp[7] : 0;
  int ndesc = isl ? tp[8] : 0;
  // chain lanes: first depth past the end of my chain / chain index; bit c of chainmask:
  // leaf chain c of my tree exists
  const int chain_end = isl ? tp[11] : 0, mychain = isl ? tp[12] : 0, chainmask = isl ? tp[13] : 0;
  // lane of my ancestor at depth e (e <= depth)
  auto anc_at = [&](int e) -> int { return e < TL ? tbase + e : lane - (depth - e); };
  int llimited = isl ? tp[9] : 0;
  int lact = isl ? tp[10] : -1;
  const T lactcoef = isl ? M.link_act_coef()[L] : (T)0;
  int hasdof[3];
  hasdof[0] = isl && llimited; hasdof[1] = lane < nk; hasdof[2] = lane + 64 < nk;
  const bool isk[2] = {lane < nk, lane + 64 < nk};
  const int kid[2] = {lane, lane + 64};
  int kdof[2], kact[2];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    kdof[s] = isk[s] ? kdof_raw[s] : 0;
    kact[s] = isk[s] ? kact_raw[s] : -1;
  }
// Per-lane model constants are re-read from the (L2-resident) tables inside the stage
// that uses them instead of being pinned in registers for the whole kernel: the kernel
// is register-bound, and `fresh()` keeps the compiler from hoisting the loads back out
// of the substep loop.
#define RPK_LOAD_GEOMETRY                                                                  \
  T lpos[3], lmat[9], laxis[3], lanchor[3], lipos[3], linert[6], tref[3];                  \
  T kpos[2][3], khalf[2][3], krb[2];                                                       \
  {                                                                                        \
    const T *p_lpos = fresh(M.link_lpos()), *p_axis = fresh(M.link_axis()),                    \
            *p_anchor = fresh(M.link_anchor()), *p_ipos = fresh(M.link_ipos()),                \
            *p_tref = fresh(M.tree_ref()), *p_lmat = fresh(M.link_lmat()),                     \
            *p_inert = fresh(M.link_inertia()), *p_kpos = fresh(M.key_pos()),                  \
            *p_khalf = fresh(M.key_half()), *p_krb = fresh(M.key_rbound());                    \
    _Pragma("unroll") for (int k = 0; k < 3; k++) {                                        \
      lpos[k] = isl ? p_lpos[3 * L + k] : (T)0;                                            \
      laxis[k] = isl ? p_axis[3 * L + k] : (T)0;                                           \
      lanchor[k] = isl ? p_anchor[3 * L + k] : (T)0;                                       \
      lipos[k] = isl ? p_ipos[3 * L + k] : (T)0;                                           \
      tref[k] = isl ? p_tref[3 * ltree + k] : (T)0;                                        \
    }                                                                                      \
    _Pragma("unroll") for (int k = 0; k < 9; k++) lmat[k] = isl ? p_lmat[9 * L + k] : (T)0; \
    _Pragma("unroll") for (int k = 0; k < 6; k++) linert[k] = isl ? p_inert[6 * L + k] : (T)0; \
    if (isl && parent < 0 && S.tree_offset) {                                              \
      _Pragma("unroll") for (int k = 0; k < 3; k++)                                        \
        lpos[k] += S.tree_offset[((size_t)env * M.ntree + ltree) * 3 + k];                 \
    }                                                                                      \
    _Pragma("unroll") for (int s = 0; s < 2; s++) {                                        \
      const int K = isk[s] ? kid[s] : 0;                                                   \
      krb[s] = isk[s] ? p_krb[K] : (T)0;                                                   \
      _Pragma("unroll") for (int k = 0; k < 3; k++) {                                      \
        kpos[s][k] = isk[s] ? p_kpos[3 * K + k] : (T)0;                                    \
        khalf[s][k] = isk[s] ? p_khalf[3 * K + k] : (T)0;                                  \
      }                                                                                    \
    }                                                                                      \
  }                                                                                        \
  const T lmass = isl ? fresh(M.link_mass())[L] : (T)0;                                      \
  const T larm = isl ? fresh(M.link_armature())[L] : (T)0;
#define RPK_LOAD_LIMITS                                                                    \
  T lo[3], hi[3], limK[3], limB[3], limW[3];                                               \
  {                                                                                        \
    const T *p_lr = fresh(M.link_range()), *p_kr = fresh(M.key_range());                       \
    lo[0] = isl ? p_lr[2 * L] : (T)0; hi[0] = isl ? p_lr[2 * L + 1] : (T)0;                \
    limK[0] = isl ? fresh(M.link_lim_K())[L] : (T)0;                                         \
    limB[0] = isl ? fresh(M.link_lim_B())[L] : (T)0;                                         \
    limW[0] = isl ? fresh(M.link_invw_dof())[L] : (T)0;                                      \
    _Pragma("unroll") for (int s = 0; s < 2; s++) {                                        \
      const int K = isk[s] ? kid[s] : 0;                                                   \
      lo[1 + s] = isk[s] ? p_kr[2 * K] : (T)0; hi[1 + s] = isk[s] ? p_kr[2 * K + 1] : (T)0; \
      limK[1 + s] = isk[s] ? fresh(M.key_lim_K())[K] : (T)0;                                 \
      limB[1 + s] = isk[s] ? fresh(M.key_lim_B())[K] : (T)0;                                 \
      limW[1 + s] = isk[s] ? fresh(M.key_invw_dof())[K] : (T)0;                              \
    }                                                                                      \
  }                                                                                        \
  const T lflB = isl ? fresh(M.link_fl_B())[L] : (T)0;
#define RPK_LOAD_DYN                                                                       \
  const T ldamp = isl ? fresh(M.link_damping())[L] : (T)0;                                   \
  const T lstiff = isl ? fresh(M.link_stiffness())[L] : (T)0;                                \
  const T lsref = isl ? fresh(M.link_springref())[L] : (T)0;                                 \
  const T lfloss = isl ? fresh(M.link_floss())[L] : (T)0;                                    \
  const T lflR = isl ? fresh(M.link_fl_R())[L] : (T)1;                                       \
  const T lflD = (T)1 / lflR;                                                              \
  T kM[2], kstiff[2], ksref[2], kdamp[2], kmass[2], khx[2];                                \
  _Pragma("unroll") for (int s = 0; s < 2; s++) {                                          \
    const int K = isk[s] ? kid[s] : 0;                                                     \
    kM[s] = isk[s] ? fresh(M.key_M())[K] : (T)1;                                             \
    kstiff[s] = isk[s] ? fresh(M.key_stiffness())[K] : (T)0;                                 \
    ksref[s] = isk[s] ? fresh(M.key_springref())[K] : (T)0;                                  \
    kdamp[s] = isk[s] ? fresh(M.key_damping())[K] : (T)0;                                    \
    kmass[s] = isk[s] ? fresh(M.key_mass())[K] : (T)0;                                       \
    khx[s] = isk[s] ? fresh(M.key_half())[3 * K] : (T)0;                                     \
  }
  // actuator owned by this lane (hand actuators only; key actuators live with the key)
  const bool isa = lane < nu && M.act_kind()[lane < nu ? lane : 0] == 0;
  const int A = lane < nu ? lane : 0;

  // ------------------------------------------------------------------- state
  const size_t eo = (size_t)env * nv;
  T q[3], qd[3], qw[3], qapp[3];
  if constexpr (MODE != 0) {
    // (the solver builds keep the reads under their predicates: they are over their register budget, and the batch below
    // cost the fp32 build 22 more spilled registers)
    q[0] = isl ? S.qpos[eo + ldof] : (T)0; qd[0] = isl ? S.qvel[eo + ldof] : (T)0;
    qw[0] = isl ? S.warm[eo + ldof] : (T)0;
    qapp[0] = (isl && S.qfrc_applied) ? S.qfrc_applied[eo + ldof] : (T)0;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      q[1 + s] = isk[s] ? S.qpos[eo + kdof[s]] : (T)0;
      qd[1 + s] = isk[s] ? S.qvel[eo + kdof[s]] : (T)0;
      qw[1 + s] = isk[s] ? S.warm[eo + kdof[s]] : (T)0;
      qapp[1 + s] = (isk[s] && S.qfrc_applied) ? S.qfrc_applied[eo + kdof[s]] : (T)0;
    }
  } else {
    // (ldof / kdof are 0 for lanes without the dof: every address is the env's own state)
    T q_[3], qd_[3], qw_[3], qa_[3] = {0, 0, 0};
    q_[0] = S.qpos[eo + ldof]; qd_[0] = S.qvel[eo + ldof]; qw_[0] = S.warm[eo + ldof];
#pragma unroll
    for (int s = 0; s < 2; s++) { q_[1 + s] = S.qpos[eo + kdof[s]]; qd_[1 + s] = S.qvel[eo + kdof[s]]; qw_[1 + s] = S.warm[eo + kdof[s]]; }
    if (S.qfrc_applied) {
      qa_[0] = S.qfrc_applied[eo + ldof];
#pragma unroll
      for (int s = 0; s < 2; s++) qa_[1 + s] = S.qfrc_applied[eo + kdof[s]];
    }
    __builtin_amdgcn_sched_barrier(0);
    q[0] = isl ? q_[0] : (T)0; qd[0] = isl ? qd_[0] : (T)0; qw[0] = isl ? qw_[0] : (T)0; qapp[0] = isl ? qa_[0] : (T)0;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      q[1 + s] = isk[s] ? q_[1 + s] : (T)0; qd[1 + s] = isk[s] ? qd_[1 + s] : (T)0;
      qw[1 + s] = isk[s] ? qw_[1 + s] : (T)0; qapp[1 + s] = isk[s] ? qa_[1 + s] : (T)0;
    }
  }
  T ctrl = (lane < nu) ? S.ctrl[(size_t)env * nu + lane] : (T)0;
  if (lane < nu && M.act_ctrllimited()[A])
    ctrl = fmin(M.act_ctrlrange()[2 * A + 1], fmax(M.act_ctrlrange()[2 * A], ctrl));
  T kctrl[2] = {0, 0};
#pragma unroll
  for (int s = 0; s < 2; s++) if (isk[s] && kact[s] >= 0) {
    T c = S.ctrl[(size_t)env * nu + kact[s]];
    if (M.act_ctrllimited()[kact[s]])
      c = fmin(M.act_ctrlrange()[2 * kact[s] + 1], fmax(M.act_ctrlrange()[2 * kact[s]], c));
    kctrl[s] = c;
  }
  T time = S.time[env];
  if (env_active == 0) return;  // masked env (all loads above are speculative and harmless)


  PROF(0);
  // values produced by the position/velocity stage and consumed by the next
  // acceleration stage
  T cdofr[6], qbias = 0, alen = 0, avel = 0;
  T Mr[MD + 1];  // this link's mass-matrix row over its ancestors (diag at [depth])
#pragma unroll
  for (int e = 0; e <= MD; e++) Mr[e] = 0;
  int tree_ok = 1;     // all contacts lie on single root-to-leaf paths (uniform)
  int sdepth = -1;  // solver-slot lanes: anchor link depth
  // Lanes are numbered in preorder (trunk chain, then the leaf chains), so the ancestor
  // of link `lk` at depth e is arithmetic: trunk base + e on the trunk, lk - (depth - e)
  // on lk's own chain.  (depth, trunk length, trunk base) of the links a lane needs:
  int sTL = 0, sTB = 0, salink = 0;          // slot lanes: anchor link of my key
  auto anc_of = [](int lk, int dl, int tl, int tb, int e) -> int { return e < tl ? tb + e : lk - (dl - e); };
  T ksin[2] = {0, 0}, kcos[2] = {1, 1};
  int ncon = 0, nkt = 0, nent = 0, maxm = 0;  // contacts, touched keys, Jacobian entries, max per contact
  // rows
  T fr_aref = 0;
  int lim_sign[3] = {0, 0, 0};
  T lim_D[3] = {0, 0, 0}, lim_aref[3] = {0, 0, 0};
  T con_aref[4] = {0, 0, 0, 0}, con_D = 0, con_mu = 0, con_n[3] = {0, 0, 0}, con_t1[3] = {0, 0, 0},
    con_t2[3] = {0, 0, 0};
  int con_A = -1, con_B = -1, con_slot = -1, con_cross = 0;
  T con_dist = 0;
  unsigned long long dirty_mask = 0;  // rows that need the dense block (uniform)
  unsigned long long con_maskA = 0, con_maskB = 0;
  int niter_last = 0;

  const size_t lf = (size_t)env * RPK_NLF * 64 + lane, li = (size_t)env * RPK_NLI * 64 + lane;
#define LF(i) B.lanef[lf + (size_t)(i) * 64]
#define LI(i) B.lanei[li + (size_t)(i) * 64]
  {
    // ======================================================================
    // MODE 1: ACCELERATION STAGE + EULER  (mj_step2)
    // ======================================================================
    if constexpr (MODE == 1) {
      // ---- what the position/velocity kernel left behind
      {
        ncon = B.hdr[env * 8]; nkt = B.hdr[env * 8 + 1];
        dirty_mask = ((unsigned long long)(unsigned)B.hdr[env * 8 + 3] << 32) | (unsigned)B.hdr[env * 8 + 2];
        nent = B.hdr[env * 8 + 4]; maxm = B.hdr[env * 8 + 5];
        // (fields only some lanes own cross for those lanes only: fewer cache lines per env)
        qbias = LF(0);
        if (lane < nu) { alen = LF(1); avel = LF(2); }
        ksin[0] = LF(3); kcos[0] = LF(5);
        if (isk[1]) { ksin[1] = LF(4); kcos[1] = LF(6); }
        fr_aref = LF(7);
        {
          const int ls = LI(0);
          lim_sign[0] = (ls & 3) - 1; lim_sign[1] = ((ls >> 2) & 3) - 1; lim_sign[2] = ((ls >> 4) & 3) - 1;
        }
#pragma unroll
        for (int k = 0; k < 3; k++) if (lim_sign[k] != 0) { lim_D[k] = LF(8 + k); lim_aref[k] = LF(11 + k); }
        if (lane < ncon) {  // the contact lanes' fields: only those lanes' cache lines cross (other lanes: the defaults)
          con_D = LF(14); con_mu = LF(15);
#pragma unroll
          for (int k = 0; k < 3; k++) { con_n[k] = LF(16 + k); con_t1[k] = LF(19 + k); con_t2[k] = LF(22 + k); }
#pragma unroll
          for (int k = 0; k < 4; k++) con_aref[k] = LF(25 + k);
          con_A = LI(1); con_B = LI(2); con_slot = LI(3); con_cross = LI(4);
          con_maskA = ((unsigned long long)(unsigned)LI(6) << 32) | (unsigned)LI(5);
          con_maskB = ((unsigned long long)(unsigned)LI(8) << 32) | (unsigned)LI(7);
        }
        sdepth = LI(9);
        if (isl) {
#pragma unroll
          for (int e = 0; e <= MD; e++) {
            Mr[e] = B.RM[((size_t)env * RPK_NLX(MD) + lane) * (MD + 1) + e];
            sm.RM[lane][e] = Mr[e];
          }
        }
        for (int i = lane; i < nent; i += 64) {
          const size_t e = (size_t)env * RpCaps<T>::NE + i;
          sm.entJ[i][0] = B.entJ[e * 3]; sm.entJ[i][1] = B.entJ[e * 3 + 1]; sm.entJ[i][2] = B.entJ[e * 3 + 2];
          sm.entM[i][0] = B.entM[e * 2]; sm.entM[i][1] = B.entM[e * 2 + 1];
        }
        if (lane < 16) {
          const int* sl = B.slots + (size_t)env * 64;
          sm.slotkey[lane] = (short)sl[lane];
          sm.slotlink[lane] = (short)sl[16 + lane];
          sm.slotmask[lane] = ((unsigned long long)(unsigned)sl[48 + lane] << 32) | (unsigned)sl[32 + lane];
        }
        if (lane < RPK_NKEYS / 4) ((int*)sm.keyslot)[lane] = B.keyslot[(size_t)env * (RPK_NKEYS / 4) + lane];
        WSYNC();
        {
          const int sl_ = LI(11);
          salink = sl_ & 255; sTL = (sl_ >> 8) & 255; sTB = (sl_ >> 16) & 255;
        }
      }
      RPK_LOAD_DYN
      // ---- actuation [MJ: mj_fwdActuation]
      T aforce = 0;
      if (isa) {
        aforce = M.act_gain()[A] * ctrl + M.act_bias()[3 * A] + M.act_bias()[3 * A + 1] * alen +
                 M.act_bias()[3 * A + 2] * avel;
        if (M.act_forcelimited()[A])
          aforce = fmin(M.act_forcerange()[2 * A + 1], fmax(M.act_forcerange()[2 * A], aforce));
        sm.actf[lane] = aforce;
        S.act_force[(size_t)env * nu + lane] = aforce;
      }
      WSYNC();
      T qfs[3], qs[3];
      {
        T qact = (isl && lact >= 0) ? lactcoef * sm.actf[lact] : (T)0;
        T qpas = -lstiff * (q[0] - lsref) - ldamp * qd[0];
        qfs[0] = isl ? (qpas - qbias + qapp[0] + qact) : (T)0;
#pragma unroll
        for (int s = 0; s < 2; s++) {
          // passive spring/damper, gravity torque m*g*(hx)*cos(q) about +y, actuator
          // (all key constants are 0 for lanes without a key: no branch needed)
          const T grav = -kmass[s] * M.gz * khx[s] * kcos[s] - kmass[s] * M.gx * khx[s] * ksin[s];
          T f = -kstiff[s] * (q[1 + s] - ksref[s]) - kdamp[s] * qd[1 + s] + grav + qapp[1 + s];
          {
            if (isk[s] && kact[s] >= 0) {
              T af = M.act_gain()[kact[s]] * kctrl[s];
              if (M.act_forcelimited()[kact[s]])
                af = fmin(M.act_forcerange()[2 * kact[s] + 1], fmax(M.act_forcerange()[2 * kact[s]], af));
              f += M.act_coef()[2 * kact[s]] * af;
              S.act_force[(size_t)env * nu + kact[s]] = af;
            }
          }
          qfs[1 + s] = f;
          qs[1 + s] = f / kM[s];
        }
      }
      PROF(1);
      // ---- hybrid tree-sparse / dense factor + solve [MJ: mj_factorM / mj_solveLD by levels].
      // Row r of the symmetric system is held by lane r as Rr[e] = A[r][anc_e(r)]
      // (diag at e = depth).  Touched keys (nslots of them, lanes nl..nl+nslots-1)
      // are extra leaves hanging under their anchor link.  Rows whose bit is clear in
      // `dm` ("clean") only couple to their ancestors and are eliminated leaf-to-root
      // with no fill-in.  The rows in `dm` ("dirty": supports of cross-chain contacts,
      // an ancestor-closed set) receive the Schur complement and are solved with a
      // small dense Cholesky.  The caller has already put the cross-contact blocks into the packed
      // block sm.H (zeroed + accumulated during the Hessian assembly); the Schur rows are added here.
      auto tree_solve = [&](T* Rr, T rhs, int nslots, unsigned long long dm) -> T {
        // trunk length of this lane's tree; a compile-time constant in the FIXED_TL build
        // (every predicate on it folds: -20 % instructions in the chain elimination)
        const int TLX = FIXED_TL > 0 ? FIXED_TL : TL;
        // Chain-blocked elimination.  Each tree is a trunk chain (<= TC links) carrying up
        // to five leaf chains (<= 5 links).  The first lane of every chain ("leader")
        // gathers its chain's rows and eliminates the clean links, deepest first, entirely
        // in registers; what that leaves on the trunk is summed per trunk row, the trunk
        // leader eliminates the clean trunk links, the dense block (if any) solves the
        // dirty rows, and the leaders back-substitute.  Same arithmetic as the level-by-
        // level L^T D L, five LDS hand-overs instead of one per tree level.
        const bool isslot = !isl && lane < nl + nslots;
        const bool dirty = (dm >> lane) & 1;
        const int mydiag = isl ? depth : sdepth + 1;
        // LDS reads below are issued unconditionally on in-bounds addresses and the value
        // is selected afterwards: a conditional read costs a branch and serialises the wait.
        T Dslot = 1;
        // ---- key leaves first (they hang under chain links): the slot lanes publish their
        // scaled rows, the link lanes fold them into their register rows
        if (nslots > 0) {
          if (isslot) {
            T Dk = (T)1;
#pragma unroll
            for (int e = 0; e <= MD; e++) if (e == mydiag) Dk = Rr[e];
            if (!dirty) {
              if (!(Dk >= RPK_MINVAL)) { Dk = RPK_MINVAL; warn |= 4; }
              Dslot = Dk;
            }
            const T inv = dirty ? (T)1 : rcp_nr(Dk);
#pragma unroll
            for (int e = 0; e <= MD; e++) if (e <= mydiag) sm.R[lane][e] = e < mydiag ? Rr[e] * inv : Rr[e];
            sm.Dg[lane] = Dk;
            sm.xs[lane] = rhs;
          }
          WSYNC();
          if (isl) {
            for (int sidx = 0; sidx < nslots; sidx++) {
              const T* Lk = sm.R[nl + sidx];
              T lrow[MD + 1];
#pragma unroll
              for (int e = 0; e <= MD; e++) lrow[e] = Lk[e];
              const T lk = Lk[depth], dk = sm.Dg[nl + sidx], xk = sm.xs[nl + sidx];
              if (((sm.slotmask[sidx] >> lane) & 1) && !((dm >> (nl + sidx)) & 1)) {
                const T t = lk * dk;
#pragma unroll
                for (int e = 0; e <= MD; e++) if (e <= depth) Rr[e] -= t * lrow[e];
                rhs -= lk * xk;
              }
            }
          }
        }
        if (isl && depth >= TLX) {
#pragma unroll
          for (int e = 0; e <= MD; e++) if (e <= depth) sm.R[lane][e] = Rr[e];
          sm.xs[lane] = rhs;
        }
        WSYNC();
        PROF(20);
        // ---- chain leaders: local variables 0..TC-1 = trunk, TC..TC+4 = my chain (depth order)
        const bool leader = isl && depth == TLX && TLX > 0;
        const int clen = chain_end - TLX;
        T Ac[5][TC + 5];   // Ac[ci][j]: chain link ci vs local variable j <= TC+ci
        T rc_[5], inv_[5];
        T dT[NT], drT[TC];  // what the eliminated links leave on the trunk block / rhs
        int mc[5];
#pragma unroll
        for (int k = 0; k < NT; k++) dT[k] = 0;
#pragma unroll
        for (int k = 0; k < TC; k++) drT[k] = 0;
        if (leader) {
#pragma unroll
          for (int ci = 0; ci < 5; ci++) {
            const bool pi = ci < clen;
            mc[ci] = pi && !((dm >> (lane + ci)) & 1);
            const T* row = sm.R[lane + ci];   // lane + ci < 64: always in bounds
            const T* rowc = row + TLX;         // chain columns start at depth TLX
            const T xr_ = sm.xs[lane + ci];
            rc_[ci] = pi ? xr_ : (T)0;
#pragma unroll
            for (int j = 0; j < TC + 5; j++) {
              if (j <= TC + ci) {
                const bool pj = j < TC ? j < TLX : (j - TC) < clen;
                const T raw = j < TC ? row[j] : rowc[j - TC];
                Ac[ci][j] = (pi && pj) ? raw : (j == TC + ci ? (T)1 : (T)0);
              }
            }
          }
#pragma unroll
          for (int ci = 4; ci >= 0; ci--) {
            const int v = TC + ci;
            T dv = Ac[ci][v];
            if (mc[ci] && !(dv >= RPK_MINVAL)) { dv = RPK_MINVAL; warn |= 4; }
            const T iv = mc[ci] ? rcp_nr(dv) : (T)0;
            inv_[ci] = iv;
            T l[TC + 4];
#pragma unroll
            for (int i = 0; i < TC + 4; i++) if (i < v) l[i] = Ac[ci][i] * iv;
            // chain rows above me
#pragma unroll
            for (int ci2 = 0; ci2 < 4; ci2++) {
              if (ci2 < ci) {
                const int i = TC + ci2;
#pragma unroll
                for (int j = 0; j < TC + 5; j++) if (j <= i) Ac[ci2][j] -= l[i] * Ac[ci][j];
                rc_[ci2] -= l[i] * rc_[ci];
              }
            }
            // trunk block
#pragma unroll
            for (int i = 0; i < TC; i++) {
#pragma unroll
              for (int j = 0; j < TC; j++) if (j <= i) dT[i * (i + 1) / 2 + j] -= l[i] * Ac[ci][j];
              drT[i] -= l[i] * rc_[ci];
            }
#pragma unroll
            for (int i = 0; i < TC + 4; i++) if (i < v && mc[ci]) Ac[ci][i] = l[i];
          }
          // rows / rhs of the links I did not eliminate go back for the dense block (the
          // LDS rows of eliminated links are dead, so every present row is stored)
#pragma unroll
          for (int ci = 0; ci < 5; ci++) {
            if (ci < clen) {
              T* row = sm.R[lane + ci];
              T* rowc = row + TLX;
#pragma unroll
              for (int j = 0; j < TC; j++) if (j < TLX) row[j] = Ac[ci][j];
#pragma unroll
              for (int j = TC; j < TC + 5; j++) if (j <= TC + ci) rowc[j - TC] = Ac[ci][j];
              sm.xs[lane + ci] = rc_[ci];
            }
          }
          // trunk deltas, one NREC-entry record per chain, at the end of the dense block's storage (the
          // block itself already holds the cross-contact terms; its capacity check leaves this room)
          T* rec = sm.H + HREC0 + (size_t)(ltree * 5 + mychain) * NREC;
#pragma unroll
          for (int k = 0; k < NT; k++) rec[k] = dT[k];
#pragma unroll
          for (int k = 0; k < TC; k++) rec[NT + k] = drT[k];
        }
        WSYNC();
        // ---- trunk rows collect the chains' contributions (fixed order: deterministic)
        if (isl && depth < TLX) {
          const int tro = depth * (depth + 1) / 2;
#pragma unroll
          for (int c = 0; c < 5; c++) {
            const T* rec = sm.H + HREC0 + (size_t)(ltree * 5 + c) * NREC;
            T dv[TC];
#pragma unroll
            for (int e = 0; e < TC; e++) dv[e] = rec[tro + e < NT ? tro + e : NT - 1];
            const T dr = rec[NT + depth];
            const bool has = (chainmask >> c) & 1;
#pragma unroll
            for (int e = 0; e < TC; e++) if (has && e <= depth) Rr[e] += dv[e];
            if (has) rhs += dr;
          }
#pragma unroll
          for (int e = 0; e < TC; e++) if (e <= depth) sm.R[lane][e] = Rr[e];
          sm.xs[lane] = rhs;
        }
        WSYNC();
        // ---- trunk leader eliminates the clean trunk links (deepest first)
        const bool tleader = isl && depth == 0;
        T At[TC][TC], rt[TC], invt[TC];
        int mt[TC];
        if (tleader) {
#pragma unroll
          for (int i = 0; i < TC; i++) {
            const bool pi = i < TLX;
            mt[i] = pi && !((dm >> (lane + i)) & 1);
            const T xin = sm.xs[lane + i];
            rt[i] = pi ? xin : (T)0;
#pragma unroll
            for (int j = 0; j < TC; j++) {
              if (j <= i) {
                const T raw = sm.R[lane + i][j];
                At[i][j] = pi ? raw : (j == i ? (T)1 : (T)0);
              }
            }
          }
#pragma unroll
          for (int v = TC - 1; v >= 0; v--) {
            T dv = At[v][v];
            if (mt[v] && !(dv >= RPK_MINVAL)) { dv = RPK_MINVAL; warn |= 4; }
            const T iv = mt[v] ? rcp_nr(dv) : (T)0;
            invt[v] = iv;
            T l[TC - 1];
#pragma unroll
            for (int i = 0; i < TC - 1; i++) if (i < v) l[i] = At[v][i] * iv;
#pragma unroll
            for (int i = 0; i < TC - 1; i++) {
              if (i < v) {
#pragma unroll
                for (int j = 0; j < TC - 1; j++) if (j <= i) At[i][j] -= l[i] * At[v][j];
                rt[i] -= l[i] * rt[v];
              }
            }
#pragma unroll
            for (int i = 0; i < TC - 1; i++) if (i < v && mt[v]) At[v][i] = l[i];
          }
#pragma unroll
          for (int i = 0; i < TC; i++) {
            if (i < TLX) {  // (rows of eliminated links are dead: store them all)
#pragma unroll
              for (int j = 0; j < TC; j++) if (j <= i) sm.R[lane + i][j] = At[i][j];
              sm.xs[lane + i] = rt[i];
            }
          }
        }
        WSYNC();
        PROF(21);
        // ---- dense block on the dirty rows (Schur complement + cross-contact terms)
        if (dm) {   // (the caller passes dm = 0 when the block would not fit: RP_WARN_DENSE_FULL)
          T x = (isl || isslot) ? sm.xs[lane] : (T)0;
          WSYNC();
          const int nD = __popcll(dm);
          auto cidx = [&](int l) -> int { return __popcll(dm & lanemask_lt(l)); };
          const int ci = cidx(lane);
          // every (row, ancestor) element has exactly one owner lane: plain read-modify-write
          if (dirty) {
            if (isl) {
              for (int e = 0; e <= depth; e++) sm.H[tri(ci, cidx(anc_at(e)))] += sm.R[lane][e];
            } else {
              sm.H[tri(ci, ci)] += sm.R[lane][mydiag];
#pragma unroll
              for (int e = 0; e < MD; e++) {
                if (e <= sdepth) {
                  const int a_ = anc_of(salink, sdepth, sTL, sTB, e);
                  if ((dm >> a_) & 1) sm.H[tri(ci, cidx(a_))] += sm.R[lane][e];
                }
              }
            }
          }
          WSYNC();
          PROF(22);
          PROF(23);
          // the rhs of compact row r (owned by a dirty lane) is row nD of the packed block
          if (dirty) sm.H[tri(nD, 0) + ci] = x;
          WSYNC();
          T xr = dense_factor_solve(sm.H, nD, lane, &warn);
          WSYNC();
          if (lane < nD) sm.Dg[lane] = xr;
          WSYNC();
          if (dirty) sm.xs[lane] = sm.Dg[ci];
          WSYNC();
        }
        PROF(25);
        // ---- back-substitution: trunk leader, then chain leaders, then key leaves
        if (tleader) {
          T xt[TC];
#pragma unroll
          for (int v = 0; v < TC; v++) xt[v] = sm.xs[lane + v];
#pragma unroll
          for (int v = 0; v < TC; v++) {
            T xv = rt[v] * invt[v];
#pragma unroll
            for (int i = 0; i < TC - 1; i++) if (i < v) xv -= At[v][i] * xt[i];
            xt[v] = mt[v] ? xv : (v < TLX ? xt[v] : (T)0);
            if (v < TLX) sm.xs[lane + v] = xt[v];
          }
        }
        WSYNC();
        if (leader) {
          T xl[TC + 5];
#pragma unroll
          for (int j = 0; j < TC; j++) { const T xin = sm.xs[tbase + j]; xl[j] = j < TLX ? xin : (T)0; }
#pragma unroll
          for (int ci = 0; ci < 5; ci++) xl[TC + ci] = sm.xs[lane + ci];
#pragma unroll
          for (int ci = 0; ci < 5; ci++) {
            const int v = TC + ci;
            T xv = rc_[ci] * inv_[ci];
#pragma unroll
            for (int i = 0; i < TC + 4; i++) if (i < v) xv -= Ac[ci][i] * xl[i];
            xl[v] = mc[ci] ? xv : (ci < clen ? xl[v] : (T)0);
            if (ci < clen) sm.xs[lane + ci] = xl[v];
          }
        }
        WSYNC();
        T x = isl ? sm.xs[lane] : (T)0;
        if (isslot) {
          x = sm.xs[lane];
          if (!dirty) {
            x = rhs / Dslot;
#pragma unroll
            for (int e = 0; e < MD; e++)
              if (e <= sdepth) x -= sm.R[lane][e] * sm.xs[anc_of(salink, sdepth, sTL, sTB, e)];
          }
        }
        WSYNC();
        return x;
      };
      // ---- qacc_smooth = M^-1 qfrc_smooth (M is always tree-sparse)
      {
        T Rr[MD + 1];
#pragma unroll
        for (int e = 0; e <= MD; e++) Rr[e] = Mr[e];
        qs[0] = tree_solve(Rr, qfs[0], 0, 0ull);
      }

      PROF(2);
      // ---- constraint solve [MJ: mj_solNewton]
      const int nsys = nl + nkt;
      const bool hascon = lane < ncon && con_D > 0;
      const T scale = (T)1 / (M.meaninertia * (T)(nv > 1 ? nv : 1));
      T qa[3], Ma[3], qfc[3];
      Rows<T> jar, frc;
      int fr_quad = 0, lim_act[3] = {0, 0, 0}, con_act[4] = {0, 0, 0, 0};

      // Contact terms run on "entry lanes": entry = (contact, one dof it touches) with the
      // 3-vector d(contact point velocity)/d(qvel of that dof).  Lane l of pass p owns entry
      // 64 p + l; sums over the entries of a contact / over the contacts of a dof are
      // fire-and-forget LDS adds.
      // y = J x for the rows owned by this lane (x in per-lane slot registers)
      auto mulJ = [&](const T* x, Rows<T>& out) {
        sm.vec[0][lane] = x[0];
        if (isk[0]) sm.keyvec[0][kid[0]] = x[1];
        if (isk[1]) sm.keyvec[0][kid[1]] = x[2];
        if (lane < ncon) { sm.cv[lane][0] = 0; sm.cv[lane][1] = 0; sm.cv[lane][2] = 0; }
        WSYNC();
        out.fr = x[0];
#pragma unroll
        for (int s = 0; s < 3; s++) out.lim[s] = (T)lim_sign[s] * x[s];
        for (int e0 = 0; e0 < nent; e0 += 64) {
          const int e = e0 + lane < nent ? e0 + lane : nent - 1;
          const int m0 = sm.entM[e][0];
          const int ln = RPK_EM_LANE(m0), c = RPK_EM_CON(m0);
          const T j0 = sm.entJ[e][0], j1 = sm.entJ[e][1], j2 = sm.entJ[e][2];
          const T xl = sm.vec[0][ln];
          const T xk = sm.keyvec[0][sm.slotkey[ln >= nl ? ln - nl : 0]];
          const T xv = ln < nl ? xl : xk;
          if (e0 + lane < nent) {
            lds_add(&sm.cv[c][0], j0 * xv); lds_add(&sm.cv[c][1], j1 * xv); lds_add(&sm.cv[c][2], j2 * xv);
          }
        }
        WSYNC();
        T vc[3] = {0, 0, 0};
        if (lane < ncon) { vc[0] = sm.cv[lane][0]; vc[1] = sm.cv[lane][1]; vc[2] = sm.cv[lane][2]; }
        T vn = dot3(con_n, vc), v1 = con_mu * dot3(con_t1, vc), v2 = con_mu * dot3(con_t2, vc);
        out.con[0] = vn + v1; out.con[1] = vn - v1; out.con[2] = vn + v2; out.con[3] = vn - v2;
        WSYNC();
      };
      // forces + active set from jar; returns this lane's share of the constraint cost
      auto update = [&](const Rows<T>& ja, Rows<T>& f) -> T {
        T cost = 0;
        f.fr = 0; fr_quad = 0;
        if (isl && lfloss > 0) {
          T x = ja.fr, rf = lflR * lfloss;
          if (x <= -rf) { f.fr = lfloss; cost += -(T)0.5 * rf * lfloss - lfloss * x; }
          else if (x >= rf) { f.fr = -lfloss; cost += -(T)0.5 * rf * lfloss + lfloss * x; }
          else { f.fr = -lflD * x; fr_quad = 1; cost += (T)0.5 * lflD * x * x; }
        }
#pragma unroll
        for (int s = 0; s < 3; s++) {
          f.lim[s] = 0; lim_act[s] = 0;
          if (lim_sign[s] != 0 && ja.lim[s] < 0) {
            f.lim[s] = -lim_D[s] * ja.lim[s]; lim_act[s] = 1;
            cost += (T)0.5 * lim_D[s] * ja.lim[s] * ja.lim[s];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
          f.con[r] = 0; con_act[r] = 0;
          if (hascon && ja.con[r] < 0) {
            f.con[r] = -con_D * ja.con[r]; con_act[r] = 1;
            cost += (T)0.5 * con_D * ja.con[r] * ja.con[r];
          }
        }
        return cost;
      };
      // out = J^T f
      auto mulJT = [&](const Rows<T>& f, T* out) {
        out[0] = f.fr + (T)lim_sign[0] * f.lim[0];
        out[1] = (T)lim_sign[1] * f.lim[1];
        out[2] = (T)lim_sign[2] * f.lim[2];
        T fn = f.con[0] + f.con[1] + f.con[2] + f.con[3];
        T f1 = con_mu * (f.con[0] - f.con[1]), f2 = con_mu * (f.con[2] - f.con[3]);
        if (lane < ncon) {
#pragma unroll
          for (int k = 0; k < 3; k++) sm.cv[lane][k] = fn * con_n[k] + f1 * con_t1[k] + f2 * con_t2[k];
        }
        sm.jt[lane] = 0;
        WSYNC();
        for (int e0 = 0; e0 < nent; e0 += 64) {
          const int e = e0 + lane < nent ? e0 + lane : nent - 1;
          const int m0 = sm.entM[e][0];
          const int ln = RPK_EM_LANE(m0), c = RPK_EM_CON(m0);
          const T v = sm.entJ[e][0] * sm.cv[c][0] + sm.entJ[e][1] * sm.cv[c][1] + sm.entJ[e][2] * sm.cv[c][2];
          if (e0 + lane < nent) lds_add(&sm.jt[ln], v);
        }
        WSYNC();
        if (isl) out[0] += sm.jt[lane];
#pragma unroll
        for (int s = 0; s < 2; s++) {
          const int ks = isk[s] ? sm.keyslot[kid[s]] : -1;
          const T v = sm.jt[ks >= 0 ? nl + ks : 0];
          if (ks >= 0) out[1 + s] += v;
        }
        WSYNC();
      };
      auto gauss = [&](const T* qa_, const T* Ma_) -> T {
        T g = 0;
#pragma unroll
        for (int s = 0; s < 3; s++) g += (Ma_[s] - qfs[s]) * (qa_[s] - qs[s]);
        return (T)0.5 * g;
      };
      // y = M x with the tree-sparse rows: ancestor terms from this lane's own row
      // (registers), descendant terms from the rows of the next `ndesc` lanes (preorder).
      auto mulM = [&](const T* x, T* out) {
        sm.xs[lane] = x[0];
        WSYNC();
        T y = 0;
        if (isl) {
          // (reads in flight together, terms in the old order: a predicated read per ancestor and a read pair per
          // descendant were up to 35 dependent LDS round trips for a trunk link)
          T xa_[MD + 1];
#pragma unroll
          for (int e = 0; e <= MD; e++) xa_[e] = sm.xs[e <= depth ? anc_at(e) : lane];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e <= MD; e++) { const T t_ = y + Mr[e] * xa_[e]; y = e <= depth ? t_ : y; }
          int j = 1;
          for (; j + 3 <= ndesc; j += 4) {
            T m_[4], x_[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { m_[u] = sm.RM[lane + j + u][depth]; x_[u] = sm.xs[lane + j + u]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; u++) y += m_[u] * x_[u];
          }
          for (; j <= ndesc; j++) y += sm.RM[lane + j][depth] * sm.xs[lane + j];
        }
        out[0] = y;
        out[1] = kM[0] * x[1];
        out[2] = kM[1] * x[2];
        WSYNC();
      };
      auto sub_aref = [&](Rows<T>& r) {
        r.fr -= fr_aref;
#pragma unroll
        for (int s = 0; s < 3; s++) r.lim[s] -= lim_aref[s];
#pragma unroll
        for (int k = 0; k < 4; k++) r.con[k] -= con_aref[k];
      };

      // any constraint rows at all?  (uniform)
      int anyrow = (isl && lfloss > 0) || lim_sign[0] || lim_sign[1] || lim_sign[2] || hascon;
      anyrow = __ballot(anyrow) != 0ull;
      niter_last = 0;
      if (!anyrow) {
#pragma unroll
        for (int s = 0; s < 3; s++) { qa[s] = qs[s]; qfc[s] = 0; }
      } else {
        // warmstart [MJ: warmstart()]
        Rows<T> jtmp;
        mulJ(qs, jtmp); sub_aref(jtmp);
        T cost_smooth = wave_sum(update(jtmp, frc));
#pragma unroll
        for (int s = 0; s < 3; s++) qa[s] = qw[s];
        mulM(qa, Ma);
        mulJ(qa, jar); sub_aref(jar);
        T cost = wave_sum(update(jar, frc) + gauss(qa, Ma));
#ifdef RP_SOLVER_TRACE   // (emulator-only diagnostics)
        if (env == 0 && lane < ncon) printf("  contact %d: D %.6e aref %.6e %.6e jar(smooth) %.6e %.6e %.6e %.6e cross %d A %d B %d\n", lane, (double)con_D, (double)con_aref[0], (double)con_aref[2],
                                (double)jtmp.con[0], (double)jtmp.con[1], (double)jtmp.con[2], (double)jtmp.con[3], con_cross, con_A, con_B);
        if (lane == 0) printf("solver trace: env %d ncon %d nent %d cost_smooth %.10e cost_warm %.10e\n", env, ncon, nent, (double)cost_smooth, (double)cost);
#endif
        if (cost > cost_smooth) {
#pragma unroll
          for (int s = 0; s < 3; s++) { qa[s] = qs[s]; Ma[s] = qfs[s]; }
          jar = jtmp;
          cost = wave_sum(update(jar, frc) + gauss(qa, Ma));
        }
        mulJT(frc, qfc);
        T grad[3];
#pragma unroll
        for (int s = 0; s < 3; s++) grad[s] = Ma[s] - qfs[s] - qfc[s];

        PROF(3);
        const int maxit = S.max_newton;
        for (int iter = 0; iter < maxit; iter++) {
          // ---- H = M + J^T D J on the coupled system (hand dofs + touched keys)
          // key diagonals and gradients to the solver slots
#pragma unroll
          for (int s = 0; s < 2; s++) if (isk[s]) {
            sm.keyvec[0][kid[s]] = kM[s] + (lim_act[1 + s] ? lim_D[1 + s] : (T)0);
            sm.keyvec[1][kid[s]] = grad[1 + s];
          }
          WSYNC();
          const bool isslot = !isl && lane < nsys;
          T rhs = isl ? grad[0] : (T)0, slotdiag = 0;
          if (isslot) {
            int k = sm.slotkey[lane - nl];
            slotdiag = sm.keyvec[0][k];
            rhs = sm.keyvec[1][k];
          }
          const T mydiag_add = (fr_quad ? lflD : (T)0) + (lim_act[0] ? lim_D[0] : (T)0);
          // per-contact 3x3 weight C = sum_r D_r w_r w_r^T with w = n +- mu t
          T Cm[6];  // xx yy zz xy xz yz
          {
            T Dr[4];
#pragma unroll
            for (int r = 0; r < 4; r++) Dr[r] = con_act[r] ? con_D : (T)0;
            T sn = Dr[0] + Dr[1] + Dr[2] + Dr[3];
            T a1 = con_mu * (Dr[0] - Dr[1]), a2 = con_mu * (Dr[2] - Dr[3]);
            T b1 = con_mu * con_mu * (Dr[0] + Dr[1]), b2 = con_mu * con_mu * (Dr[2] + Dr[3]);
            const int ia[6] = {0, 1, 2, 0, 0, 1}, ib[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
            for (int e = 0; e < 6; e++) {
              int a = ia[e], b = ib[e];
              Cm[e] = sn * con_n[a] * con_n[b] + a1 * (con_n[a] * con_t1[b] + con_t1[a] * con_n[b]) +
                      a2 * (con_n[a] * con_t2[b] + con_t2[a] * con_n[b]) + b1 * con_t1[a] * con_t1[b] +
                      b2 * con_t2[a] * con_t2[b];
            }
          }
          T x;
          {
            // rows: mass matrix + per-dof terms (link lanes), key diagonal (slot lanes) ...
            if (lane < ncon) {
#pragma unroll
              for (int e = 0; e < 6; e++) sm.cC[lane][e] = Cm[e];
            }
            if (isl) {
#pragma unroll
              for (int e = 0; e <= MD; e++) if (e <= depth) sm.R[lane][e] = Mr[e] + (e == depth ? mydiag_add : (T)0);
            } else if (isslot) {
#pragma unroll
              for (int e = 0; e <= MD; e++) if (e <= sdepth + 1) sm.R[lane][e] = e == sdepth + 1 ? slotdiag : (T)0;
            }
            WSYNC();
            // ... + J^T C J of every contact.  Entry lane a of contact c holds u = C_c J_a and
            // walks the entries b <= a of its contact (dofs in lane order: b is an ancestor
            // of a, or the same dof): single-chain contacts add u.J_b to row(a)[col(b)] of the
            // tree rows, cross-chain contacts to the packed dense block of the dirty rows (zeroed
            // here; tree_solve adds the Schur complement of the clean rows to it) -- one pass over
            // the entries for both destinations.
            unsigned long long dmx = dirty_mask;
            if (dmx && tri(__popcll(dmx) + 1, 0) > HREC0) { warn |= 32; dmx = 0; }  // cannot hold the block: drop cross terms
            if (dmx) {
              const int nD = __popcll(dmx);
              for (int i = lane; i < tri(nD, 0); i += 64) sm.H[i] = 0;
            }
            auto cidx = [&](int l) -> int { return __popcll(dmx & lanemask_lt(l)); };
            WSYNC();
            {
              for (int e0 = 0; e0 < nent; e0 += 64) {
                const bool valid = e0 + lane < nent;
                const int e = valid ? e0 + lane : nent - 1;
                const int m0 = sm.entM[e][0], m1 = sm.entM[e][1];
                const int ln = RPK_EM_LANE(m0), c = RPK_EM_CON(m0);
                const int base = RPK_EM_BASE(m1), rank = RPK_EM_RANK(m1);
                const T ja0 = sm.entJ[e][0], ja1 = sm.entJ[e][1], ja2 = sm.entJ[e][2];
                const T* C = sm.cC[c];
                const T C0 = C[0], C1 = C[1], C2 = C[2], C3 = C[3], C4 = C[4], C5 = C[5];
                const T u0 = C0 * ja0 + C3 * ja1 + C4 * ja2;
                const T u1 = C3 * ja0 + C1 * ja1 + C5 * ja2;
                const T u2 = C4 * ja0 + C5 * ja1 + C2 * ja2;
                const bool cross = RPK_EM_CROSS(m0) != 0;
                const bool mine = valid && (!cross || dmx != 0);
                const int cia = cross ? cidx(ln) : 0;
                for (int k0 = 0; k0 < maxm; k0 += 4) {
                  // four entries per trip: the twelve LDS reads go out together
                  int mb[4];
                  T val[4];
#pragma unroll
                  for (int u = 0; u < 4; u++) {
                    const int b = base + k0 + u < nent ? base + k0 + u : nent - 1;
                    mb[u] = sm.entM[b][0];
                    val[u] = u0 * sm.entJ[b][0] + u1 * sm.entJ[b][1] + u2 * sm.entJ[b][2];
                  }
#pragma unroll
                  for (int u = 0; u < 4; u++) {
                    if (mine && k0 + u <= rank) {
                      T* dst = cross ? &sm.H[tri(cia, cidx(RPK_EM_LANE(mb[u])))] : &sm.R[ln][RPK_EM_COL(mb[u])];
                      lds_add(dst, val[u]);
                    }
                  }
                }
              }
            }
            WSYNC();
            T Rr[MD + 1];
#pragma unroll
            for (int e = 0; e <= MD; e++) Rr[e] = sm.R[lane][e];
            WSYNC();
            PROF(4);
            x = tree_solve(Rr, rhs, nkt, dmx);
          }
          T search[3];
          search[0] = isl ? -x : (T)0;
          if (!isl && lane < nsys) sm.keyvec[0][sm.slotkey[lane - nl]] = -x;
          WSYNC();
#pragma unroll
          for (int s = 0; s < 2; s++) {
            // branch-free on purpose (selects, no divergent region around the key slots: see
            // the toolchain note in DESIGN.md); kM is 1 for lanes without a key
            const int kk = isk[s] ? kid[s] : 0;
            const int ks_ = sm.keyslot[kk];
            const T via_slot = sm.keyvec[0][kk];
            const T own = -grad[1 + s] / (kM[s] + (lim_act[1 + s] ? lim_D[1 + s] : (T)0));
            search[1 + s] = isk[s] ? (ks_ >= 0 ? via_slot : own) : (T)0;
          }
          WSYNC();
          PROF(5);
          T snorm = N::sqrt(wave_sum(search[0] * search[0] + search[1] * search[1] + search[2] * search[2]));
          if (!(snorm >= RPK_MINVAL)) break;
          T Mv[3];
          mulM(search, Mv);
          Rows<T> jv;
          mulJ(search, jv);
          T g0 = 0, g1 = 0, g2 = 0;
#pragma unroll
          for (int s = 0; s < 3; s++) {
            g0 += (Ma[s] - qfs[s]) * (qa[s] - qs[s]);
            g1 += search[s] * (Ma[s] - qfs[s]);
            g2 += search[s] * Mv[s];
          }
          g0 = (T)0.5 * wave_sum(g0); g1 = wave_sum(g1); g2 = (T)0.5 * wave_sum(g2);
          PROF(6);
          // ---- exact line search [MJ: PrimalSearch]
          // [MJ: PrimalPrepare] along the search direction every row's cost is a quadratic in alpha while
          // the row stays in one zone: q0 + alpha q1 + alpha^2 q2 with q = D (jar^2/2, jar jv, jv^2/2).
          // The coefficients are formed once per Newton iteration; an evaluation then only tests each
          // row's zone at alpha and sums the coefficients of the active rows [MJ: PrimalEval].  Rows a
          // lane does not own get zero coefficients.
          T qfr[3] = {0, 0, 0}, qfl[2] = {0, 0}, frf = -1;   // friction row: quadratic zone, linear zones (+-), zone bound R f
          if (isl && lfloss > 0) {
            frf = lflR * lfloss;
            qfr[0] = (T)0.5 * lflD * jar.fr * jar.fr; qfr[1] = lflD * jar.fr * jv.fr; qfr[2] = (T)0.5 * lflD * jv.fr * jv.fr;
            qfl[0] = -(T)0.5 * frf * lfloss; qfl[1] = lfloss * jar.fr;   // linear zones: qfl[0] -+ qfl[1], slope -+ lfloss jv
          }
          const T frs = lfloss * jv.fr;
          T qlim[3][3], qcon[4][3];
#pragma unroll
          for (int s = 0; s < 3; s++) {
            const T Dp = lim_sign[s] != 0 ? lim_D[s] : (T)0;
            qlim[s][0] = (T)0.5 * Dp * jar.lim[s] * jar.lim[s]; qlim[s][1] = Dp * jar.lim[s] * jv.lim[s];
            qlim[s][2] = (T)0.5 * Dp * jv.lim[s] * jv.lim[s];
          }
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const T Dp = hascon ? con_D : (T)0;
            qcon[r][0] = (T)0.5 * Dp * jar.con[r] * jar.con[r]; qcon[r][1] = Dp * jar.con[r] * jv.con[r];
            qcon[r][2] = (T)0.5 * Dp * jv.con[r] * jv.con[r];
          }
          const bool any_con = ncon > 0;
          // phi(alpha) and its first two derivatives; the one-sided Newton iterations on phi' never
          // read the cost and skip that wave reduction
          auto ls_eval = [&](T alpha, T& d1, T& d2, const bool with_cost) -> T {
#ifdef RPK_LS_COUNTER
            if (S.prof && env == 0 && lane == 0) sm.prof[26] += 1;  // evaluations (debug counter)
#endif
            T s0 = 0, s1 = 0, s2 = 0;
            {
              const T xx = jar.fr + alpha * jv.fr;
              const bool lo_ = xx <= -frf, hi_ = xx >= frf;   // (frf < 0 on lanes without the row: both true, zero coefficients)
              s0 = lo_ ? qfl[0] - qfl[1] : (hi_ ? qfl[0] + qfl[1] : qfr[0]);
              s1 = lo_ ? -frs : (hi_ ? frs : qfr[1]);
              s2 = (lo_ || hi_) ? (T)0 : qfr[2];
              if (frf < 0) { s0 = 0; s1 = 0; }
            }
#pragma unroll
            for (int s = 0; s < 3; s++) {
              const T xx = jar.lim[s] + alpha * jv.lim[s];
              if (xx < 0) { s0 += qlim[s][0]; s1 += qlim[s][1]; s2 += qlim[s][2]; }
            }
            if (any_con) {
#pragma unroll
              for (int r = 0; r < 4; r++) {
                const T xx = jar.con[r] + alpha * jv.con[r];
                if (xx < 0) { s0 += qcon[r][0]; s1 += qcon[r][1]; s2 += qcon[r][2]; }
              }
            }
            s1 = wave_sum(s1) + g1; s2 = wave_sum(s2) + g2;
            d1 = s1 + (T)2 * alpha * s2;
            d2 = (T)2 * s2;
            if (!with_cost) return (T)0;
            s0 = wave_sum(s0) + g0;
            return s0 + alpha * (s1 + alpha * s2);
          };
          T gtol = M.tolerance * M.ls_tolerance * snorm / scale;
          // phi at alpha = 0 needs no row evaluation: phi(0) is the current cost,
          // phi'(0) = grad . search, and phi''(0) = search^T H search = -phi'(0) because the
          // search direction solves H search = -grad.  The first trial point is the full
          // Newton step alpha = 1.
          // (fp64 only: in fp32 the cost carried over from the last update and the cost
          // formula of ls_eval differ by more than the improvements being compared.)
          T f0, h0, c0;
          if constexpr (sizeof(T) == 8) {
            f0 = wave_sum(grad[0] * search[0] + grad[1] * search[1] + grad[2] * search[2]);
            h0 = -f0; c0 = cost;
          } else {
            c0 = ls_eval((T)0, f0, h0, true);
          }
          // [MJ: engine_solver.c PrimalSearch, restated; the oracle's primal_search is the same
          // procedure written sequentially]  All of this is wave-uniform scalar control flow; a
          // point's cost is only reduced where the procedure reads it.  (Measured alternatives with
          // fewer inlined copies of ls_eval -- a state machine around one evaluation site, a to-do
          // list around two -- were 1 ... 4 % slower than this straight transcription.)
          struct LsPnt { T a, c, d0, d1; };
          auto ls_point = [&](T al, const bool with_cost) -> LsPnt {
            LsPnt p; p.a = al; p.c = ls_eval(al, p.d0, p.d1, with_cost); return p;
          };
          T alpha = 0;
          if (h0 > 0) {
            const int max_ls = S.max_ls;
            int evals = 2;                       // (alpha = 0 and the first Newton point)
            const LsPnt p0 = {(T)0, c0, f0, h0};
            LsPnt p1 = ls_point(-f0 / h0, true);  // always one Newton step
            if (p0.c < p1.c) p1 = p0;
            alpha = p1.a;
            if (!(N::abs(p1.d0) < gtol)) {
              const T dir = p1.d0 < 0 ? (T)1 : (T)-1;
              LsPnt p2 = p1;
              bool p2update = false, converged = false;
              while (p1.d0 * dir <= -gtol && evals < max_ls) {   // one-sided search
                p2 = p1; p2update = true;
                p1 = ls_point(p1.a - p1.d0 / p1.d1, false); evals++;
                if (N::abs(p1.d0) < gtol) { converged = true; break; }
              }
              alpha = p1.a;   // converged, or failed to bracket
              if (!converged && evals < max_ls && p2update) {     // bracketed search
                LsPnt p2next = p1;   // (its cost is never read: it is not converged, and the costs of the bracket ends are re-evaluated below)
                LsPnt p1next = ls_point(p1.a - p1.d0 / p1.d1, true); evals++;
                // [MJ: updateBracket] the candidate on p's side of the root whose slope is closest to zero
                auto update_bracket = [&](LsPnt& p, const LsPnt& ca, const LsPnt& cb, const LsPnt& cc, LsPnt& pnext) -> bool {
                  bool moved = false;
                  auto consider = [&](const LsPnt& c) {
                    if ((p.d0 < 0 && c.d0 < 0 && p.d0 < c.d0) || (p.d0 > 0 && c.d0 > 0 && p.d0 > c.d0)) { p = c; moved = true; }
                  };
                  consider(ca); consider(cb); consider(cc);
                  if (moved) { pnext = ls_point(p.a - p.d0 / p.d1, true); evals++; }
                  return moved;
                };
                bool settled = false;
                while (evals < max_ls) {
                  const LsPnt pmid = ls_point((T)0.5 * (p1.a + p2.a), true); evals++;
                  const LsPnt ca = p1next, cb = p2next;   // this round's candidates: ca, cb, pmid
                  T bestc = 0; bool found = false;
                  if (N::abs(ca.d0) < gtol) { found = true; bestc = ca.c; alpha = ca.a; }
                  if (N::abs(cb.d0) < gtol && (!found || cb.c < bestc)) { found = true; bestc = cb.c; alpha = cb.a; }
                  if (N::abs(pmid.d0) < gtol && (!found || pmid.c < bestc)) { found = true; alpha = pmid.a; }
                  if (found) { settled = true; break; }
                  const bool b1 = update_bracket(p1, ca, cb, pmid, p1next);
                  const bool b2 = update_bracket(p2, ca, cb, pmid, p2next);
                  if (!b1 && !b2) { alpha = pmid.a; settled = true; break; }   // numerical accuracy reached
                }
                if (!settled) {   // evaluations exhausted: the better end of the bracket, if it beats alpha = 0
                  T t1, t2;
                  const T c1 = ls_eval(p1.a, t1, t2, true), c2 = ls_eval(p2.a, t1, t2, true);
                  alpha = (c1 <= c2 && c1 < p0.c) ? p1.a : ((c2 <= c1 && c2 < p0.c) ? p2.a : (T)0);
                }
              }
            }
          }
          PROF(7);
          if (!(alpha > 0)) break;
#pragma unroll
          for (int s = 0; s < 3; s++) { qa[s] += alpha * search[s]; Ma[s] += alpha * Mv[s]; }
          jar.fr += alpha * jv.fr;
#pragma unroll
          for (int s = 0; s < 3; s++) jar.lim[s] += alpha * jv.lim[s];
#pragma unroll
          for (int r = 0; r < 4; r++) jar.con[r] += alpha * jv.con[r];
          T oldcost = cost;
          cost = wave_sum(update(jar, frc) + gauss(qa, Ma));
          mulJT(frc, qfc);
          niter_last = iter + 1;
          T gn = 0;
#pragma unroll
          for (int s = 0; s < 3; s++) { grad[s] = Ma[s] - qfs[s] - qfc[s]; gn += grad[s] * grad[s]; }
          gn = N::sqrt(wave_sum(gn));
          PROF(8);
          if (scale * (oldcost - cost) < M.tolerance || scale * gn < M.tolerance) break;
        }
      }
#pragma unroll
      for (int s = 0; s < 3; s++) qw[s] = qa[s];
      // forces of the pyramidal contact rows, for the acceleration-stage sensors (MODE 2)
      if (S.con_force && lane < RPK_NCOUT) {
#pragma unroll
        for (int r = 0; r < 4; r++)
          S.con_force[((size_t)env * RPK_NCOUT + lane) * 4 + r] = (anyrow && lane < ncon) ? frc.con[r] : (T)0;
      }

      PROF(8);
      // ---- Euler with implicit joint damping [MJ: mj_Euler, eulerdamp]
      T qe[3];
      {
        T Rr[MD + 1];
#pragma unroll
        for (int e = 0; e <= MD; e++) Rr[e] = Mr[e] + ((isl && e == depth) ? h * ldamp : (T)0);
        qe[0] = tree_solve(Rr, qfs[0] + qfc[0], 0, 0ull);
      }
#pragma unroll
      for (int s = 0; s < 2; s++) qe[1 + s] = (qfs[1 + s] + qfc[1 + s]) / (kM[s] + h * kdamp[s]);
#pragma unroll
      for (int s = 0; s < 3; s++) { qd[s] += h * qe[s]; q[s] += h * qd[s]; }
      time += h;
      PROF(9);
      // ---- new state
      if (isl) { S.qpos[eo + ldof] = q[0]; S.qvel[eo + ldof] = qd[0]; S.warm[eo + ldof] = qw[0]; }
#pragma unroll
      for (int s = 0; s < 2; s++) if (isk[s]) {
        S.qpos[eo + kdof[s]] = q[1 + s]; S.qvel[eo + kdof[s]] = qd[1 + s]; S.warm[eo + kdof[s]] = qw[1 + s];
      }
    }

    // ======================================================================
    // MODE 0: POSITION + VELOCITY STAGE  (mj_step1)
    // ======================================================================
    if constexpr (MODE != 1) {
    // (the link's own spatial inertia is formed twice from the link frame in LDS -- here for the composite
    // inertias, and again after the collision phase for the bias forces -- with the same arithmetic, hence the
    // same bits: ten doubles carried across the collision phase instead were spilled inside its loops)
    auto link_inertia = [&](T* cin) {
      const T *p_ipos = fresh(M.link_ipos()), *p_inert = fresh(M.link_inertia()), *p_tref = fresh(M.tree_ref());
      const T lm_ = isl ? fresh(M.link_mass())[L] : (T)0;
      T xm_[9], xp_[3], ip_[3], in_[6], tr_[3];
#pragma unroll
      for (int k = 0; k < 9; k++) xm_[k] = isl ? sm.xmat[L][k] : (k % 4 == 0 ? (T)1 : (T)0);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        xp_[k] = isl ? sm.xpos[L][k] : (T)0; ip_[k] = isl ? p_ipos[3 * L + k] : (T)0;
        tr_[k] = isl ? p_tref[3 * ltree + k] : (T)0;
      }
#pragma unroll
      for (int k = 0; k < 6; k++) in_[k] = isl ? p_inert[6 * L + k] : (T)0;
      T t[3], A9[9], dd[3];
      mat_vec(t, xm_, ip_);
#pragma unroll
      for (int k = 0; k < 3; k++) dd[k] = xp_[k] + t[k] - tr_[k];
      // A = xm * Iloc (Iloc symmetric)
      const T I9[9] = {in_[0], in_[3], in_[4], in_[3], in_[1], in_[5], in_[4], in_[5], in_[2]};
      mat_mul(A9, xm_, I9);
      T Iw[6];  // xx yy zz xy xz yz of A * xm^T
      Iw[0] = A9[0] * xm_[0] + A9[1] * xm_[1] + A9[2] * xm_[2];
      Iw[1] = A9[3] * xm_[3] + A9[4] * xm_[4] + A9[5] * xm_[5];
      Iw[2] = A9[6] * xm_[6] + A9[7] * xm_[7] + A9[8] * xm_[8];
      Iw[3] = A9[0] * xm_[3] + A9[1] * xm_[4] + A9[2] * xm_[5];
      Iw[4] = A9[0] * xm_[6] + A9[1] * xm_[7] + A9[2] * xm_[8];
      Iw[5] = A9[3] * xm_[6] + A9[4] * xm_[7] + A9[5] * xm_[8];
      const T d2 = dot3(dd, dd);
      cin[0] = Iw[0] + lm_ * (d2 - dd[0] * dd[0]);
      cin[1] = Iw[1] + lm_ * (d2 - dd[1] * dd[1]);
      cin[2] = Iw[2] + lm_ * (d2 - dd[2] * dd[2]);
      cin[3] = Iw[3] - lm_ * dd[0] * dd[1];
      cin[4] = Iw[4] - lm_ * dd[0] * dd[2];
      cin[5] = Iw[5] - lm_ * dd[1] * dd[2];
      cin[6] = lm_ * dd[0]; cin[7] = lm_ * dd[1]; cin[8] = lm_ * dd[2]; cin[9] = lm_;
    };
    constexpr int NCX = RpCaps<T>::NC;   // contact capacity: RPK_NCL records in LDS, the rest in the env's overflow records
    T* const ovf = B.covf + (size_t)env * (RPK_NC - RPK_NCL) * 12;
    int* const ovi = B.covi + (size_t)env * (RPK_NC - RPK_NCL) * 4;
    // field f of MY contact's record (contact lanes): 0-2 position, 3-5 normal, 6 dist, 7 mu, 8 kterm, 9 B, 10 D
    auto conf = [&](const int f) -> T {
      T v = 0;
      if constexpr (PART != 1) {
        if (NCX <= RPK_NCL || lane < RPK_NCL)
          v = f < 3 ? sm.cpos[lane][f] : (f < 6 ? sm.cn[lane][f - 3] : (f == 6 ? sm.cdist[lane] : sm.cpar[lane][f - 7]));
        else v = ovf[(size_t)(lane - RPK_NCL) * 12 + f];
      }
      return v;
    };
    // ... and its integer fields: 0 link A, 1 link B (or RPK_KEYBASE + key), 2 / 3 model geom ids
    auto coni = [&](const int f) -> int {
      int v = 0;
      if constexpr (PART != 1) {
        if (NCX <= RPK_NCL || lane < RPK_NCL) v = f == 0 ? sm.cA[lane] : (f == 1 ? sm.cB[lane] : (f == 2 ? sm.cgA[lane] : sm.cgB[lane]));
        else v = ovi[(size_t)(lane - RPK_NCL) * 4 + f];
      }
      return v;
    };
    if constexpr (PART != 2) {
    {
      bool bad = !(N::abs(q[0]) < (T)1e10) || !(N::abs(q[1]) < (T)1e10) || !(N::abs(q[2]) < (T)1e10) ||
                 !(N::abs(qd[0]) < (T)1e10) || !(N::abs(qd[1]) < (T)1e10) || !(N::abs(qd[2]) < (T)1e10);
      if (__ballot(bad)) warn |= 1;
    }
    RPK_LOAD_GEOMETRY
    // ---- forward kinematics by tree level [MJ: mj_kinematics]
    T xp[3] = {0, 0, 0}, xm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, axw[3] = {0, 0, 0}, anw[3] = {0, 0, 0};
    for (int d = 0; d < M.maxdepth; d++) {
      if (isl && depth == d) {
        T pp[3] = {0, 0, 0}, pm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (parent >= 0) {
#pragma unroll
          for (int k = 0; k < 3; k++) pp[k] = sm.xpos[parent][k];
#pragma unroll
          for (int k = 0; k < 9; k++) pm[k] = sm.xmat[parent][k];
        }
        T t[3], m0[9];
        mat_vec(t, pm, lpos);
        T pos[3] = {pp[0] + t[0], pp[1] + t[1], pp[2] + t[2]};
        mat_mul(m0, pm, lmat);
        mat_vec(axw, m0, laxis);
        mat_vec(t, m0, lanchor);
        anw[0] = pos[0] + t[0]; anw[1] = pos[1] + t[1]; anw[2] = pos[2] + t[2];
        if (jtype == JNT_SLIDE_) {
#pragma unroll
          for (int k = 0; k < 3; k++) xp[k] = pos[k] + axw[k] * q[0];
#pragma unroll
          for (int k = 0; k < 9; k++) xm[k] = m0[k];
        } else {
          T s, c;
          N::sincos(q[0], &s, &c);
          T oc = (T)1 - c, ax = laxis[0], ay = laxis[1], az = laxis[2];
          T R[9] = {c + oc * ax * ax, oc * ax * ay - s * az, oc * ax * az + s * ay,
                    oc * ax * ay + s * az, c + oc * ay * ay, oc * ay * az - s * ax,
                    oc * ax * az - s * ay, oc * ay * az + s * ax, c + oc * az * az};
          mat_mul(xm, m0, R);
          mat_vec(t, xm, lanchor);
          xp[0] = anw[0] - t[0]; xp[1] = anw[1] - t[1]; xp[2] = anw[2] - t[2];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
          sm.xpos[lane][k] = xp[k];
          if constexpr (PART != 1) { sm.xaxis[lane][k] = axw[k]; sm.xanchor[lane][k] = anw[k]; }   // (the front part hands them over in RpStage::frames)
        }
#pragma unroll
        for (int k = 0; k < 9; k++) sm.xmat[lane][k] = xm[k];
      }
      WSYNC();
    }
    if (MODE == 0 && lane < M.nsite && !(S.stale_outputs && substep == nsub - 1)) {  // site positions of this state
      int sl = M.site_link()[lane];
      T t[3];
      mat_vec(t, sm.xmat[sl], M.site_pos() + 3 * lane);
#pragma unroll
      for (int k = 0; k < 3; k++) S.site_xpos[((size_t)env * M.nsite + lane) * 3 + k] = sm.xpos[sl][k] + t[k];
    }
    if constexpr (PART == 1) {   // the link frames, for the back part (lane-major: coalesced both ways)
      if (isl) {
        T* fr_ = B.frames + (size_t)env * RPK_NFRAME * 64 + lane;
#pragma unroll
        for (int k = 0; k < 3; k++) { fr_[(size_t)k * 64] = xp[k]; fr_[(size_t)(12 + k) * 64] = axw[k]; fr_[(size_t)(15 + k) * 64] = anw[k]; }
#pragma unroll
        for (int k = 0; k < 9; k++) fr_[(size_t)(3 + k) * 64] = xm[k];
      }
    }
    PROF(10);
    // ---- spatial inertia and motion axis about the tree reference point [MJ: mj_comPos]
    {
      T cin[10];
      link_inertia(cin);
      if (jtype == JNT_SLIDE_) {
        cdofr[0] = cdofr[1] = cdofr[2] = 0;
        cdofr[3] = axw[0]; cdofr[4] = axw[1]; cdofr[5] = axw[2];
      } else {
        T off[3] = {tref[0] - anw[0], tref[1] - anw[1], tref[2] - anw[2]};
        cdofr[0] = axw[0]; cdofr[1] = axw[1]; cdofr[2] = axw[2];
        cross3(cdofr + 3, axw, off);
      }
      if (isl) {
#pragma unroll
        for (int k = 0; k < 6; k++) sm.cdof[lane][k] = cdofr[k];
#pragma unroll
        for (int k = 0; k < 10; k++) sm.acc[lane][k] = cin[k];
      }
    }
    WSYNC();
    // ---- composite inertias, children -> parent by level and sibling rank [MJ: mj_crb]
    // (siblings add into their parent's cell together: fire-and-forget LDS adds, at most five lanes on a cell at
    // the finger roots, instead of one read-modify-write round per sibling rank)
    for (int d = M.maxdepth - 1; d >= 1; d--) {
      if (isl && depth == d) {
#pragma unroll
        for (int k = 0; k < 10; k++) lds_add(&sm.acc[parent][k], sm.acc[lane][k]);
      }
      WSYNC();
    }
    if (isl) {
      T crb[10], buf[6];
#pragma unroll
      for (int k = 0; k < 10; k++) crb[k] = sm.acc[lane][k];
      mul_inert(buf, crb, cdofr);
#pragma unroll
      for (int k = 0; k < MD; k++) {
        if (k <= depth) {
          int a = anc_at(k);
          T v = dot6(sm.cdof[a], buf);
          if (a == lane) v += larm;
          Mr[k] = v;
        }
      }
      // handed over right away: the rows are not needed again in this kernel, and the collision
      // phase that follows is the register-hungriest part of it
      if constexpr (MODE == 0) {
#pragma unroll
        for (int e = 0; e <= MD; e++) B.RM[((size_t)env * RPK_NLX(MD) + lane) * (MD + 1) + e] = Mr[e];
      }
    }
    PROF(11);
    // ---- key poses, geom centres
#pragma unroll
    for (int s = 0; s < 2; s++) {
      if (isk[s]) {
        N::sincos(q[1 + s], &ksin[s], &kcos[s]);
        if constexpr (PART != 1) { sm.kq[kid[s]] = q[1 + s]; sm.keyslot[kid[s]] = -1; }
      }
    }
    WSYNC();
    // (the model's per-geom constants first, all reads in flight together: fetched where they were used they were a dozen
    // dependent trips to L2, most of this phase's time)
    const bool isg = lane < M.ngeom;
    const int G_ = isg ? lane : 0;
    const int ggl = M.geom_link()[G_], gty = M.geom_type()[G_];
    T gp[3] = {M.geom_pos()[3 * G_], M.geom_pos()[3 * G_ + 1], M.geom_pos()[3 * G_ + 2]};
    T gmat[9];
#pragma unroll
    for (int k = 0; k < 9; k++) gmat[k] = M.geom_mat()[9 * G_ + k];
    const T gbc0 = M.geom_bcap()[2 * G_], gbc1 = M.geom_bcap()[2 * G_ + 1];
    const T gsz[3] = {M.geom_size()[3 * G_], M.geom_size()[3 * G_ + 1], M.geom_size()[3 * G_ + 2]};
    // (... and the broad phase's: bounding radius, pair mask, key-capable flag)
    const T grb_ = M.geom_rbound()[G_];
    const unsigned gpm0_ = (unsigned)M.geom_pairmask()[2 * G_], gpm1_ = (unsigned)M.geom_pairmask()[2 * G_ + 1];
    const auto gkc_ = M.geom_iskeycap()[G_];
    __builtin_amdgcn_sched_barrier(0);
    if (isg) {
      const int gl = ggl;
      if (gl >= 0) {
        T t[3];
        mat_vec(t, sm.xmat[gl], gp);
        gp[0] = sm.xpos[gl][0] + t[0]; gp[1] = sm.xpos[gl][1] + t[1]; gp[2] = sm.xpos[gl][2] + t[2];
      }
      sm.gpos[lane][0] = gp[0]; sm.gpos[lane][1] = gp[1]; sm.gpos[lane][2] = gp[2];
      // capsule axis (third column of the world geom frame) for the segment prefilter
      T az[3] = {gmat[2], gmat[5], gmat[8]};
      if (gl >= 0) { T t[3]; mat_vec(t, sm.xmat[gl], az); az[0] = t[0]; az[1] = t[1]; az[2] = t[2]; }
      // (round 6) ... for EVERY geom: its bounding capsule about the geom's z axis (model/engine_tables.py: a capsule's own
      // half length and radius; the tightest capsule around a hull's vertices / a cylinder; a box keeps (0, bounding
      // radius), its culls are the separating-axis tests below).  Until now a hull entered the segment test as its bounding
      // SPHERE, and a fingertip hull next to the neighbouring finger's capsule passed it most of the time: 13 of the 17
      // hull candidates per env and mj_step on the replay, each of them two or three trips of the portal refinement.
      sm.gax[lane][0] = (float)az[0]; sm.gax[lane][1] = (float)az[1]; sm.gax[lane][2] = (float)az[2];
      sm.gax[lane][3] = (float)gbc0;
      sm.grr[lane] = (float)gbc1;
    }
    // geoms are sorted capsules first: box b is geom ncap + b; `boxmask`: the geoms that are boxes (not hulls)
    const unsigned long long boxmask = __ballot(isg && gty == GEOM_BOX_);
    const int ncap = __popcll(__ballot(isg && gty == GEOM_CAPSULE_));
    if (lane >= ncap && isg && lane - ncap < RPK_NBOXF) {
      const int gl = ggl;
      T mw[9];
      if (gl >= 0) mat_mul(mw, sm.xmat[gl], gmat);
      else {
#pragma unroll
        for (int k = 0; k < 9; k++) mw[k] = gmat[k];
      }
      float* gb = sm.gbox[lane - ncap];
#pragma unroll
      for (int k = 0; k < 9; k++) gb[k] = (float)mw[k];
#pragma unroll
      for (int k = 0; k < 3; k++) gb[9 + k] = (float)gsz[k];
    }
    WSYNC();

    PROF(18);
    // ---- broad phase + narrow phase, streamed through a bounded work list
    // [MJ: mj_collision].  Broad phase: lane g holds geom g and collects, as a 64-bit mask,
    // the partners j > g whose bounding spheres touch (63 readlane broadcasts, no memory,
    // no per-pair bookkeeping); lane k holds keys k, k+64 and collects the hand capsules
    // near them.  The culling runs in fp32 with inflated radii: it only has to be a
    // superset, the narrow phase decides in working precision.  The masks are then
    // drained one bit per lane per round into the work list; whenever 64 candidates are
    // pending they are narrow-phased, so the list never overflows whatever the pose.
    int nwork = 0;
    ncon = 0;
    {
    const bool isg = lane < M.ngeom;
    // (read unconditionally and selected: a predicated LDS read per value was eight dependent round trips)
    const T gq0_ = sm.gpos[lane][0], gq1_ = sm.gpos[lane][1], gq2_ = sm.gpos[lane][2];
    const float ga0_ = sm.gax[lane][0], ga1_ = sm.gax[lane][1], ga2_ = sm.gax[lane][2], ga3_ = sm.gax[lane][3], grr_ = sm.grr[lane];
    __builtin_amdgcn_sched_barrier(0);
    const float fcx = isg ? (float)gq0_ : 0.f, fcy = isg ? (float)gq1_ : 0.f,
                fcz = isg ? (float)gq2_ : 0.f;
    const float frb = isg ? (float)grb_ * 1.0001f + 1e-6f : 0.f;
    const unsigned long long gpm = isg ? (((unsigned long long)gpm1_ << 32) | gpm0_) : 0ull;
    const bool gkc = isg && gkc_ != 0;
    const float fax = isg ? ga0_ : 0.f, fay = isg ? ga1_ : 0.f, faz = isg ? ga2_ : 0.f;
    const float fhl = isg ? ga3_ : 0.f, frr = isg ? grr_ : 0.f;
    unsigned hitlo = 0, hithi = 0;
#pragma unroll
    for (int j0 = 0; j0 < 64; j0 += 8) {
      if (j0 < M.ngeom) {
#pragma unroll
        for (int jj = 0; jj < 8; jj++) {
          const int j = j0 + jj;
          const float dx = fcx - bcast(fcx, j), dy = fcy - bcast(fcy, j), dz = fcz - bcast(fcz, j);
          const float rr = frb + bcast(frb, j);
          const bool hit = dx * dx + dy * dy + dz * dz <= rr * rr;
          if (j < 32) hitlo |= hit ? (1u << j) : 0u; else hithi |= hit ? (1u << (j - 32)) : 0u;
        }
      }
    }
    // allowed pairs this lane owns (engine_tables.py deals every static pair to one of its two lanes)
    unsigned long long remA = (((unsigned long long)hithi << 32) | hitlo) & gpm;
    // keys: every geom that reaches down to the keyboard (a lane) walks the keys whose y-interval can reach it.
    // (The loop used to run the other way round -- the key lanes over the ~40 near geoms, one or two per trip,
    // 32 k cycles per mj_step for 8.5 candidates.  Keys lie along y in index order, so the keys a geom can touch
    // are an index window around (y - y_first) / pitch; its half-width covers the keys' half-widths and their
    // worst deviation from the even grid, both taken from the model at run time: a superset for any layout.)
    // (Capsule builds.  The hull builds keep the key lanes' loop over the near geoms: with the window walk their
    // register allocation spilled inside the CRB / RNE level loops -- 207 spilled VGPRs against 144 -- and the
    // step got 2.8 % slower although candidate generation itself went from 40 k to 26 k cycles.)
    int kcount = 0;
    // (ROUND 5: the window walk in every build.  The split stage's front part has no narrow phase to share registers
    // with (160 VGPRs, no scratch) and takes the walk's 17 k cycles per mj_step; and since both ways of running the stage
    // must emit the same candidates in the same ORDER to stay bit-identical, the one-kernel hull builds follow (round 4
    // measured them 0.6 % slower with it).  RPK_MESH_KEYLANES = the old loop, for experiments.)
#ifdef RPK_MESH_KEYLANES
    constexpr bool KEYLANES = MESH != 0;
#else
    constexpr bool KEYLANES = false;
#endif
    unsigned long long remK0 = 0, remK1 = 0;   // hull builds: the capsules near this lane's two keys
    {
      // extents along world x, y, z of MY geom: a capsule's own axis-aligned extents |axis| * half length +
      // radius, the oriented box's for boxes (a palm box hovering over the keyboard is no candidate), the
      // bounding radius otherwise
      float gex = frb, gey = frb, gez = frb;
      {
        const int bi_ = lane - ncap;
        if (isg && bi_ >= 0 && bi_ < RPK_NBOXF) {
          const float* gb_ = sm.gbox[bi_];
          gex = fabsf(gb_[0]) * gb_[9] + fabsf(gb_[1]) * gb_[10] + fabsf(gb_[2]) * gb_[11] + 1e-4f;
          gey = fabsf(gb_[3]) * gb_[9] + fabsf(gb_[4]) * gb_[10] + fabsf(gb_[5]) * gb_[11] + 1e-4f;
          gez = fabsf(gb_[6]) * gb_[9] + fabsf(gb_[7]) * gb_[10] + fabsf(gb_[8]) * gb_[11] + 1e-4f;
        } else if (isg && lane < ncap && gty == GEOM_CAPSULE_) {
          // (fp32 with an allowance of 0.1 mm + 1e-4 of the size: the cull only has to be a superset)
          gex = fminf(frb, (fabsf(fax) * fhl + frr) * 1.0001f + 1e-4f);
          gey = fminf(frb, (fabsf(fay) * fhl + frr) * 1.0001f + 1e-4f);
          gez = fminf(frb, (fabsf(faz) * fhl + frr) * 1.0001f + 1e-4f);
        }
      }
      const bool near_me = nk > 0 && gkc && (fcz - gez <= (float)M.key_zmax);
      if constexpr (KEYLANES) {
        unsigned long long near_mask = __ballot(near_me);
        float kx[2], kz[2], kpx[2], kpy[2], ktop[2], khx_[2], khy_[2], krb_[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
          // key box centre = anchor + R_y(q) (hx,0,0)
          kx[s] = (float)(kpos[s][0] - khalf[s][0] + khalf[s][0] * kcos[s]);
          kz[s] = (float)(kpos[s][2] - khalf[s][0] * ksin[s]);
          kpx[s] = (float)kpos[s][0]; kpy[s] = (float)kpos[s][1];
          ktop[s] = (float)(kpos[s][2] + khalf[s][2]) + 0.01f;
          khx_[s] = (float)khalf[s][0] + 0.01f; khy_[s] = (float)khalf[s][1];
          krb_[s] = (float)krb[s] * 1.0001f + 1e-6f;
        }
        // two near geoms per trip (their broadcasts and tests interleave)
        while (near_mask) {
          const int g0 = __ffsll((long long)near_mask) - 1;
          near_mask &= near_mask - 1;
          const bool two = near_mask != 0ull;
          const int g1 = two ? __ffsll((long long)near_mask) - 1 : g0;
          near_mask &= near_mask - 1;   // (0 & -1 = 0)
#pragma unroll
          for (int u = 0; u < 2; u++) {
            const int g = u == 0 ? g0 : g1;
            const float cx = bcast(fcx, g), cy = bcast(fcy, g), cz = bcast(fcz, g), rb = bcast(frb, g);
            const float ex_ = bcast(gex, g), ey_ = bcast(gey, g), ez_ = bcast(gez, g);
#pragma unroll
            for (int s = 0; s < 2; s++) {
              const float dx = kx[s] - cx, dy = kpy[s] - cy, dz = kz[s] - cz, rr = rb + krb_[s];
              // bounding spheres, then a conservative box test (the key only rotates about
              // y, so its y-extent is exact; x/z get a 1 cm allowance)
              const bool hit = (u == 0 || two) && isk[s] && dx * dx + dy * dy + dz * dz <= rr * rr && fabsf(dy) <= khy_[s] + ey_ &&
                               fabsf(cx - kpx[s]) <= khx_[s] + ex_ && cz - ez_ <= ktop[s];
              if (s == 0) remK0 |= hit ? (1ull << g) : 0ull; else remK1 |= hit ? (1ull << g) : 0ull;
            }
          }
        }
      } else {
      float kx[2], kz[2], kpy[2], khy_[2];
#pragma unroll
      for (int s = 0; s < 2; s++) {
        // key box centre = anchor + R_y(q) (hx,0,0)
        kx[s] = (float)(kpos[s][0] - khalf[s][0] + khalf[s][0] * kcos[s]);
        kz[s] = (float)(kpos[s][2] - khalf[s][0] * ksin[s]);
        kpy[s] = (float)kpos[s][1]; khy_[s] = (float)khalf[s][1];
      }
      // the even grid through the first and the last key, and how far a key's y-interval can reach from its node
      const float ky0 = bcast(kpy[0], 0);
      const float kyl = nk > 64 ? bcast(kpy[1], (nk - 1) & 63) : bcast(kpy[0], nk > 0 ? nk - 1 : 0);
      const float kpitch = nk > 1 ? (kyl - ky0) / (float)(nk - 1) : 1.f;
      float dev = 0.f;
#pragma unroll
      for (int s = 0; s < 2; s++) if (isk[s]) dev = fmaxf(dev, fabsf(kpy[s] - (ky0 + (float)kid[s] * kpitch)) + khy_[s]);
      const float kslack = __int_as_float(wave_max(__float_as_int(dev))) * 1.0001f + 1e-5f;   // (non-negative floats order like ints)
      const float apitch = fabsf(kpitch) > 1e-9f ? fabsf(kpitch) : 1e-9f;
      int klo = 0, kn = 0;
      if (near_me) {
        const float c = (fcy - ky0) / kpitch, w = (gey + kslack) / apitch + 1e-3f;
        const float lo_ = floorf(c - w), hi_ = ceilf(c + w);
        const int lo = lo_ < 0.f ? 0 : (lo_ > (float)(nk - 1) ? nk : (int)lo_), hi = hi_ < 0.f ? -1 : (hi_ > (float)(nk - 1) ? nk - 1 : (int)hi_);
        klo = lo; kn = hi - lo + 1 > 0 ? hi - lo + 1 : 0;
      }
      const int kmax = wave_max(kn);
      for (int r = 0; r < kmax; r++) {
        const bool on = r < kn;
        const int k = on ? klo + r : 0;
        const int src = k & 63;
        const bool s1 = k >= 64;
        // the key's moving box centre from the lane that holds the key (ds_bpermute); what does not move, from the
        // model tables (the same fp32 values the key lanes used to hold)
        const float tx0 = __shfl(kx[0], src, 64), tx1 = __shfl(kx[1], src, 64), tz0 = __shfl(kz[0], src, 64), tz1 = __shfl(kz[1], src, 64);
        const float kx_ = s1 ? tx1 : tx0, kz_ = s1 ? tz1 : tz0;
        const int kk = k < nk ? k : 0;
        const T *kp_ = M.key_pos() + 3 * kk, *kh_ = M.key_half() + 3 * kk;
        // (the key's seven constants in one trip to L2: read where they were used they were three)
        const T kp0_ = kp_[0], kp1_ = kp_[1], kp2_ = kp_[2], kh0_ = kh_[0], kh1_ = kh_[1], kh2_ = kh_[2], krbr_ = M.key_rbound()[kk];
        __builtin_amdgcn_sched_barrier(0);
        const float kpx_ = (float)kp0_, kpy_ = (float)kp1_, ktop_ = (float)(kp2_ + kh2_) + 0.01f;
        const float khx2 = (float)kh0_ + 0.01f, khy2 = (float)kh1_, krb2 = (float)krbr_ * 1.0001f + 1e-6f;
        const float dx = kx_ - fcx, dy = kpy_ - fcy, dz = kz_ - fcz, rr = frb + krb2;
        // bounding spheres, then a conservative box test (the key only rotates about
        // y, so its y-extent is exact; x/z get a 1 cm allowance)
        const bool hit = on && k < nk && dx * dx + dy * dy + dz * dz <= rr * rr && fabsf(dy) <= khy2 + gey &&
                         fabsf(fcx - kpx_) <= khx2 + gex && fcz - gez <= ktop_;
        const unsigned long long hm = __ballot(hit);
        const int at = kcount + __popcll(hm & lanemask_lt(lane));
        if (hit && at < RPK_KLIST) sm.klist[at] = (unsigned short)((lane << 7) | k);
        kcount += __popcll(hm);
      }
      if (kcount > RPK_KLIST) { warn |= 16; kcount = RPK_KLIST; }   // (RP_WARN_WORK_FULL: more geom-key candidates than the list holds)
      }
    }
    PROF(12);
#ifndef RPK_MARK
    if (S.prof && env == 0) {
      int ca = (int)wave_sum((float)__popcll(remA)), ck = KEYLANES ? (int)wave_sum((float)(__popcll(remK0) + __popcll(remK1))) : kcount;
      if (lane == 0) { sm.prof[28] += ca; sm.prof[29] += ck; }
    }
#endif
    int gen_phase = 0;
    int lpos = 0, lcount = 0;   // compacted geom-geom candidates: next / number of list entries
    int kpos_ = 0;              // geom-key candidates: next list entry
    while (true) {
      // ---- drain rounds, in a loop of their own (until 64 candidates are pending or the masks are empty): the
      // narrow phase below is the register-hungriest part of this kernel, and in one loop with it the
      // allocator spilled the drain loop's own variables -- every round then paid scratch round trips
      // (measured: 34 k cycles per mj_step in the capsule builds, 163 k in the hull builds)
      while (gen_phase < (KEYLANES ? 3 : 2) && (PART == 1 || nwork < 64)) {
      // ---- one drain round: every lane contributes at most one candidate
      {
        // Geom-geom candidates are COMPACTED first: the sphere-overlap hits sit unevenly in the lanes (a palm box
        // touches a dozen spheres, most capsules two or three), and one candidate per lane and round took as many
        // rounds as the busiest lane has hits (measured 12.5 per mj_step for 185 candidates).  Every lane writes
        // its hits (owner << 6 | partner) at its prefix offset into a flat list, and a round takes the next 64
        // entries whoever owns them: three rounds.  (A list that does not hold all hits is refilled.)
        if (gen_phase == 0 && lpos >= lcount) {
          if (__ballot(remA != 0ull) == 0ull) { gen_phase = 1; continue; }
          const int myc = __popcll(remA);
          int incl = myc;
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
          int k = incl - myc;
          const int total = bcast(incl, 63);
          unsigned long long m_ = remA;
          while (__ballot(m_ != 0ull && k < RPK_GLIST) != 0ull) {
            if (m_ != 0ull && k < RPK_GLIST) {
              const int b_ = __ffsll((long long)m_) - 1;
              m_ &= m_ - 1;
              sm.glist[k] = (unsigned short)((lane << 6) | b_);
              k++;
            }
          }
          remA = m_;   // (what did not fit stays for the next refill)
          lcount = total < RPK_GLIST ? total : RPK_GLIST;
          lpos = 0;
          WSYNC();
        }
        const unsigned long long rem = gen_phase == 1 ? remK0 : remK1;   // (hull builds)
        bool has = gen_phase == 0 ? lpos + lane < lcount : (KEYLANES ? rem != 0ull : kpos_ + lane < kcount);
#ifndef RPK_MARK
        if (S.prof && env == 0 && lane == 0) sm.prof[26] += 1;   // (diagnostic: drain rounds)
#endif
        if (__ballot(has) == 0ull) gen_phase++;
        else {
          // the pair of this lane, from the lists: geom-geom (a = the lane that found it), then geom-key
          int a = lane, bit = 0;
          if (gen_phase == 0) {
            const int item = has ? (int)sm.glist[lpos + lane] : 0;
            a = item >> 6; bit = item & 63;
            lpos += 64;
          } else if constexpr (KEYLANES) {
            bit = has ? __ffsll((long long)rem) - 1 : 0;
            const unsigned long long rest = rem & (rem - 1);
            if (gen_phase == 1) remK0 = rest; else remK1 = rest;
            a = gen_phase == 1 ? kid[0] : kid[1];   // (key, geom)
          } else {
            const int item = has ? (int)sm.klist[kpos_ + lane] : 0;
            a = item & 127; bit = item >> 7;   // (key, geom)
            kpos_ += 64;
          }
          if (gen_phase == 0) {
            // (geom a's data from LDS: the same fp32 values its lane held in registers)
            const float fcx = (float)sm.gpos[a][0], fcy = (float)sm.gpos[a][1], fcz = (float)sm.gpos[a][2];
            const float fax = sm.gax[a][0], fay = sm.gax[a][1], faz = sm.gax[a][2], fhl = sm.gax[a][3], frr = sm.grr[a];
            // segment-segment distance (boxes: centre point with their bounding radius)
            // against the sum of radii, fp32 with a 0.1 mm allowance: most sphere-overlap
            // candidates between neighbouring phalanges end here
            const float bx = sm.gax[bit][0], by = sm.gax[bit][1], bz = sm.gax[bit][2], l2 = sm.gax[bit][3];
            const float r2 = sm.grr[bit];
            const float rx = fcx - (float)sm.gpos[bit][0], ry = fcy - (float)sm.gpos[bit][1],
                        rz = fcz - (float)sm.gpos[bit][2];
            const float bb = fax * bx + fay * by + faz * bz;
            const float cc = fax * rx + fay * ry + faz * rz, ff = bx * rx + by * ry + bz * rz;
            const float den = 1.f - bb * bb;
            float x1 = den > 1e-6f ? fminf(fhl, fmaxf(-fhl, (bb * ff - cc) / den)) : 0.f;
            float x2 = bb * x1 + ff;
            if (x2 > l2 || x2 < -l2) {
              x2 = fminf(l2, fmaxf(-l2, x2));
              x1 = fminf(fhl, fmaxf(-fhl, bb * x2 - cc));
            }
            const float ex = rx + fax * x1 - bx * x2, ey = ry + fay * x1 - by * x2, ez = rz + faz * x1 - bz * x2;
            const float reach = frr + r2 + 1e-4f;
            has = has && (ex * ex + ey * ey + ez * ez <= reach * reach);
            // boxes: separating-axis test on the three box axes (capsule = segment + radius)
            const int bi = bit - ncap;
            const bool bbox = bi >= 0 && bi < RPK_NBOXF;
            const float* gb = sm.gbox[bbox ? bi : 0];
            if (bbox) {
              const float reach2 = frr + 1e-4f;
              bool sep = false;
              float akv[3];
#pragma unroll
              for (int k = 0; k < 3; k++) {
                const float ck = gb[k] * rx + gb[3 + k] * ry + gb[6 + k] * rz;
                const float ak = gb[k] * fax + gb[3 + k] * fay + gb[6 + k] * faz;
                akv[k] = fabsf(ak);
                sep = sep || (fabsf(ck) - fhl * fabsf(ak) > gb[9 + k] + reach2);
              }
              // (round 5) ... the segment's own axis and the three cross axes d x b_k: with them the test is the complete
              // separating-axis test of a segment against the box, inflated by the radius (a superset still: rounded
              // corners only make the true shape smaller).  A fingertip hull's bounding box against the NEIGHBOURING
              // finger's capsule passed the three box axes most of the time: 13 hull candidates per env and mj_step.
              {
                const float cd = fax * rx + fay * ry + faz * rz;
                sep = sep || (fabsf(cd) - fhl > gb[9] * akv[0] + gb[10] * akv[1] + gb[11] * akv[2] + reach2 + 1e-6f);
                const float wx = ry * faz - rz * fay, wy = rz * fax - rx * faz, wz = rx * fay - ry * fax;   // r x d
#pragma unroll
                for (int k = 0; k < 3; k++) {
                  const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
                  const float tk = gb[k] * wx + gb[3 + k] * wy + gb[6 + k] * wz;   // r . (d x b_k)
                  const float ln = sqrtf(fmaxf(1.f - akv[k] * akv[k], 0.f));       // |d x b_k|
                  sep = sep || (fabsf(tk) > gb[9 + k1] * akv[k2] + gb[9 + k2] * akv[k1] + reach2 * ln + 1e-5f);
                }
              }
              has = has && !sep;
            }
            {
              // box against box: the full 15-axis separating-axis test in fp32 (0.1 mm allowance), so
              // that the fp64 box-box routine only ever runs for boxes that really touch.  (The vote is
              // taken by the whole wave, in uniform control flow; the test itself runs on the box-box lanes.)
              // (true boxes only: a pair with a hull goes on with the three-axis test above -- the portal
              // refinement drops a separated pair within its first two supports, and the fifteen-axis test is
              // paid by the whole wave in every round that holds such a pair)
              const int ai = a - ncap;
              const bool abox = bbox && ai >= 0 && ai < RPK_NBOXF && ((boxmask >> a) & 1) && ((boxmask >> bit) & 1);
              if (__builtin_amdgcn_ballot_w64(abox && has) != 0ull && abox) {
                const float* ga_ = sm.gbox[ai];
                float Rf[3][3], Qf[3][3], tf[3];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                  tf[i] = -(ga_[i] * rx + ga_[3 + i] * ry + ga_[6 + i] * rz);   // (B centre - A centre) in A's frame
#pragma unroll
                  for (int j = 0; j < 3; j++) {
                    Rf[i][j] = ga_[i] * gb[j] + ga_[3 + i] * gb[3 + j] + ga_[6 + i] * gb[6 + j];
                    Qf[i][j] = fabsf(Rf[i][j]) + 1e-6f;
                  }
                }
                bool sp = false;
#pragma unroll
                for (int i = 0; i < 3; i++)
                  sp = sp || fabsf(tf[i]) > ga_[9 + i] + gb[9] * Qf[i][0] + gb[10] * Qf[i][1] + gb[11] * Qf[i][2] + 1e-4f;
#pragma unroll
                for (int j = 0; j < 3; j++)
                  sp = sp || fabsf(tf[0] * Rf[0][j] + tf[1] * Rf[1][j] + tf[2] * Rf[2][j]) >
                                 gb[9 + j] + ga_[9] * Qf[0][j] + ga_[10] * Qf[1][j] + ga_[11] * Qf[2][j] + 1e-4f;
#pragma unroll
                for (int i = 0; i < 3; i++) {
#pragma unroll
                  for (int j = 0; j < 3; j++) {
                    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
                    const float tl = fabsf(tf[i2] * Rf[i1][j] - tf[i1] * Rf[i2][j]);
                    const float ra = ga_[9 + i1] * Qf[i2][j] + ga_[9 + i2] * Qf[i1][j];
                    const float rb = gb[9 + j1] * Qf[i][j2] + gb[9 + j2] * Qf[i][j1];
                    sp = sp || tl > ra + rb + 1e-4f;   // (unnormalised axis: both sides scale alike; the
                  }                                    //  allowance only makes the test more conservative)
                }
                has = has && !sp;
              }
            }
          } else if constexpr (MESH != 0) {
            // (ROUND 5) hull against key: the hull's bounding box (the box of its vertices in the geom frame: it contains
            // the hull) against the key's box, all fifteen separating axes in fp32 with the 0.1 mm allowance of the other
            // culls.  Until now such a pair went on with the window walk's world-axis test only, and a fingertip hovering
            // a few millimetres over the keyboard was a candidate for every key under it: measured 21 hull pairs per env
            // and mj_step on the replay against ~2 hull contacts -- each of them two or three trips of the portal
            // refinement for the whole wave, and 1300 of the pooled narrow phase's 1900 waves per launch.
            const int bi = bit - ncap;
            const bool hb = has && bit >= ncap && bi < RPK_NBOXF && !((boxmask >> bit) & 1);
            if (__builtin_amdgcn_ballot_w64(hb) != 0ull) {
              const int src = a & 63;
              const float c0 = (float)__shfl(kcos[0], src, 64), c1 = (float)__shfl(kcos[1], src, 64);
              const float s0 = (float)__shfl(ksin[0], src, 64), s1 = (float)__shfl(ksin[1], src, 64);
              if (hb) {
                const float kc = a >= 64 ? c1 : c0, ks = a >= 64 ? s1 : s0;
                const T *kp_ = M.key_pos() + 3 * a, *kh_ = M.key_half() + 3 * a;
                const float hx = (float)kh_[0], hy = (float)kh_[1], hz = (float)kh_[2];
                // key box: centre = anchor + R_y(q) (hx, 0, 0); axes = the columns of R_y(q)
                const float bcx = (float)kp_[0] - hx + hx * kc, bcy = (float)kp_[1], bcz = (float)kp_[2] - hx * ks;
                const float kb[12] = {kc, 0.f, ks, 0.f, 1.f, 0.f, -ks, 0.f, kc, hx, hy, hz};
                const float* ga_ = sm.gbox[bi];
                const float rx = (float)sm.gpos[bit][0] - bcx, ry = (float)sm.gpos[bit][1] - bcy, rz = (float)sm.gpos[bit][2] - bcz;   // (A centre - B centre)
                float Rf[3][3], Qf[3][3], tf[3];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                  tf[i] = -(ga_[i] * rx + ga_[3 + i] * ry + ga_[6 + i] * rz);   // (B centre - A centre) in A's frame
#pragma unroll
                  for (int j = 0; j < 3; j++) {
                    Rf[i][j] = ga_[i] * kb[j] + ga_[3 + i] * kb[3 + j] + ga_[6 + i] * kb[6 + j];
                    Qf[i][j] = fabsf(Rf[i][j]) + 1e-6f;
                  }
                }
                bool sp = false;
#pragma unroll
                for (int i = 0; i < 3; i++)
                  sp = sp || fabsf(tf[i]) > ga_[9 + i] + kb[9] * Qf[i][0] + kb[10] * Qf[i][1] + kb[11] * Qf[i][2] + 1e-4f;
#pragma unroll
                for (int j = 0; j < 3; j++)
                  sp = sp || fabsf(tf[0] * Rf[0][j] + tf[1] * Rf[1][j] + tf[2] * Rf[2][j]) >
                                 kb[9 + j] + ga_[9] * Qf[0][j] + ga_[10] * Qf[1][j] + ga_[11] * Qf[2][j] + 1e-4f;
#pragma unroll
                for (int i = 0; i < 3; i++) {
#pragma unroll
                  for (int j = 0; j < 3; j++) {
                    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
                    const float tl = fabsf(tf[i2] * Rf[i1][j] - tf[i1] * Rf[i2][j]);
                    const float ra = ga_[9 + i1] * Qf[i2][j] + ga_[9 + i2] * Qf[i1][j];
                    const float rb = kb[9 + j1] * Qf[i][j2] + kb[9 + j2] * Qf[i][j1];
                    sp = sp || tl > ra + rb + 1e-4f;
                  }
                }
                has = has && !sp;
              }
            }
          }
          const unsigned long long mk = __ballot(has);
#ifndef RPK_MARK
          if (S.prof && env == 0 && gen_phase == 0) {
            const int nb = __popcll(__ballot(has && M.geom_type()[bit] != GEOM_CAPSULE_));
            if (lane == 0) sm.prof[27] += nb;
          }
#endif
          const int idx = nwork + __popcll(mk & lanemask_lt(lane));
          if constexpr (PART == 1) {
            if (has && idx < RPK_NCAND) {
              if (gen_phase == 0) sm.clist[idx] = (a < bit ? a : bit) | ((a < bit ? bit : a) << 16);
              else sm.clist[idx] = bit | ((RPK_KEYBASE + a) << 16);
            }
          } else if constexpr (PART != 1) { if (has) {
            // (a static pair is owned by either of its lanes: geom 1 of the pair is the lower one)
            if (gen_phase == 0) { sm.work[idx][0] = (short)(a < bit ? a : bit); sm.work[idx][1] = (short)(a < bit ? bit : a); }
            else { sm.work[idx][0] = (short)bit; sm.work[idx][1] = (short)(RPK_KEYBASE + a); }
          } }
          nwork += __popcll(mk);
        }
      }
      }
      PROF(23);   // (drain rounds: candidate prefilters)
      if constexpr (PART == 1) break;   // (everything is on the candidate list)
      else {
      if (nwork == 0) break;   // (the masks are empty and nothing is pending)
      WSYNC();
      // ---- narrow phase on the first min(64, nwork) candidates
      // [MJ: mjc_*, mj_contactParam, mj_makeImpedance]
      const int nproc = nwork < 64 ? nwork : 64;
#ifndef RPK_MARK
      if (S.prof && env == 0 && lane == 0) { sm.prof[30] += nproc; sm.prof[31] += 1; }
#endif
      {
      const int base = 0;
      int w = base + lane;
      RawCon<T> rc[3];
      RawCon<T> bbx[5];   // (box-box points four to eight: memory-resident, indexed dynamically)
      int n = 0, ga = 0, gb = 0;
      T pB[8], invw = 0;
      // MESH builds: both kinds of hull pair (key box vs hull, hand geom vs hull) collect their arguments
      // here and meet at ONE call of the MPR routine below -- with a call in each branch a pass that holds
      // both kinds walks the (serial, one-lane-at-a-time) routine twice
      // (the two geom records are memory-resident -- the routine is a real call -- and are filled where the pair
      // is recognised: thirty doubles held in registers across the capsule / box code instead cost the MESH builds
      // 150 k cycles of spills per mj_step)
      bool mpr = false, mpr_flip = false;
      CGeom<T> mpr_a, mpr_b;
      mpr_a.type = GEOM_BOX_; mpr_b.type = GEOM_BOX_; mpr_a.nvert = 0; mpr_b.nvert = 0; mpr_a.vadr = 0; mpr_b.vadr = 0;
      mpr_a.flip = 0; mpr_b.flip = 0; mpr_a.graph = 0; mpr_b.graph = 0;
      auto hull_of = [&](CGeom<T>& g, int type, int geom) {
        const bool hull = type == GEOM_MESH_ && geom >= 0;
        const int gi = geom >= 0 ? geom : 0;
        g.type = type;
        g.nvert = hull ? M.geom_vertnum()[gi] : 0;
        g.vadr = hull ? M.geom_vertadr()[gi] : 0;
        g.flip = hull ? M.geom_vertflip()[gi] : 0;
        g.graph = (MESH > 1 && hull) ? M.geom_vertgraph()[gi] : 0;   // (MESH = 2: builds for scenes with graph hulls)
      };
      if (w < nproc) {
        ga = sm.work[w][0]; gb = sm.work[w][1];
        int la = M.geom_link()[ga];
        T mA[9], posA[3] = {sm.gpos[ga][0], sm.gpos[ga][1], sm.gpos[ga][2]};
        if (la >= 0) mat_mul(mA, sm.xmat[la], M.geom_mat() + 9 * ga);
        else {
#pragma unroll
          for (int k = 0; k < 9; k++) mA[k] = M.geom_mat()[9 * ga + k];
        }
        invw = M.geom_invw()[ga];
        // the partner is a box (a key, or a box geom) and geom ga a capsule or a box: both branches hand
        // the box over and meet at one capsule-box / box-box call (a pass that holds key pairs and
        // hand-hand pairs would otherwise walk each routine twice)
        bool boxside = false;
        T bxp[3] = {0, 0, 0}, bxm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const T* bxs = M.geom_size();
        if (gb >= RPK_KEYBASE) {
          int k = gb - RPK_KEYBASE;
          T s, c;
          N::sincos(sm.kq[k], &s, &c);
          T hx = M.key_half()[3 * k];
          T bp[3] = {M.key_pos()[3 * k] - hx + hx * c, M.key_pos()[3 * k + 1], M.key_pos()[3 * k + 2] - hx * s};
          T bm[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
          if (MESH && M.geom_type()[ga] == GEOM_MESH_) {
            // (box, hull) in geom-type order: the key is geom 1 of the pair; the engine keeps the hand
            // geom as side A of the contact, so the normal is turned around
            mpr = true; mpr_flip = true;
            hull_of(mpr_a, GEOM_BOX_, -1); hull_of(mpr_b, GEOM_MESH_, ga);
#pragma unroll
            for (int i = 0; i < 3; i++) { mpr_a.pos[i] = bp[i]; mpr_a.size[i] = M.key_half()[3 * k + i]; mpr_b.pos[i] = posA[i]; mpr_b.size[i] = M.geom_size()[3 * ga + i]; }
#pragma unroll
            for (int i = 0; i < 9; i++) { mpr_a.mat[i] = bm[i]; mpr_b.mat[i] = mA[i]; }
          } else if (MESH > 1 && M.geom_type()[ga] == GEOM_CYL_) {
            // (cylinder, box) in geom-type order: the hand geom is geom 1 of the pair, nothing to turn around
            mpr = true;
            hull_of(mpr_a, GEOM_CYL_, ga); hull_of(mpr_b, GEOM_BOX_, -1);
#pragma unroll
            for (int i = 0; i < 3; i++) { mpr_a.pos[i] = posA[i]; mpr_a.size[i] = M.geom_size()[3 * ga + i]; mpr_b.pos[i] = bp[i]; mpr_b.size[i] = M.key_half()[3 * k + i]; }
#pragma unroll
            for (int i = 0; i < 9; i++) { mpr_a.mat[i] = mA[i]; mpr_b.mat[i] = bm[i]; }
          } else {
            boxside = true; bxs = M.key_half() + 3 * k;
#pragma unroll
            for (int i = 0; i < 3; i++) bxp[i] = bp[i];
#pragma unroll
            for (int i = 0; i < 9; i++) bxm[i] = bm[i];
          }
#pragma unroll
          for (int e = 0; e < 8; e++) pB[e] = M.key_cparam()[e];
          invw += M.key_invw_body()[k];
        } else {
          int lb = M.geom_link()[gb];
          T mB[9], posB[3] = {sm.gpos[gb][0], sm.gpos[gb][1], sm.gpos[gb][2]};
          if (lb >= 0) mat_mul(mB, sm.xmat[lb], M.geom_mat() + 9 * gb);
          else {
#pragma unroll
            for (int k = 0; k < 9; k++) mB[k] = M.geom_mat()[9 * gb + k];
          }
          if (M.geom_type()[gb] == GEOM_CAPSULE_)
            n = capsule_capsule(rc, posA, mA, M.geom_size() + 3 * ga, posB, mB, M.geom_size() + 3 * gb);
          else if (MESH && (M.geom_type()[gb] == GEOM_MESH_ || (MESH > 1 && (M.geom_type()[ga] == GEOM_CYL_ || M.geom_type()[gb] == GEOM_CYL_)))) {
            // (hulls, and in the MESH = 2 builds cylinders: every such pair goes to the portal refinement, in geom-type
            // order -- capsule < cylinder < box < hull, which is lane order)
            mpr = true;
            hull_of(mpr_a, M.geom_type()[ga], ga); hull_of(mpr_b, MESH > 1 ? M.geom_type()[gb] : GEOM_MESH_, gb);
#pragma unroll
            for (int i = 0; i < 3; i++) { mpr_a.pos[i] = posA[i]; mpr_a.size[i] = M.geom_size()[3 * ga + i]; mpr_b.pos[i] = posB[i]; mpr_b.size[i] = M.geom_size()[3 * gb + i]; }
#pragma unroll
            for (int i = 0; i < 9; i++) { mpr_a.mat[i] = mA[i]; mpr_b.mat[i] = mB[i]; }
          } else {
            boxside = true; bxs = M.geom_size() + 3 * gb;
#pragma unroll
            for (int i = 0; i < 3; i++) bxp[i] = posB[i];
#pragma unroll
            for (int i = 0; i < 9; i++) bxm[i] = mB[i];
          }
#pragma unroll
          for (int e = 0; e < 8; e++) pB[e] = M.geom_cparam()[8 * gb + e];
          invw += M.geom_invw()[gb];
        }
        if (boxside) {
          if (M.geom_type()[ga] == GEOM_CAPSULE_) n = capsule_box(rc, posA, mA, M.geom_size() + 3 * ga, bxp, bxm, bxs);
          else n = RPK_BOXBOX(rc, bbx, posA, mA, M.geom_size() + 3 * ga, bxp, bxm, bxs);
        }
      }
      if constexpr (MESH != 0) {
        PROF(13);
        if (__ballot(mpr) != 0ull) {   // (the whole wave walks the portal refinement together)
          RawCon<T> rcm[1];
          const int nm = convex_mpr_wave<T, (MESH > 1)>(rcm, &mpr_a, &mpr_b, M.mesh_vert(), M.hull_vert, M.hull_graph, mpr, M.mpr_tol, M.mpr_tol_poly);
          if (mpr) {
            n = nm; rc[0] = rcm[0];
            if (mpr_flip) { rc[0].n[0] = -rc[0].n[0]; rc[0].n[1] = -rc[0].n[1]; rc[0].n[2] = -rc[0].n[2]; }
          }
        }
        PROF(19);
      }
      PROF(13);
      // One contact slot of every lane: records in lane order.  The first RPK_NCL records live in LDS; the rest (rare:
      // hand-on-hand pile-ups) go to the env's global overflow records through `emit_ovf`, a separate cold block, so that
      // the hot path keeps the code and the register allocation it had with one store target.
      auto contact_params = [&](const T dist, T* par) {   // par: mu, kterm (K*imp*dist), B, D
        const T* pA = M.geom_cparam() + 8 * ga;
        T solref0 = (T)0.5 * (pA[0] + pB[0]), solref1 = (T)0.5 * (pA[1] + pB[1]);
        T solimp[5];
#pragma unroll
        for (int e = 0; e < 5; e++) solimp[e] = (T)0.5 * (pA[2 + e] + pB[2 + e]);
        T mu = fmax(pA[7], pB[7]);
        if (solref0 > 0) solref0 = fmax(solref0, (T)2 * h);
        T dmax = fmin((T)0.9999, fmax((T)0.0001, solimp[1]));
        T Kc = (T)1 / fmax(RPK_MINVAL, dmax * dmax * solref0 * solref0 * solref1 * solref1);
        T Bc = (T)2 / fmax(RPK_MINVAL, dmax * solref0);
        T imp = impedance(solimp, dist);
        T Rn = fmax(RPK_MINVAL, ((T)1 - imp) * invw * ((T)1 + mu * mu) / imp);
        const T mur = mu * M.mu_scale;   // (opt.impratio: the regularised friction coefficient)
        T Rpy = fmax(RPK_MINVAL, (T)2 * mur * mur * Rn);
        par[0] = mu; par[1] = Kc * imp * dist; par[2] = Bc; par[3] = (T)1 / Rpy;
      };
      auto emit = [&](const int idx, const RawCon<T>& r) {   // idx < RPK_NCL
        T par[4];
        contact_params(r.dist, par);
#pragma unroll
        for (int k = 0; k < 3; k++) { sm.cpos[idx][k] = r.pos[k]; sm.cn[idx][k] = r.n[k]; }
        sm.cdist[idx] = r.dist;
#pragma unroll
        for (int k = 0; k < 4; k++) sm.cpar[idx][k] = par[k];
        sm.cA[idx] = M.geom_link()[ga];
        sm.cB[idx] = gb >= RPK_KEYBASE ? gb : M.geom_link()[gb];
        sm.cgA[idx] = M.geom_modelid()[ga];
        sm.cgB[idx] = gb >= RPK_KEYBASE ? M.key_geomid()[gb - RPK_KEYBASE] : M.geom_modelid()[gb];
      };
      auto emit_ovf = [&](const int at, const RawCon<T>& r) {   // RPK_NCL <= at < NCX
        T par[4];
        contact_params(r.dist, par);
        T* o = ovf + (size_t)(at - RPK_NCL) * 12;
        int* oi = ovi + (size_t)(at - RPK_NCL) * 4;
#pragma unroll
        for (int k = 0; k < 3; k++) { o[k] = r.pos[k]; o[3 + k] = r.n[k]; }
        o[6] = r.dist;
#pragma unroll
        for (int k = 0; k < 4; k++) o[7 + k] = par[k];
        oi[0] = M.geom_link()[ga];
        oi[1] = gb >= RPK_KEYBASE ? gb : M.geom_link()[gb];
        oi[2] = M.geom_modelid()[ga];
        oi[3] = gb >= RPK_KEYBASE ? M.key_geomid()[gb - RPK_KEYBASE] : M.geom_modelid()[gb];
      };
      auto emit_slot = [&](const bool has, const RawCon<T>& r) {
        unsigned long long mk = __ballot(has);
        int idx = ncon + __popcll(mk & lanemask_lt(lane));
        if (has && idx < RPK_NCL) emit(idx, r);
        if (ncon + __popcll(mk) > RPK_NCL) {   // (uniform, rare)
          if (NCX > RPK_NCL && has && idx >= RPK_NCL && idx < NCX) emit_ovf(idx, r);
          if (ncon + __popcll(mk) > NCX) {
            // capacity overflow (RP_WARN_CONTACT_FULL is raised below): keep the deepest
            // contacts -- one that does not fit replaces the shallowest stored contact if it
            // penetrates more.  Uniform loop, one overflowing lane at a time.
            RPK_STAGE_FENCE();
            unsigned long long ovm = __ballot(has && idx >= NCX);
            while (ovm) {
              const int L = __ffsll((long long)ovm) - 1;
              ovm &= ovm - 1;
              const T dL = bcast(r.dist, L);
              T worst = sm.cdist[0];
              int wi = 0;
              for (int c = 1; c < NCX; c++) {
                const T d = (NCX <= RPK_NCL || c < RPK_NCL) ? sm.cdist[c < RPK_NCL ? c : 0] : ovf[(size_t)(c - RPK_NCL) * 12 + 6];
                if (d > worst) { worst = d; wi = c; }
              }
              if (dL < worst && lane == L) { if (NCX <= RPK_NCL || wi < RPK_NCL) emit(wi < RPK_NCL ? wi : 0, r); else emit_ovf(wi, r); }
              RPK_STAGE_FENCE();
            }
          }
        }
        ncon += __popcll(mk);
      };
#pragma unroll
      for (int slot = 0; slot < 3; slot++) emit_slot(n > slot, rc[slot]);
      // box-box pairs resting face to face: points four to eight, from the routine's result array
      if (RPK_BOXBOX_MAX > 3 && __ballot(n > 3) != 0ull) {
        for (int slot = 3; slot < 8; slot++) {
          RawCon<T> r = bbx[n > slot ? slot - 3 : 0];
          emit_slot(n > slot, r);
        }
      }
      }
      PROF(24);
      // drop the processed candidates
      WSYNC();
      short w0 = 0, w1 = 0;
      const bool mv = lane + 64 < nwork;
      if (mv) { w0 = sm.work[lane + 64][0]; w1 = sm.work[lane + 64][1]; }
      WSYNC();
      if (mv) { sm.work[lane][0] = w0; sm.work[lane][1] = w1; }
      nwork -= nproc;
      WSYNC();
      }   // (PART != 1)
    }
    if constexpr (PART == 1) {
      // ---- split stage, front part: the candidate list leaves the wave.  Every candidate gets its result records
      // (by pair type: 2 / 2 / 8 / 1) and an entry on its type's POOLED list, which the narrow-phase kernel walks
      // with lane = candidate whatever env it came from.  (Where an entry lands on that list depends on the order
      // in which the waves of the launch arrive; the results do not: they return to the candidate's own records.)
      int nc_ = nwork;
      if (nc_ > RPK_NCAND) { warn |= 64; nc_ = RPK_NCAND; }   // (RP_WARN_SPLIT_FULL)
      WSYNC();
      int rbase = 0;
      int* const cl_ = B.cand + (size_t)env * RPK_NCAND * 2;
      int* const tc_ = B.tcount + B.tcount_off + (env & (RPK_NSTRIPE - 1)) * RPK_NTYPE_PAD;
      for (int c0 = 0; c0 < nc_; c0 += 64) {
        const int i = c0 + lane;
        const bool in = i < nc_;
        const int pair = in ? sm.clist[in ? i : 0] : 0;
        const int ga = pair & 0xffff, gb = (pair >> 16) & 0xffff;
        const int ta = M.geom_type()[ga];
        int ty;
        if (gb >= RPK_KEYBASE) ty = ta == GEOM_CAPSULE_ ? 1 : (ta == GEOM_BOX_ ? 2 : 3);
        else {
          const int tb = M.geom_type()[gb];
          ty = tb == GEOM_CAPSULE_ ? 0 : ((tb == GEOM_MESH_ || ta == GEOM_MESH_ || (MESH > 1 && (ta == GEOM_CYL_ || tb == GEOM_CYL_))) ? 3 : (ta == GEOM_CAPSULE_ ? 1 : 2));
        }
        if (MESH != 0 && ty == 3) {   // hull pairs: the bucket of lanes that need the same vertex scans (rp_model.hpp)
          const int hb_ = gb >= RPK_KEYBASE ? ga : gb;   // (side B of the refinement: the hull)
          const int vlast = M.geom_vertadr()[M.ngeom > 0 ? M.ngeom - 1 : 0];   // (geoms are sorted by type: hulls last)
          ty = 3 + ((gb < RPK_KEYBASE && ta == GEOM_MESH_) ? 1 : 0) + (M.geom_vertadr()[hb_] != vlast ? 2 : 0);
        }
        const int wdt = ty == 2 ? 8 : (ty >= 3 ? 1 : 2);
        int incl = in ? wdt : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        const int myb = rbase + incl - (in ? wdt : 0);
        const bool ok = in && myb + wdt <= RPK_NRES;
        if (__ballot(in && !ok) != 0ull) warn |= 64;   // (its records do not fit: the candidate is dropped, flagged: RP_WARN_SPLIT_FULL)
        rbase += bcast(incl, 63);
        unsigned long long mine = 0ull;
        int cntT = 0;
#pragma unroll
        for (int t = 0; t < RPK_NTYPE; t++) {
          const unsigned long long mt = __ballot(ok && ty == t);
          if (ty == t) mine = mt;
          if (lane == t) cntT = __popcll(mt);
        }
        // (B.tlist null: the fused schedule -- the env's own wave runs its narrow phase from the candidate list, no pooled lists)
        int tb_ = 0;
        if (B.tlist && lane < RPK_NTYPE && cntT > 0) tb_ = atomicAdd(&tc_[lane], cntT);
        const int tbase = __shfl(tb_, ty, 64);
        if (in) {
          cl_[2 * i] = pair; cl_[2 * i + 1] = (ok ? myb : 0) | (ty << 16) | (ok ? 1 << 20 : 0);
          B.cres_n[(size_t)env * RPK_NCAND + i] = 0;
        }
        if (ok && B.tlist) {
          int4* e_ = (int4*)(B.tlist + (((size_t)ty * RPK_NSTRIPE + (env & (RPK_NSTRIPE - 1))) * B.tstride + (size_t)(S.env_base / RPK_NSTRIPE) * RPK_NCAND + tbase +
                                       __popcll(mine & lanemask_lt(lane))) * 4);
          int4 v_; v_.x = env; v_.y = pair; v_.z = myb; v_.w = i;
          *e_ = v_;
        }
      }
      if (lane == 0) {
        B.ncand[env] = nc_;
        if ((warn & 64) && B.split_dropped) atomicAdd(B.split_dropped, 1);   // (the host then stops choosing this schedule)
      }
    }
    }
    }  // PART != 2
    if constexpr (PART == 2) {
      // ---- split stage, back part: the link frames the front part left, then the narrow phase's contacts in the
      // emission order of the whole stage -- candidates in chunks of 64 (the passes of its work list), within a chunk
      // point 1 of every candidate, then point 2, ... -- so that both ways of running the stage produce the same bits.
      if (isl) {
        const T* fr_ = B.frames + (size_t)env * RPK_NFRAME * 64 + lane;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          sm.xpos[lane][k] = fr_[(size_t)k * 64]; sm.xaxis[lane][k] = fr_[(size_t)(12 + k) * 64]; sm.xanchor[lane][k] = fr_[(size_t)(15 + k) * 64];
        }
#pragma unroll
        for (int k = 0; k < 9; k++) sm.xmat[lane][k] = fr_[(size_t)(3 + k) * 64];
      }
#pragma unroll
      for (int s = 0; s < 2; s++) {
        if (isk[s]) { N::sincos(q[1 + s], &ksin[s], &kcos[s]); sm.keyslot[kid[s]] = -1; }
      }
      ncon = 0;
      const int nc_ = B.ncand[env];
      const int* const cl_ = B.cand + (size_t)env * RPK_NCAND * 2;
      const T* const res_ = B.cres + (size_t)env * RPK_NRES * 12;
      for (int c0 = 0; c0 < nc_; c0 += 64) {
        const int i = c0 + lane;
        const bool in = i < nc_;
        const int pair = in ? cl_[2 * i] : 0, meta = in ? cl_[2 * i + 1] : 0;
        const int n = in ? B.cres_n[(size_t)env * RPK_NCAND + i] : 0;
        const int ga = pair & 0xffff, gb = (pair >> 16) & 0xffff, rb = meta & 0xffff;
        const int iA = M.geom_link()[ga], iB = gb >= RPK_KEYBASE ? gb : M.geom_link()[gb >= RPK_KEYBASE ? 0 : gb];
        const int igA = M.geom_modelid()[ga], igB = gb >= RPK_KEYBASE ? M.key_geomid()[gb - RPK_KEYBASE] : M.geom_modelid()[gb >= RPK_KEYBASE ? 0 : gb];
        auto put = [&](const int at, const T* r) {   // record r -> contact slot `at` (LDS, or the env's overflow records)
          if (NCX <= RPK_NCL || at < RPK_NCL) {
            const int a_ = at < RPK_NCL ? at : 0;
#pragma unroll
            for (int k = 0; k < 3; k++) { sm.cpos[a_][k] = r[k]; sm.cn[a_][k] = r[3 + k]; }
            sm.cdist[a_] = r[6];
#pragma unroll
            for (int k = 0; k < 4; k++) sm.cpar[a_][k] = r[7 + k];
            sm.cA[a_] = iA; sm.cB[a_] = iB; sm.cgA[a_] = igA; sm.cgB[a_] = igB;
          } else {
            T* o = ovf + (size_t)(at - RPK_NCL) * 12;
            int* oi = ovi + (size_t)(at - RPK_NCL) * 4;
#pragma unroll
            for (int k = 0; k < 11; k++) o[k] = r[k];
            oi[0] = iA; oi[1] = iB; oi[2] = igA; oi[3] = igB;
          }
        };
        // (the first result record of every candidate is requested together with the geom tables above: one trip to L2
        // instead of two)
        T r0v[11];
        {
          const T* r0p = res_ + (size_t)rb * 12;
#pragma unroll
          for (int k = 0; k < 11; k++) r0v[k] = r0p[k];
        }
        __builtin_amdgcn_sched_barrier(0);
        const int nslot = __ballot(n > 3) != 0ull ? 8 : 3;
        for (int slot = 0; slot < nslot; slot++) {
          const bool has = n > slot;
          const unsigned long long mk = __ballot(has);
          if (mk == 0ull) continue;
          T r[11];
          if (slot == 0) {
#pragma unroll
            for (int k = 0; k < 11; k++) r[k] = r0v[k];
          } else {
            const T* rg = res_ + (size_t)(rb + (has ? slot : 0)) * 12;
#pragma unroll
            for (int k = 0; k < 11; k++) r[k] = rg[k];
          }
          const int idx = ncon + __popcll(mk & lanemask_lt(lane));
          if (has && idx < NCX) put(idx, r);
          if (ncon + __popcll(mk) > NCX) {
            // capacity overflow: keep the deepest contacts (the whole stage's procedure, one overflowing lane at a time)
            RPK_STAGE_FENCE();
            unsigned long long ovm = __ballot(has && idx >= NCX);
            const T myd = has ? r[6] : (T)0;
            while (ovm) {
              const int L_ = __ffsll((long long)ovm) - 1;
              ovm &= ovm - 1;
              const T dL = bcast(myd, L_);
              T worst = sm.cdist[0];
              int wi = 0;
              for (int c = 1; c < NCX; c++) {
                const T d = (NCX <= RPK_NCL || c < RPK_NCL) ? sm.cdist[c < RPK_NCL ? c : 0] : ovf[(size_t)(c - RPK_NCL) * 12 + 6];
                if (d > worst) { worst = d; wi = c; }
              }
              if (dL < worst && lane == L_) put(wi, r);
              RPK_STAGE_FENCE();
            }
          }
          ncon += __popcll(mk);
        }
      }
      if (lane == 0) B.ncand[env] = -1;   // (consumed)
    }
    if constexpr (PART != 1) {
    if (ncon > NCX) { warn |= 2; ncon = NCX; }
    WSYNC();
    if (NCX > RPK_NCL && ncon > RPK_NCL) RPK_STAGE_FENCE();   // (the overflow records: written by the emitting lanes, read below by the contact lanes)

    // The position / velocity stages are register-bound around the narrow phase: the state is re-read here
    // (L2 hits) instead of being carried -- that is: spilled to scratch and reloaded -- across the collision.
    if constexpr (MODE != 1) {
      {
        const int4* rec = (const int4*)(fresh(M.lane_topo()) + 16 * L);
        const int4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
        parent = isl ? r0.x : -1; depth = isl ? r0.y : -1; jtype = isl ? r0.z : 0; sibrank = isl ? r0.w : 0;
        ltree = isl ? r1.x : 0; ldof = isl ? r1.y : 0; tbase = isl ? r1.z : 0; TL = isl ? r1.w : 0;
        ndesc = isl ? r2.x : 0; llimited = isl ? r2.y : 0; lact = isl ? r2.z : -1;
#pragma unroll
        for (int s = 0; s < 2; s++) {
          kdof[s] = isk[s] ? fresh(M.key_dof())[kid[s]] : 0;
          kact[s] = isk[s] ? fresh(M.key_act())[kid[s]] : -1;
        }
      }
      const T *p_q = fresh(S.qpos), *p_v = fresh(S.qvel);
      q[0] = isl ? p_q[eo + ldof] : (T)0; qd[0] = isl ? p_v[eo + ldof] : (T)0;
#pragma unroll
      for (int s = 0; s < 2; s++) {
        q[1 + s] = isk[s] ? p_q[eo + kdof[s]] : (T)0;
        qd[1 + s] = isk[s] ? p_v[eo + kdof[s]] : (T)0;
      }
    }
    PROF(13);
    // ---- solver slots for touched keys
    int dpA_ = -1, dpB_ = -1;   // depths of my contact's two links (read with their ancestor masks)
    T khx_pre = 0, khz_pre = 0;  // slot lanes: hinge line of my key (read with the anchor's topology record)
    {
      int kb = -1;
      if (lane < ncon) {
        const int a_ = coni(0), b_ = coni(1);
        if (b_ >= RPK_KEYBASE) kb = b_ - RPK_KEYBASE;
        else if (a_ >= RPK_KEYBASE) kb = a_ - RPK_KEYBASE;
      }
      bool first = kb >= 0;
      for (int c2 = 0; c2 < ncon; c2++) {
        int k2 = bcast(kb, c2);
        if (c2 < lane && k2 == kb) first = false;
      }
      unsigned long long fm = __ballot(first);
      int slot = __popcll(fm & lanemask_lt(lane));
      nkt = __popcll(fm);
      int cap = min(64 - nl, 16);
      if (first && slot < cap) { sm.slotkey[slot] = kb; sm.keyslot[kb] = slot; }
      if (nkt > cap) { warn |= 8; nkt = cap; }
      WSYNC();
      con_slot = kb >= 0 ? sm.keyslot[kb] : -1;
      // per-contact registers
      con_A = -1; con_B = -1; con_D = 0; con_mu = 0; con_maskA = 0; con_maskB = 0;
      dpA_ = -1; dpB_ = -1;
      if (lane < ncon) {
        con_A = coni(0); con_B = coni(1);
        con_mu = conf(7);
        con_D = conf(10);
        con_dist = conf(6);  // (the LDS copy is recycled by the velocity stage)
        if (kb >= 0 && con_slot < 0) con_D = 0;  // dropped (slot overflow)
#pragma unroll
        for (int k = 0; k < 3; k++) con_n[k] = conf(3 + k);
        make_frame(con_n, con_t1, con_t2);
        {
          // (both bodies' ancestor masks requested together, clamped indices, selected afterwards)
          const bool la_ = con_A >= 0 && con_A < RPK_KEYBASE, lb_ = con_B >= 0 && con_B < RPK_KEYBASE;
          const int ia_ = la_ ? con_A : 0, ib_ = lb_ ? con_B : 0;
          const unsigned a0_ = M.link_ancmask_u()[2 * ia_], a1_ = M.link_ancmask_u()[2 * ia_ + 1];
          const unsigned b0_ = M.link_ancmask_u()[2 * ib_], b1_ = M.link_ancmask_u()[2 * ib_ + 1];
          const int da_ = M.link_depth()[ia_], db_ = M.link_depth()[ib_];   // (for the key anchors below)
          __builtin_amdgcn_sched_barrier(0);
          if (la_) { con_maskA = ((unsigned long long)a1_ << 32) | a0_; dpA_ = da_; }
          if (lb_) { con_maskB = ((unsigned long long)b1_ << 32) | b0_; dpB_ = db_; }
        }
      }
    }
    // ======================================================================
    // ---- nested link-link contacts (one link an ancestor of the other) collapse to a
    // single chain of columns; then decide whether the Hessian is tree-structured.
    // chains of the two bodies as they are (a dof on both chains moves both bodies: its
    // Jacobian column is the difference, i.e. exactly zero)
    const unsigned long long omA = con_maskA, omB = con_maskB;
    const int bodyA = con_A, bodyB = con_B;   // (the nested-contact collapse below edits con_A / con_B)
    T con_pos[3] = {0, 0, 0};
    if constexpr (MODE == 2) {
      if (lane < ncon) { con_pos[0] = conf(0); con_pos[1] = conf(1); con_pos[2] = conf(2); }
    }
    {
      int cross = 0;
      if (lane < ncon) {
        const bool la = con_A >= 0 && con_A < RPK_KEYBASE, lb = con_B >= 0 && con_B < RPK_KEYBASE;
        if (la && lb) {
          int deep = -1;
          if ((con_maskA & ~con_maskB) == 0) deep = 1;       // A's chain is a prefix of B's
          else if ((con_maskB & ~con_maskA) == 0) deep = 0;
          if (deep < 0) cross = 1;
          else {  // the support of the contact is the deeper chain
            const int sh = 1 - deep;
            if (sh == 0) { con_A = -1; con_maskA = 0; }
            else { con_B = -1; con_maskB = 0; }
          }
        }
      }
      // anchor link of every touched key = deepest link pressing it; all other links
      // pressing the same key must lie on the anchor's chain
      const int mylink = (con_A >= 0 && con_A < RPK_KEYBASE) ? con_A : ((con_B >= 0 && con_B < RPK_KEYBASE) ? con_B : -1);
      const unsigned long long mymask = con_maskA | con_maskB;
      const int mydepthc = (lane < ncon && mylink >= 0) ? (mylink == bodyA ? dpA_ : dpB_) : -1;
      for (int sidx = 0; sidx < nkt; sidx++) {
        int cand = (lane < ncon && con_slot == sidx && mylink >= 0) ? ((mydepthc << 8) | lane) : -1;
        cand = wave_max(cand);
        int cl = cand >= 0 ? (cand & 255) : 0;
        unsigned long long am = ((unsigned long long)(unsigned)bcast((int)(mymask >> 32), cl) << 32) |
                                (unsigned)bcast((int)(mymask & 0xffffffffu), cl);
        int alink = bcast(mylink, cl);
        if (cand < 0) { am = 0; alink = -1; }
        if (lane < ncon && con_slot == sidx && (mymask & ~am) != 0) cross = 1;
        if (lane == 0) { sm.slotmask[sidx] = am; sm.slotlink[sidx] = alink; }
      }
      // a key pressed from two different chains makes all of its contacts cross contacts
      {
        unsigned long long badslots = 0;
        if (cross && lane < ncon && con_slot >= 0) badslots = 1ull << con_slot;
        badslots = wave_or(badslots);
        if (lane < ncon && con_slot >= 0 && ((badslots >> con_slot) & 1)) cross = 1;
      }
      con_cross = cross;
      {
        unsigned long long dmk = 0;
        if (cross) {
          dmk = con_maskA | con_maskB;
          if (con_slot >= 0) dmk |= 1ull << (nl + con_slot);
        }
        dmk = wave_or(dmk);
        dirty_mask = dmk;
      }
      tree_ok = dirty_mask == 0ull;
      WSYNC();
      sdepth = -1;
      if (!isl && lane < nl + nkt) {
        // (the anchor's topology record and the key's hinge line in one trip to L2; the hinge line is used by the Jacobian
        // pass below)
        const int al = sm.slotlink[lane - nl];
        const int k_ = sm.slotkey[lane - nl];
        const int* rec = M.lane_topo() + 16 * (al >= 0 ? al : 0);
        const int rc1 = rec[1], rc6 = rec[6], rc7 = rec[7];
        const T kp0 = M.key_pos()[3 * k_], kh0 = M.key_half()[3 * k_], kp2 = M.key_pos()[3 * k_ + 2];
        __builtin_amdgcn_sched_barrier(0);
        khx_pre = kp0 - kh0; khz_pre = kp2;
        if (al >= 0) {
          sdepth = rc1;
          salink = al; sTB = rc6; sTL = rc7;
        }
      }
    }
    WSYNC();
    // ---- contact Jacobians [MJ: mj_jacDifPair], J qvel for the reference accelerations, and
    // the entry list handed to the solver, in one pass over the contacts: every dof lane in
    // the support of contact c (link lanes on either chain, the key's solver slot) computes
    // its own column d(contact point velocity)/d(qvel), adds column * qvel to the contact's
    // sum and writes its entry.  Entries of a contact are in lane order: ancestors first,
    // the key slot last.
    {
      if (isk[0]) sm.keyvec[0][kid[0]] = qd[1];
      if (isk[1]) sm.keyvec[0][kid[1]] = qd[2];
      T cvr[3] = {0, 0, 0};   // J qvel of MY contact (contact lanes)
      unsigned long long sup = 0;
      if (lane < ncon) {
        sup = con_maskA | con_maskB;
        if (con_slot >= 0) sup |= 1ull << (nl + con_slot);
      }
      int cnt = __popcll(sup), base = 0, tot = 0;
      for (int c2 = 0; c2 < ncon; c2++) { const int b2 = bcast(cnt, c2); if (c2 < lane) base += b2; tot += b2; }
      if (tot > RpCaps<T>::NE) {
        // entry capacity overflow (rare: hand-hand pile-ups, ~14 entries per contact): drop the
        // shallowest contacts until the rest fits.  Uniform loops.
        warn |= 2;
        while (tot > RpCaps<T>::NE) {
          int wi = 0;
          T worst = 0;
          bool any = false;
          for (int c2 = 0; c2 < ncon; c2++) {
            const T d2 = bcast(con_dist, c2);
            if (bcast(cnt, c2) > 0 && (!any || d2 > worst)) { worst = d2; wi = c2; any = true; }
          }
          tot -= bcast(cnt, wi);
          if (lane == wi) { cnt = 0; sup = 0; con_D = 0; }
        }
        base = 0;
        for (int c2 = 0; c2 < ncon; c2++) { const int b2 = bcast(cnt, c2); if (c2 < lane) base += b2; }
      }
      const int nent = wave_max(lane < ncon ? base + cnt : 0), maxm = wave_max(cnt);
      if constexpr (MODE == 0) {
        LI(10) = base | (cnt << 12);
        if (lane == 0) {
          B.hdr[env * 8 + 4] = nent; B.hdr[env * 8 + 5] = maxm;
          // capacity class of this env's solve (rp_solver2.hpp): light = fits the lean solver stage
          B.hdr[env * 8 + 6] = (S.lean && MD == RPK_MAXD && M.ntree <= 2 && ncon <= LeanCaps::NC && nent <= (S.lean > 1 && S.lean < LeanCaps::NE ? S.lean : LeanCaps::NE) &&
                                __popcll(dirty_mask) <= LeanCaps::HMAX && nkt <= LeanCaps::NK && nl + nkt <= 64) ? 1 : 0;
        }
      }
      WSYNC();
      const int mycol = isl ? depth : sdepth + 1;
      T ax_[3] = {0, 0, 0}, an_[3] = {0, 0, 0}, xv = 0, khx_ = 0, khz_ = 0;
      if (isl) {
#pragma unroll
        for (int k = 0; k < 3; k++) { ax_[k] = sm.xaxis[lane][k]; an_[k] = sm.xanchor[lane][k]; }
        xv = qd[0];
      } else if (lane < nl + nkt) {
        const int k = sm.slotkey[lane - nl];
        xv = sm.keyvec[0][k];
        khx_ = khx_pre;  // hinge line x, z
        khz_ = khz_pre;
      }
      for (int c = 0; c < ncon; c++) {
        const unsigned long long sc = ((unsigned long long)(unsigned)bcast((int)(sup >> 32), c) << 32) |
                                      (unsigned)bcast((int)(sup & 0xffffffffu), c);
        const unsigned long long mA = ((unsigned long long)(unsigned)bcast((int)(omA >> 32), c) << 32) |
                                      (unsigned)bcast((int)(omA & 0xffffffffu), c);
        const unsigned long long mB = ((unsigned long long)(unsigned)bcast((int)(omB >> 32), c) << 32) |
                                      (unsigned)bcast((int)(omB & 0xffffffffu), c);
        const int cA = bcast(con_A, c), cb = bcast(base, c), cc = bcast(cnt, c), cx = bcast(con_cross, c);
        T px, py, pz;
        if (NCX <= RPK_NCL || c < RPK_NCL) { px = sm.cpos[c][0]; py = sm.cpos[c][1]; pz = sm.cpos[c][2]; }
        else { const T* o_ = ovf + (size_t)(c - RPK_NCL) * 12; px = o_[0]; py = o_[1]; pz = o_[2]; }
        T jv3[3] = {0, 0, 0};   // my column times my velocity
        if ((sc >> lane) & 1) {
          T j3[3];
          if (isl) {
            T col[3];
            if (jtype == JNT_SLIDE_) { col[0] = ax_[0]; col[1] = ax_[1]; col[2] = ax_[2]; }
            else {
              const T r[3] = {px - an_[0], py - an_[1], pz - an_[2]};
              cross3(col, ax_, r);
            }
            const T sA = ((mA >> lane) & 1) ? (T)-1 : (T)0, sB = ((mB >> lane) & 1) ? (T)1 : (T)0;
            j3[0] = sA * col[0] + sB * col[0]; j3[1] = sA * col[1] + sB * col[1]; j3[2] = sA * col[2] + sB * col[2];
          } else {
            const T sg = cA >= RPK_KEYBASE ? (T)-1 : (T)1;   // the key is body 1 (side A) or body 2
            const T rx = px - khx_, rz = pz - khz_;
            j3[0] = sg * rz; j3[1] = 0; j3[2] = sg * -rx;    // (0,1,0) x r
          }
          jv3[0] = j3[0] * xv; jv3[1] = j3[1] * xv; jv3[2] = j3[2] * xv;
          const int rank = __popcll(sc & lanemask_lt(lane));
          if constexpr (MODE == 0) {
            const size_t e = (size_t)env * RpCaps<T>::NE + cb + rank;
            B.entJ[e * 3] = j3[0]; B.entJ[e * 3 + 1] = j3[1]; B.entJ[e * 3 + 2] = j3[2];
            B.entM[e * 2] = RPK_EM0(lane, c, mycol, cx);
            B.entM[e * 2 + 1] = RPK_EM1(cb, cc, rank);
          }
        }
        // J qvel of contact c: a wave sum (DPP) of the ~10 member lanes' products, kept by the contact's lane.
        // (It used to be three LDS adds into the contact's cell: ten lanes on one address cost ~450 cycles of
        // the CU's LDS pipe per instruction, scratch/ub/ldsadd_ub.hip.)
        const T s0 = wave_sum(jv3[0]), s1 = wave_sum(jv3[1]), s2 = wave_sum(jv3[2]);
        if (lane == c) { cvr[0] = s0; cvr[1] = s1; cvr[2] = s2; }
      }
      WSYNC();
      if (lane < ncon) {
        const T vc[3] = {cvr[0], cvr[1], cvr[2]};
        const T vn = dot3(con_n, vc), v1 = con_mu * dot3(con_t1, vc), v2 = con_mu * dot3(con_t2, vc);
        const T Bc = conf(9), kt = conf(8);
        con_aref[0] = -Bc * (vn + v1) - kt; con_aref[1] = -Bc * (vn - v1) - kt;
        con_aref[2] = -Bc * (vn + v2) - kt; con_aref[3] = -Bc * (vn - v2) - kt;
      } else {
        con_aref[0] = con_aref[1] = con_aref[2] = con_aref[3] = 0;
      }
    }
    WSYNC();
    PROF(14);
    // VELOCITY STAGE
    // ======================================================================
    // ---- spatial velocities, axis derivatives [MJ: mj_comVel]
    if constexpr (MODE != 1) {
      // the motion axis about the tree reference point again, from the link frames in LDS (same arithmetic
      // as in mj_comPos above: same bits), instead of six more registers carried across the collision
      if (isl) {
        const T* p_tref = fresh(M.tree_ref());
        const T axw[3] = {sm.xaxis[lane][0], sm.xaxis[lane][1], sm.xaxis[lane][2]};
        if (jtype == JNT_SLIDE_) {
          cdofr[0] = cdofr[1] = cdofr[2] = 0;
          cdofr[3] = axw[0]; cdofr[4] = axw[1]; cdofr[5] = axw[2];
        } else {
          T off[3];
#pragma unroll
          for (int k = 0; k < 3; k++) off[k] = p_tref[3 * ltree + k] - sm.xanchor[lane][k];
          cdofr[0] = axw[0]; cdofr[1] = axw[1]; cdofr[2] = axw[2];
          cross3(cdofr + 3, axw, off);
        }
      }
    }
    T cv[6] = {0, 0, 0, 0, 0, 0}, cdd[6] = {0, 0, 0, 0, 0, 0};
    for (int d = 0; d < M.maxdepth; d++) {
      if (isl && depth == d) {
        T pv[6] = {0, 0, 0, 0, 0, 0};
        if (parent >= 0) {
#pragma unroll
          for (int k = 0; k < 6; k++) pv[k] = sm.vel[parent][k];
        }
        T t1[3], t2[3];
        cross3(cdd, pv, cdofr);
        cross3(t1, pv, cdofr + 3); cross3(t2, pv + 3, cdofr);
        cdd[3] = t1[0] + t2[0]; cdd[4] = t1[1] + t2[1]; cdd[5] = t1[2] + t2[2];
#pragma unroll
        for (int k = 0; k < 6; k++) { cv[k] = pv[k] + cdofr[k] * qd[0]; sm.vel[lane][k] = cv[k]; }
      }
      WSYNC();
    }
    // ---- bias forces: recursive Newton-Euler with gravity as base acceleration [MJ: mj_rne]
    T ca[6] = {0, 0, 0, 0, 0, 0};
    const T gscale = fresh(M.link_gscale())[L];   // (once, in front of the level loop: inside it every level waited for its own trip to L2)
    for (int d = 0; d < M.maxdepth; d++) {
      if (isl && depth == d) {
        T pa[6] = {0, 0, 0, -M.gx * gscale, -M.gy * gscale, -M.gz * gscale};
        if (parent >= 0) {
#pragma unroll
          for (int k = 0; k < 6; k++) pa[k] = sm.vel[parent][k];
        }
#pragma unroll
        for (int k = 0; k < 6; k++) ca[k] = pa[k] + cdd[k] * qd[0];
      }
      WSYNC();  // all reads of this level's parents (cvel or cacc) are done
      if (isl && depth == d) {
#pragma unroll
        for (int k = 0; k < 6; k++) sm.vel[lane][k] = ca[k];
      }
      WSYNC();
    }
    {
      T f1[6], iv[6], f2[6], t1[3], t2[3];
      T cin[10];
      link_inertia(cin);
      mul_inert(f1, cin, ca);
      mul_inert(iv, cin, cv);
      cross3(t1, cv, iv); cross3(t2, cv + 3, iv + 3);
      f2[0] = t1[0] + t2[0]; f2[1] = t1[1] + t2[1]; f2[2] = t1[2] + t2[2];
      cross3(f2 + 3, cv, iv + 3);
      if (isl) {
#pragma unroll
        for (int k = 0; k < 6; k++) sm.acc[lane][k] = f1[k] + f2[k];
      }
    }
    WSYNC();
    for (int d = M.maxdepth - 1; d >= 1; d--) {
      if (isl && depth == d) {
#pragma unroll
        for (int k = 0; k < 6; k++) lds_add(&sm.acc[parent][k], sm.acc[lane][k]);
      }
      WSYNC();
    }
    qbias = isl ? dot6(cdofr, sm.acc[lane]) : (T)0;
    PROF(15);
    // ---- transmission [MJ: mj_transmission]: actuator length / velocity
    sm.vec[0][lane] = q[0]; sm.vec[1][lane] = qd[0];
    if (isk[0]) sm.keyvec[0][kid[0]] = qd[1];
    if (isk[1]) sm.keyvec[0][kid[1]] = qd[2];
    WSYNC();
    if (isa) {
      // (read here, not in the prologue: nothing else needs them and the stage is register-bound)
      const int alane0 = fresh(M.act_lane())[2 * A], alane1 = fresh(M.act_lane())[2 * A + 1];
      const T acoef0 = fresh(M.act_coef())[2 * A], acoef1 = fresh(M.act_coef())[2 * A + 1];
      alen = acoef0 * sm.vec[0][alane0] + (alane1 >= 0 ? acoef1 * sm.vec[0][alane1] : (T)0);
      avel = acoef0 * sm.vec[1][alane0] + (alane1 >= 0 ? acoef1 * sm.vec[1][alane1] : (T)0);
    }
    RPK_LOAD_LIMITS
    // ---- constraint rows: reference accelerations [MJ: mj_makeConstraint, mj_referenceConstraint]
    fr_aref = -lflB * qd[0];
#pragma unroll
    for (int s = 0; s < 3; s++) {
      lim_sign[s] = 0; lim_D[s] = 0; lim_aref[s] = 0;
      if (hasdof[s]) {
        T dl = q[s] - lo[s], du = hi[s] - q[s];
        T dist = 0;
        if (dl < 0) { lim_sign[s] = 1; dist = dl; }
        else if (du < 0) { lim_sign[s] = -1; dist = du; }
        if (lim_sign[s] != 0) {
          const T* si = (s == 0) ? (M.link_lim_solimp() + 5 * L) : (M.key_lim_solimp() + 5 * kid[s - 1]);
          T imp = impedance(si, dist);
          T R = fmax(RPK_MINVAL, ((T)1 - imp) * limW[s] / imp);
          lim_D[s] = (T)1 / R;
          lim_aref[s] = -limB[s] * ((T)lim_sign[s] * qd[s]) - limK[s] * imp * dist;
        }
      }
    }
    WSYNC();

    PROF(16);
    // ---- per-substep key activation trace (Piano._update_key_state, piano.py:178-192)
    if (MODE == 0 && S.key_trace && substep >= 0) {
      unsigned long long b0 = __ballot(isk[0] && (fmin(hi[1], fmax(lo[1], q[1])) >= hi[1] - (T)0.00872665));
      unsigned long long b1 = __ballot(isk[1] && (fmin(hi[2], fmax(lo[2], q[2])) >= hi[2] - (T)0.00872665));
      if (lane == 0) {
        uint32_t* o = S.key_trace + ((size_t)env * nsub + substep) * 4;
        o[0] = (uint32_t)b0; o[1] = (uint32_t)(b0 >> 32); o[2] = (uint32_t)b1; o[3] = (uint32_t)(b1 >> 32);
      }
    }
    // ---- hand over to the solver kernel
    if constexpr (MODE == 0) {
      LF(0) = qbias;
      if (lane < nu) { LF(1) = alen; LF(2) = avel; }
      LF(3) = ksin[0]; LF(5) = kcos[0];
      if (isk[1]) { LF(4) = ksin[1]; LF(6) = kcos[1]; }
      LF(7) = fr_aref;
#pragma unroll
      for (int k = 0; k < 3; k++) if (lim_sign[k] != 0) { LF(8 + k) = lim_D[k]; LF(11 + k) = lim_aref[k]; }
      if (lane < ncon) {  // (the solver stage reads these for lanes < ncon only)
        LF(14) = con_D; LF(15) = con_mu;
#pragma unroll
        for (int k = 0; k < 3; k++) { LF(16 + k) = con_n[k]; LF(19 + k) = con_t1[k]; LF(22 + k) = con_t2[k]; }
#pragma unroll
        for (int k = 0; k < 4; k++) LF(25 + k) = con_aref[k];
        LI(1) = con_A; LI(2) = con_B; LI(3) = con_slot; LI(4) = con_cross;
        LI(5) = (int)(con_maskA & 0xffffffffu); LI(6) = (int)(con_maskA >> 32);
        LI(7) = (int)(con_maskB & 0xffffffffu); LI(8) = (int)(con_maskB >> 32);
      }
      LI(0) = (lim_sign[0] + 1) | ((lim_sign[1] + 1) << 2) | ((lim_sign[2] + 1) << 4);
      LI(9) = sdepth;
      LI(11) = salink | (sTL << 8) | (sTB << 16);
      if (lane < 16) {
        int* sl = B.slots + (size_t)env * 64;
        sl[lane] = sm.slotkey[lane]; sl[16 + lane] = sm.slotlink[lane];
        sl[32 + lane] = (int)(sm.slotmask[lane] & 0xffffffffu); sl[48 + lane] = (int)(sm.slotmask[lane] >> 32);
      }
      if (lane < RPK_NKEYS / 4) B.keyslot[(size_t)env * (RPK_NKEYS / 4) + lane] = ((int*)sm.keyslot)[lane];
      if (lane == 0) {
        B.hdr[env * 8] = ncon; B.hdr[env * 8 + 1] = nkt;
        B.hdr[env * 8 + 2] = (int)(dirty_mask & 0xffffffffu); B.hdr[env * 8 + 3] = (int)(dirty_mask >> 32);
      }
    }

    // ======================================================================
    // MODE 2: ACCELERATION-STAGE SENSORS  [MJ: mj_rnePostConstraint, mj_sensorAcc]
    // torque sensors at every hand joint's body origin, projected on the joint axis
    // (robopianist/models/hands/shadow_hand.py:209-226, hands/base.py:101-109) and the
    // fingertip touch sensors (:248-270), from the constrained qacc (S.warm holds it) and the
    // contact row forces the solver stage stored (S.con_force).
    // ======================================================================
    if constexpr (MODE == 2) {
      if (isl) {
#pragma unroll
        for (int k = 0; k < 6; k++) sm.fext[lane][k] = 0;
      }
      sm.touch[lane] = 0;
      WSYNC();
      if (lane < ncon) {
        T f[4];
#pragma unroll
        for (int r = 0; r < 4; r++) f[r] = S.con_force[((size_t)env * RPK_NCOUT + lane) * 4 + r];
        // [MJ: mju_decodePyramid] normal force and the two friction components
        const T fn = f[0] + f[1] + f[2] + f[3];
        const T f1 = con_mu * (f[0] - f[1]), f2 = con_mu * (f[2] - f[3]);
        T F[3];
#pragma unroll
        for (int k = 0; k < 3; k++) F[k] = fn * con_n[k] + f1 * con_t1[k] + f2 * con_t2[k];
        // the force acts on body B (geom 2), its reaction on body A; keys carry no sensors
#pragma unroll
        for (int side = 0; side < 2; side++) {
          const int l = side ? bodyB : bodyA;
          if (l >= 0 && l < RPK_KEYBASE) {
            const T sg = side ? (T)1 : (T)-1;
            const T* tr = M.tree_ref() + 3 * M.link_tree()[l];
            const T r[3] = {con_pos[0] - tr[0], con_pos[1] - tr[1], con_pos[2] - tr[2]};
            T t[3];
            cross3(t, r, F);
#pragma unroll
            for (int k = 0; k < 3; k++) { lds_add(&sm.fext[l][k], sg * t[k]); lds_add(&sm.fext[l][3 + k], sg * F[k]); }
          }
        }
        if (fn > (T)0) {
          for (int st = 0; st < M.nsite; st++) {
            const T rad = M.site_touch_radius()[st];
            const int sl = M.site_link()[st];
            if (rad > (T)0 && (sl == bodyA || sl == bodyB)) {
              // ray from the contact point along the normal force (flipped when the sensor is on
              // body B) against the site's sphere [MJ: mju_rayGeom]
              T sp[3];
              mat_vec(sp, sm.xmat[sl], M.site_pos() + 3 * st);
              const T o[3] = {con_pos[0] - sm.xpos[sl][0] - sp[0], con_pos[1] - sm.xpos[sl][1] - sp[1],
                              con_pos[2] - sm.xpos[sl][2] - sp[2]};
              const T bq = (sl == bodyB ? (T)-1 : (T)1) * dot3(o, con_n), cq = dot3(o, o) - rad * rad;
              const T det = bq * bq - cq;
              if (det >= (T)1e-15 && -bq + N::sqrt(det) >= (T)0) lds_add(&sm.touch[st], fn);
            }
          }
        }
      }
      WSYNC();
      // spatial accelerations with the constrained qacc, by tree level
      const T qacc = qw[0];
      T ca2[6] = {0, 0, 0, 0, 0, 0};
      for (int d = 0; d < M.maxdepth; d++) {
        if (isl && depth == d) {
          // (gravity is NOT scaled here: gravity compensation is a joint-space passive force in
          // MuJoCo, the body-level accelerations see the full gravity)
          T pa[6] = {0, 0, 0, -M.gx, -M.gy, -M.gz};
          if (parent >= 0) {
#pragma unroll
            for (int k = 0; k < 6; k++) pa[k] = sm.vel[parent][k];
          }
#pragma unroll
          for (int k = 0; k < 6; k++) ca2[k] = pa[k] + cdd[k] * qd[0] + cdofr[k] * qacc;
        }
        WSYNC();
        if (isl && depth == d) {
#pragma unroll
          for (int k = 0; k < 6; k++) sm.vel[lane][k] = ca2[k];
        }
        WSYNC();
      }
      {
        T f1[6], iv[6], f2[6], t1[3], t2[3];
        T cin[10];
        link_inertia(cin);
        mul_inert(f1, cin, ca2);
        mul_inert(iv, cin, cv);
        cross3(t1, cv, iv); cross3(t2, cv + 3, iv + 3);
        f2[0] = t1[0] + t2[0]; f2[1] = t1[1] + t2[1]; f2[2] = t1[2] + t2[2];
        cross3(f2 + 3, cv, iv + 3);
        if (isl) {
#pragma unroll
          for (int k = 0; k < 6; k++) sm.acc[lane][k] = f1[k] + f2[k] - sm.fext[lane][k];
        }
      }
      WSYNC();
      for (int d = M.maxdepth - 1; d >= 1; d--) {
        int mr = M.level_maxrank()[d];
        for (int r = 0; r < mr; r++) {
          if (isl && depth == d && sibrank == r) {
#pragma unroll
            for (int k = 0; k < 6; k++) sm.acc[parent][k] += sm.acc[lane][k];
          }
          WSYNC();
        }
      }
      if (isl && S.sens_torque) {
        // interaction force of my body with its parent = my subtree force (the massless virtual links
        // of a multi-joint body carry the same one); moment about the body origin, on the joint axis
        const int bl = M.link_bodylink()[L];
        T axb[3], t[3];
        const T* tref = M.tree_ref() + 3 * ltree;
        mat_vec(axb, sm.xmat[bl], M.link_axis() + 3 * L);
        const T r[3] = {tref[0] - sm.xpos[bl][0], tref[1] - sm.xpos[bl][1], tref[2] - sm.xpos[bl][2]};
        const T* fi = sm.acc[lane];
        cross3(t, r, fi + 3);
        S.sens_torque[eo + ldof] = (fi[0] + t[0]) * axb[0] + (fi[1] + t[1]) * axb[1] + (fi[2] + t[2]) * axb[2];
      }
      if (lane < M.nsite && S.sens_touch) S.sens_touch[(size_t)env * M.nsite + lane] = sm.touch[lane];
    }
    }  // PART != 1
    }  // MODE 0
  }

  PROF(17);
  // ------------------------------------------------------------------ outputs
  if constexpr (MODE == 0 && PART != 1) {
  if (!(S.stale_outputs && substep == nsub - 1)) {   // (legacy_step = False: see RpState::stale_outputs)
#pragma unroll
  for (int s = 0; s < 2; s++) if (isk[s]) {
    if (kact[s] >= 0) S.act_vel[(size_t)env * nu + kact[s]] = M.act_coef()[2 * kact[s]] * qd[1 + s];
  }
  if (isa) S.act_vel[(size_t)env * nu + lane] = avel;
  if (lane < RPK_NCOUT) {
    bool v = lane < ncon;
    int ga_ = -1, gb_ = -1;
    if (v) {
      if (RpCaps<T>::NC <= RPK_NCL || lane < RPK_NCL) { ga_ = sm.cgA[lane]; gb_ = sm.cgB[lane]; }
      else {
        const int* oi_ = B.covi + ((size_t)env * (RPK_NC - RPK_NCL) + (lane - RPK_NCL)) * 4;
        ga_ = oi_[2]; gb_ = oi_[3];
      }
    }
    S.contact_geoms[((size_t)env * RPK_NCOUT + lane) * 2] = ga_;
    S.contact_geoms[((size_t)env * RPK_NCOUT + lane) * 2 + 1] = gb_;
    S.contact_dist[(size_t)env * RPK_NCOUT + lane] = v ? con_dist : (T)0;
  }
  }
  }
  if constexpr (MODE != 2) {
    int w = warn;
    w = wave_or(w);
    if (lane == 0) {
      S.warn[env] |= w;
      if constexpr (MODE == 0 && PART != 1) { if (!(S.stale_outputs && substep == nsub - 1)) S.ncon[env] = ncon; }
      if constexpr (MODE == 1) {
        S.solver_iter[env] = (niter_last & 255) | ((__popcll(dirty_mask) & 255) << 8) | ((nkt & 255) << 16);
        S.time[env] = time;
      }
    }
  }
  if (S.prof && env == 0 && lane < RPK_NPROF_STAGE) {
    WSYNC();
    atomicAdd((unsigned long long*)&S.prof[lane], (unsigned long long)sm.prof[lane]);
  }
  if constexpr (MODE != 2) {
    int* cost = MODE == 0 ? S.cost_pos : S.cost_sol;
    if (cost && lane == 0) cost[env] = (int)(((long long)__builtin_readcyclecounter() - kernel_t0) >> 8);
  }
#undef LF
#undef LI
}

template <typename T, int MODE, int FIXED_TL = 0, int MD = RPK_MAXD, int MESH = 0>
__global__ __launch_bounds__(64, (MODE != 1 || sizeof(T) == 4) ? 2 : RPK_SOL64_WAVES) void rp_stage_kernel(RpModel<T> M, RpState<T> S, RpStage<T> B, int substep,
                                                     int nsub) {
  // One env per workgroup -- or, for the full-capacity solver stage next to the lean one, a small grid walking
  // the compacted list of the envs outside the light class (RpState::heavy_list).
  const bool listed = MODE == 1 && S.heavy_list != nullptr;
  const int n = listed ? *(volatile const int*)S.heavy_cnt : (int)gridDim.x;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int env = listed ? S.heavy_list[i] : (S.order ? S.order[S.env_base + i] : S.env_base + i);
    rp_stage_body<T, MODE, FIXED_TL, MD, MESH, false>(M, S, B, substep, nsub, env, nullptr, (int)threadIdx.x);
    if (!listed) break;
    __syncthreads();
  }
  if constexpr (MODE == 1) {
    // (the list's last reader clears it: this stage, or the position stage that follows it on the same stream,
    // rp_pos_list_kernel)
    if (listed && !S.heavy_keep && threadIdx.x == 0) {
      __threadfence();
      if (atomicAdd(S.heavy_done, 1) == (int)gridDim.x - 1) {
        if (S.heavy_peak && n > *S.heavy_peak) *S.heavy_peak = n;
        *S.heavy_cnt = 0; *S.heavy_done = 0; __threadfence();
      }
    }
  }
}

// The split position / velocity stage (round 5): front part -> rp_narrow_kernel (rp_collide.hpp) -> back part.  One env
// per workgroup, as rp_stage_kernel<T, 0>; the back part's first workgroup clears the pooled lists' counters for the
// next front launch on this stream (the narrow-phase launch between them has finished reading them by then).
template <typename T, int MESH>
__global__ __launch_bounds__(64, 2) void rp_pos_front_kernel(RpModel<T> M, RpState<T> S, RpStage<T> B, int substep, int nsub) {
  const int env = S.order ? S.order[S.env_base + blockIdx.x] : S.env_base + (int)blockIdx.x;
  rp_stage_body<T, 0, 0, RPK_MAXD, MESH, false, 1>(M, S, B, substep, nsub, env, nullptr, (int)threadIdx.x);
}
template <typename T, int MESH>
__global__ __launch_bounds__(64, 2) void rp_pos_back_kernel(RpModel<T> M, RpState<T> S, RpStage<T> B, int substep, int nsub) {
  if (blockIdx.x == 0 && threadIdx.x < RPK_NSTRIPE * RPK_NTYPE_PAD) B.tcount[B.tcount_off + threadIdx.x] = 0;
  const int env = S.order ? S.order[S.env_base + blockIdx.x] : S.env_base + (int)blockIdx.x;
  rp_stage_body<T, 0, 0, RPK_MAXD, MESH, false, 2>(M, S, B, substep, nsub, env, nullptr, (int)threadIdx.x);
}

// The position / velocity stage of the envs on the compacted list (the envs outside the light class), on the companion
// stream right behind their full-capacity solver stage; the slice's own position launch skips them (RpState::skip_heavy).
// A kernel of its own: with this loop around the body, rp_stage_kernel<T, 0> itself -- the launch every env of the
// benchmark goes through -- spilled 500 registers instead of 340 (loop-invariant per-lane state hoisted out of the loop).
template <typename T, int MESH>
__global__ __launch_bounds__(64, 2) void rp_pos_list_kernel(RpModel<T> M, RpState<T> S, RpStage<T> B, int substep, int nsub) {
  const int n = *(volatile const int*)S.heavy_cnt;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    int lane = (int)threadIdx.x;
    asm volatile("" : "+v"(lane));
    rp_stage_body<T, 0, 0, RPK_MAXD, MESH, false>(M, S, B, substep, nsub, S.heavy_list[i], nullptr, lane);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(S.heavy_done, 1) == (int)gridDim.x - 1) {
      if (S.heavy_peak && n > *S.heavy_peak) *S.heavy_peak = n;
      *S.heavy_cnt = 0; *S.heavy_done = 0; __threadfence();
    }
  }
}

// ============================================================================
// Fused substeps.  One wave keeps its env through ALL substeps of an rp_step:
//   n_sub x (lean solver stage; position stage)
// in one launch -- the stages are the bodies above, run in turn over one LDS allocation (both need ~20 KB and
// 256 registers: the fused kernel has the occupancy of either).  With one launch per stage, every launch ended
// with most SIMDs waiting for the few heaviest envs (measured: 38 % of the wave slots idle over an rp_step);
// over ten substeps an env's cost averages out, and nothing waits at a launch boundary any more.
// An env that leaves the light capacity class at substep k stops there (hdr[7] = k, appended to the list);
// rp_cleanup_steps_kernel, launched right after, takes such envs through their remaining substeps with the
// full-capacity solver stage.  The hand-over between the two stages is global memory written and read by the
// same wave: a fence + L1 invalidate separates them.
// ============================================================================
template <typename T>
__device__ __forceinline__ void rp_save_prev_state(const RpModel<T>& M, const RpState<T>& S, const int env) {
  // the state the last substep's forces belong to (acceleration-stage sensors)
  for (int i = threadIdx.x; i < M.nv; i += 64) {
    S.qpos_prev[(size_t)env * M.nv + i] = S.qpos[(size_t)env * M.nv + i];
    S.qvel_prev[(size_t)env * M.nv + i] = S.qvel[(size_t)env * M.nv + i];
  }
}
// (workgroup scope: writer and reader are the same wave, behind the same vector L1, which a CU's own stores keep
// coherent; an agent-scope fence writes the L2 back and invalidates it -- 20 times per wave and step it cost
// more than the launch tails the fusion removes)

template <typename T, int MESH>
__global__ __launch_bounds__(64, 2) void rp_fused_steps_kernel(RpModel<T> M, RpState<T> S, RpStage<T> B, int nsub) {
  using namespace rpk;
  constexpr size_t NB = sizeof(SmemLean<T>) > sizeof(Smem<T, 0, RPK_MAXD>) ? sizeof(SmemLean<T>) : sizeof(Smem<T, 0, RPK_MAXD>);
  __shared__ __attribute__((aligned(16))) unsigned char smem[NB];
  const int env = S.order ? S.order[S.env_base + blockIdx.x] : S.env_base + (int)blockIdx.x;
  if (S.active && S.active[env] == 0) return;
  for (int k = 0; k < nsub; k++) {
    // (the lane index goes through an opaque copy every trip: otherwise the per-lane addresses and constants of
    // BOTH stages are loop invariants, hoisted out of the substep loop and held in registers across it -- 244
    // spilled registers)
    int lane = (int)threadIdx.x;
    asm volatile("" : "+v"(lane));
    if (*(volatile const int*)&B.hdr[env * 8 + 6] != 1) {
      // outside the light class from here on: the clean-up launch continues with substep k
      if (threadIdx.x == 0) { B.hdr[env * 8 + 7] = k; S.heavy_list[atomicAdd(S.heavy_cnt, 1)] = env; }
      return;
    }
    if (S.qpos_prev && k == nsub - 1) rp_save_prev_state(M, S, env);
    rp_lean_solver_body<T, true>(M, S, B, env, smem, lane);
    RPK_STAGE_FENCE();
    asm volatile("" : "+v"(lane));
    rp_stage_body<T, 0, 0, RPK_MAXD, MESH, true>(M, S, B, k, nsub, env, smem, lane);
    RPK_STAGE_FENCE();
  }
}

// Round 6: the same with the SPLIT position stage's bodies -- front part, the env's narrow phase in its own wave (one real
// call per pair type: rp_narrow_env), back part.  rp_fused_steps_kernel runs the one-kernel position body, which in the
// hull builds spills 276 registers (the inlined portal refinement); the front and back parts have no scratch at all
// (155 / 194 VGPRs) and the narrow-phase routines are calls that nothing lives across.  Same bodies on the same inputs as
// the per-stage split schedule: same bits.  (B.tlist is null in this schedule: the front part writes the env's candidate
// list only.)
template <typename T, int MESH>
__global__ __launch_bounds__(64, 2) void rp_fused_split_kernel(RpModel<T> M, RpState<T> S, RpStage<T> B, int nsub) {
  using namespace rpk;
  constexpr size_t NA = sizeof(SmemLean<T>) > sizeof(Smem<T, 0, RPK_MAXD>) ? sizeof(SmemLean<T>) : sizeof(Smem<T, 0, RPK_MAXD>);
  constexpr size_t NB = NA > sizeof(SmemFront<T, RPK_MAXD>) ? NA : sizeof(SmemFront<T, RPK_MAXD>);
  __shared__ __attribute__((aligned(16))) unsigned char smem[NB];
  const int env = S.order ? S.order[S.env_base + blockIdx.x] : S.env_base + (int)blockIdx.x;
  if (S.active && S.active[env] == 0) return;
  for (int k = 0; k < nsub; k++) {
    int lane = (int)threadIdx.x;
    asm volatile("" : "+v"(lane));
    if (*(volatile const int*)&B.hdr[env * 8 + 6] != 1) {
      if (threadIdx.x == 0) { B.hdr[env * 8 + 7] = k; S.heavy_list[atomicAdd(S.heavy_cnt, 1)] = env; }
      return;
    }
    if (S.qpos_prev && k == nsub - 1) rp_save_prev_state(M, S, env);
    rp_lean_solver_body<T, true>(M, S, B, env, smem, lane);
    RPK_STAGE_FENCE();
    asm volatile("" : "+v"(lane));
    rp_stage_body<T, 0, 0, RPK_MAXD, MESH, true, 1>(M, S, B, k, nsub, env, smem, lane);
    RPK_STAGE_FENCE();
    asm volatile("" : "+v"(lane));
    {
      // (the routines are real calls taking the three parameter blocks by reference: COPIES go there -- a kernel argument
      // whose address escapes is moved to scratch for the whole kernel, and every table pointer of the stage bodies,
      // which must stay in scalar registers (`fresh`), would be reloaded from there into vector registers)
      const RpModel<T> Mn = M; const RpState<T> Sn = S; const RpStage<T> Bn = B;
      rp_narrow_env<T, MESH>(Mn, Sn, Bn, env, lane);
    }
    RPK_STAGE_FENCE();
    asm volatile("" : "+v"(lane));
    rp_stage_body<T, 0, 0, RPK_MAXD, MESH, true, 2>(M, S, B, k, nsub, env, smem, lane);
    RPK_STAGE_FENCE();
  }
}

template <typename T, int MESH, int FIXED_TL>
__global__ __launch_bounds__(64, 1) void rp_cleanup_steps_kernel(RpModel<T> M, RpState<T> S, RpStage<T> B, int nsub) {
  using namespace rpk;
  constexpr size_t NA = sizeof(Smem<T, 1, RPK_MAXD>) > sizeof(Smem<T, 0, RPK_MAXD>) ? sizeof(Smem<T, 1, RPK_MAXD>) : sizeof(Smem<T, 0, RPK_MAXD>);
  constexpr size_t NB = NA > sizeof(SmemLean<T>) ? NA : sizeof(SmemLean<T>);
  __shared__ __attribute__((aligned(16))) unsigned char smem[NB];
  const int n = *(volatile const int*)S.heavy_cnt;
  RpState<T> Sh = S;   // the full-capacity solver stage takes the env whatever its class
  Sh.lean = 0; Sh.heavy_list = nullptr;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int env = S.heavy_list[i];
    for (int k = *(volatile const int*)&B.hdr[env * 8 + 7]; k < nsub; k++) {
      int lane = (int)threadIdx.x;
      asm volatile("" : "+v"(lane));
      if (S.qpos_prev && k == nsub - 1) rp_save_prev_state(M, S, env);
      // the same solver build per substep as the per-stage schedule picks (light again: the lean stage; FIXED_TL as
      // the host picks it there), so that both schedules produce the same bits
      if (*(volatile const int*)&B.hdr[env * 8 + 6] == 1) rp_lean_solver_body<T, true>(M, S, B, env, smem, lane);
      else rp_stage_body<T, 1, FIXED_TL, RPK_MAXD, 0, true>(M, Sh, B, k, nsub, env, smem, lane);
      RPK_STAGE_FENCE();
      asm volatile("" : "+v"(lane));
      rp_stage_body<T, 0, 0, RPK_MAXD, MESH, true>(M, S, B, k, nsub, env, smem, lane);
      RPK_STAGE_FENCE();
    }
  }
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(S.heavy_done, 1) == (int)gridDim.x - 1) {
      if (S.heavy_peak && n > *S.heavy_peak) *S.heavy_peak = n;
      *S.heavy_cnt = 0; *S.heavy_done = 0; __threadfence();
    }
  }
}
