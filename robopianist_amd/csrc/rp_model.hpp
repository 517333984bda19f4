// rp_kernels.hpp — fixed-topology batched rigid-body step for RoboPianist on gfx950.
//
// One environment per 64-lane wavefront (one workgroup = one wave).  Lane roles:
//   lane i < nlink        : hand link i (= hand dof i; 2 trees x 26 links)
//   lane k, k+64          : piano keys k and k+64 (closed-form 1-dof hinges)
//   lane c < ncon         : contact c (4 pyramidal rows) during the solve
//   lane a < nu           : actuator a during transmission/actuation
//   lane nlink+s          : solver slot of the s-th key currently touched by a hand
// Per-dof vectors live in registers of their owner lane; LDS holds link frames,
// the packed joint-space matrices, contact Jacobians and small staging vectors.
//
// What the reference does here: `physics.step()` x n_substeps inside
// dm_control's composer.Environment.step, configured by
// /root/reference/robopianist/suite/tasks/base.py:28,31,68-70 and reached from
// suite/__init__.py:87-93.  The arithmetic follows MuJoCo's documented
// pipeline; the CPU restatement used as the parity
// oracle is oracle/rp_oracle.c (a generic, sequential, dense-J formulation).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RPK_WAVE 64
#define RPK_NC 64        // max contacts kept per env (= RP_MAX_CONTACTS; contact c lives in lane c; fp64 -- see RpCaps)
#define RPK_NCOUT 64     // == RP_MAX_CONTACTS
#ifndef RPK_NCL          // (tests shrink it to exercise the overflow records with small scenes)
#define RPK_NCL 32       // contacts staged in the position stage's LDS; contacts RPK_NCL.. go through RpStage::covf / covi
#endif
#ifndef RPK_NE
#define RPK_NE 640      // max contact Jacobian entries (contact, dof) handed to the solver (fp64; see RpCaps)
#endif
// Contact Jacobian entry records (RpStage::entM) and the per-contact entry range (lane field LI(10)):
//   entM[.][0] = dof lane | contact << 6 | column of the dof in its row layout << 12 | cross-chain << 16
//   entM[.][1] = first entry of the contact | entries of the contact << 12 | rank of this entry << 18
//   LI(10)     = first entry | entries << 12
#define RPK_EM0(lane_, con_, col_, cross_) ((lane_) | ((con_) << 6) | ((col_) << 12) | ((cross_) << 16))
#define RPK_EM1(base_, cnt_, rank_) ((base_) | ((cnt_) << 12) | ((rank_) << 18))
#define RPK_EM_LANE(m0_) ((m0_) & 63)
#define RPK_EM_CON(m0_) (((m0_) >> 6) & 63)
#define RPK_EM_COL(m0_) (((m0_) >> 12) & 15)
#define RPK_EM_CROSS(m0_) (((m0_) >> 16) & 1)
#define RPK_EM_BASE(m1_) ((m1_) & 4095)
#define RPK_EM_CNT(m1_) (((m1_) >> 12) & 63)
#define RPK_EM_RANK(m1_) (((m1_) >> 18) & 63)
#define RPK_NBOXF 32     // boxes / hull boxes covered by the oriented-box prefilter
#ifndef RPK_HMAX
#define RPK_HMAX 60      // max rows of the dense cross-coupling block (+1 row for its rhs = one lane each; fp64)
#endif
#define RPK_WORK 128     // narrow-phase work list
// Split position stage (round 5: rp_pos_front_kernel -> rp_narrow_kernel -> rp_pos_back_kernel): per env, the
// candidates that passed the fp32 prefilters and the narrow-phase results they map to
#define RPK_NCAND 256    // candidates per env and mj_step (typical: 20-30); beyond: RP_WARN_WORK_FULL
#define RPK_NRES 384     // result records per env: 1 per capsule-capsule / hull pair, 2 per capsule-box, 8 per box-box
// pooled narrow-phase lists: 0 capsule-capsule, 1 capsule-box, 2 box-box, 3 .. 6 hull pairs (MPR) in four buckets:
// 3 + (side A is a hull too) + 2 * (the hull of side B uses another vertex set than the model's last hull) -- a wave
// walking the portal refinement scans, on every trip, every vertex set ANY of its lanes needs on either side, so lanes
// that need the same scans belong together (any bucketing is correct; this one fits "two fingertip shapes")
#define RPK_NTYPE 7
#define RPK_NTYPE_PAD 8  // counters per slice and stripe
// every list is kept in RPK_NSTRIPE stripes (env & 7): the front part's waves -- thousands per launch -- reserve their
// entries with one returning atomic per list, and on ONE counter per list those serialise (measured: 29 k cycles per
// wave for the hand-over block, most of it waiting for the atomics)
#define RPK_NSTRIPE 8
#ifndef RPK_MPR_LANES    // candidates per wave on the hull lists (the wave walks as many trips as its slowest lane needs)
#define RPK_MPR_LANES 64
#endif
#define RPK_NFRAME 18    // per-link frame fields handed from the front to the back part: xpos 3, xmat 9, xaxis 3, xanchor 3
#define RPK_KLIST 192    // geom-key candidates per mj_step (typical: 8.5; with RPK_GLIST and the box table inside the 2496 B the link table leaves)
#ifndef RPK_GLIST        // (tests shrink it to exercise the refill)
#define RPK_GLIST 256    // compacted sphere-overlap candidates per refill of the drain rounds (typical: 185 per mj_step)
#endif
#define RPK_MAXD 9       // tree depth levels held by the default kernel builds (trunk <= 4 links + chain <= 5)
#define RPK_MAXD_DEEP 13 // ... by the deep builds (trunk <= 8 links: every subset of the six forearm dofs)
#define RPK_NL 52        // max links held by the default kernel builds (two hands x (24 + 2 forearm dofs))
#define RPK_NL_DEEP 60   // ... by the deep builds (two hands x (24 + 6)); also the table capacity
#define RPK_NLX(MD) ((MD) > RPK_MAXD ? RPK_NL_DEEP : RPK_NL)
#define RPK_NKEYS 128    // max keys (2 slots per lane)
#define RPK_KEYBASE 1000 // work-list / contact encoding of "key k" = RPK_KEYBASE + k

#define JNT_SLIDE_ 2
#define JNT_HINGE_ 3
#define GEOM_CAPSULE_ 3
#define GEOM_CYL_ 5    // cylinder: geom_size = (radius, radius, half height) -- its bounding box, as for hulls; MESH = 2 builds only
#define GEOM_BOX_ 6
#define GEOM_MESH_ 7   // convex hull (vertices in the geom frame)
#define RPK_MAXMESHV 320 // vertices of all SCANNED hulls together (hulls with a vertex graph: RpModel::hull_vert, any size)
#define RPK_HULL_GRAPH_ROW 24

// All model tables live in TWO device arrays (one of T, one of int) at compile-time
// offsets (tables are padded to their maximum item counts).  A table access is then
// `base pointer + immediate`, instead of one kernarg pointer load per table: with ~80
// tables the pointers do not fit in SGPRs, and every re-load costs an s_waitcnt
// lgkmcnt(0) that also drains the LDS queue.
#define RPK_MAXACT 192
#define RPK_MAXTREE 4
//      name            items        stride
#define RPK_FTABLES(X) \
  X(link_lpos,        RPK_NL_DEEP,      3) X(link_lmat,       RPK_NL_DEEP,      9) X(link_axis,      RPK_NL_DEEP, 3) \
  X(link_anchor,      RPK_NL_DEEP,      3) X(link_mass,       RPK_NL_DEEP,      1) X(link_ipos,      RPK_NL_DEEP, 3) \
  X(link_inertia,     RPK_NL_DEEP,      6) X(link_invw_body,  RPK_NL_DEEP,      1) X(link_armature,  RPK_NL_DEEP, 1) \
  X(link_damping,     RPK_NL_DEEP,      1) X(link_stiffness,  RPK_NL_DEEP,      1) X(link_springref, RPK_NL_DEEP, 1) \
  X(link_floss,       RPK_NL_DEEP,      1) X(link_fl_R,       RPK_NL_DEEP,      1) X(link_fl_B,      RPK_NL_DEEP, 1) \
  X(link_range,       RPK_NL_DEEP,      2) X(link_lim_K,      RPK_NL_DEEP,      1) X(link_lim_B,     RPK_NL_DEEP, 1) \
  X(link_lim_solimp,  RPK_NL_DEEP,      5) X(link_invw_dof,   RPK_NL_DEEP,      1) X(link_act_coef,  RPK_NL_DEEP, 1) \
  X(link_gscale,      RPK_NL_DEEP,      1) \
  X(tree_gscale,      RPK_MAXTREE, 1) X(tree_ref,        RPK_MAXTREE, 3) \
  X(key_pos,          RPK_NKEYS,   3) X(key_half,        RPK_NKEYS,   3) X(key_mass,       RPK_NKEYS, 1) \
  X(key_M,            RPK_NKEYS,   1) X(key_stiffness,   RPK_NKEYS,   1) X(key_springref,  RPK_NKEYS, 1) \
  X(key_damping,      RPK_NKEYS,   1) X(key_range,       RPK_NKEYS,   2) X(key_lim_K,      RPK_NKEYS, 1) \
  X(key_lim_B,        RPK_NKEYS,   1) X(key_lim_solimp,  RPK_NKEYS,   5) X(key_invw_dof,   RPK_NKEYS, 1) \
  X(key_invw_body,    RPK_NKEYS,   1) X(key_rbound,      RPK_NKEYS,   1) X(key_cparam,     1,         8) \
  X(geom_size,        RPK_WAVE,    3) X(geom_pos,        RPK_WAVE,    3) X(geom_mat,       RPK_WAVE,  9) \
  X(geom_rbound,      RPK_WAVE,    1) X(geom_invw,       RPK_WAVE,    1) X(geom_cparam,    RPK_WAVE,  8) \
  X(act_coef,         RPK_MAXACT,  2) X(act_gain,        RPK_MAXACT,  1) X(act_bias,       RPK_MAXACT, 3) \
  X(act_ctrlrange,    RPK_MAXACT,  2) X(act_forcerange,  RPK_MAXACT,  2) X(site_pos,       RPK_WAVE,  3) \
  X(site_touch_radius, RPK_WAVE,   1) X(mesh_vert, RPK_MAXMESHV, 3) X(geom_bcap, RPK_WAVE, 2)
#define RPK_ITABLES(X) \
  X(lane_topo,    RPK_NL_DEEP, 16) \
  X(link_parent,  RPK_NL_DEEP, 1) X(link_depth,   RPK_NL_DEEP, 1) X(link_tree,    RPK_NL_DEEP, 1) X(link_jtype,  RPK_NL_DEEP, 1) \
  X(link_dof,     RPK_NL_DEEP, 1) X(link_sibrank, RPK_NL_DEEP, 1) X(link_limited, RPK_NL_DEEP, 1) X(link_act,    RPK_NL_DEEP, 1) \
  X(link_ndesc,   RPK_NL_DEEP, 1) X(link_anc,     RPK_NL_DEEP, RPK_MAXD_DEEP) X(link_ancmask, RPK_NL_DEEP, 2) \
  X(link_desc,    RPK_NL_DEEP, RPK_MAXD_DEEP * 5) X(level_maxrank, 1, RPK_MAXD_DEEP) \
  X(tree_base,    RPK_MAXTREE, 1) X(tree_trunk, RPK_MAXTREE, 1) X(chain_first, RPK_MAXTREE, 5) \
  X(chain_len,    RPK_MAXTREE, 5) \
  X(key_dof,      RPK_NKEYS, 1) X(key_act,   RPK_NKEYS, 1) X(key_geomid, RPK_NKEYS, 1) \
  X(geom_link,    RPK_WAVE, 1) X(geom_type,  RPK_WAVE, 1) X(geom_modelid, RPK_WAVE, 1) \
  X(geom_pairmask, RPK_WAVE, 2) X(geom_iskeycap, RPK_WAVE, 1) \
  X(act_kind,     RPK_MAXACT, 1) X(act_lane, RPK_MAXACT, 2) X(act_ctrllimited, RPK_MAXACT, 1) \
  X(act_forcelimited, RPK_MAXACT, 1) X(site_link, RPK_WAVE, 1) X(link_bodylink, RPK_NL_DEEP, 1) \
  X(geom_vertadr, RPK_WAVE, 1) X(geom_vertnum, RPK_WAVE, 1) X(geom_vertflip, RPK_WAVE, 1) X(geom_vertgraph, RPK_WAVE, 1)

struct RpLayout {
  enum : int {
#define X(name, items, stride) F_##name, F_##name##_end = F_##name + (items) * (stride) - 1,
    RPK_FTABLES(X)
#undef X
    F_TOTAL,
#define X(name, items, stride) I_##name, I_##name##_end = I_##name + (items) * (stride) - 1,
    RPK_ITABLES(X)
#undef X
    I_TOTAL
  };
};

// Solver capacities per precision.  The fp32 solver stage is sized to fit two workgroups per
// SIMD (LDS <= 20 KB, 256 registers): measured 0.50 -> 0.35 ms per launch.  The fp64 stage
// cannot (40 KB, ~500 registers) and keeps the larger capacities.
template <typename T>
struct RpCaps {
  static constexpr int NE = sizeof(T) == 4 ? 240 : RPK_NE;
  static constexpr int HMAX = sizeof(T) == 4 ? 56 : RPK_HMAX;
  // contacts per env.  fp64: one per lane -- the envs beyond the light class run the full-capacity solver stage on a
  // small grid (rp_order_kernel's compacted list), so its LDS (57 KB) costs no occupancy that matters; the position
  // stage stages the first RPK_NCL contacts in LDS and the rest in global overflow records.
  static constexpr int NC = sizeof(T) == 4 ? 32 : RPK_NC;
};

template <typename T>
struct RpModel {
  int nlink, ntree, maxdepth, nkey, ngeom, nu, nsite, nv;
  int iterations, ls_iterations;
  T timestep, gx, gy, gz, tolerance, ls_tolerance, meaninertia, key_zmax;
  // sqrt(1 / opt.impratio): a contact's regularised friction coefficient is friction * mu_scale, its pyramid edges get
  // Rpy = 2 (mu * mu_scale)^2 R [MJ: mj_makeImpedance]; the Jacobian rows keep the friction coefficient itself
  T mu_scale;
  // stopping tolerance of the portal refinement (hull / cylinder pairs): MuJoCo's uniform 1e-6 by default; mpr_tol_poly
  // applies to POLYTOPE pairs (box / hull on both sides) and differs only after rp_set_mpr_tolerance
  T mpr_tol, mpr_tol_poly;
  const T* ft;    // RpLayout::F_* offsets
  const int* it;  // RpLayout::I_* offsets
  // hulls with a vertex graph (model/hull.py: more than 32 vertices -- the real hand's forearm / wrist / palm / thumb
  // meshes): vertices [n][3] and graph rows [n][RPK_HULL_GRAPH_ROW] = degree, neighbours; any size, may be null
  const T* hull_vert;
  const int* hull_graph;
#define X(name, items, stride) \
  __device__ __forceinline__ const T* name() const { return ft + RpLayout::F_##name; }
  RPK_FTABLES(X)
#undef X
#define X(name, items, stride) \
  __device__ __forceinline__ const int* name() const { return it + RpLayout::I_##name; }
  RPK_ITABLES(X)
#undef X
  __device__ __forceinline__ const unsigned* link_ancmask_u() const {
    return (const unsigned*)(it + RpLayout::I_link_ancmask);
  }
};

template <typename T>
struct RpState {
  int nenv;
  T *qpos, *qvel, *warm, *ctrl, *qfrc_applied, *time, *tree_offset;
  T *act_force, *act_vel, *site_xpos, *contact_dist;
  int *ncon, *contact_geoms, *warn, *solver_iter;
  const int* active;    // may be null: envs with active[e]==0 are left untouched
  uint32_t* key_trace;  // may be null
  long long* prof;      // may be null: per-phase cycle counters (env 0)
  int max_newton, max_ls;
  // acceleration-stage sensors (all may be null): contact row forces written by the solver stage,
  // joint torque / touch sensors written by the sensor stage (MODE 2)
  T *con_force, *sens_torque, *sens_touch;
  // may be null: workgroup b processes env order[b].  The host keeps it sorted by descending
  // predicted cost (longest-processing-time-first: with one wave per env and four sequential
  // rounds per SIMD, a heavy env that starts last is the tail of the whole launch)
  const int* order;
  int env_base;   // workgroup b of a launch stands for position env_base + b (a launch may cover a slice of the batch)
  // may be null: shader-clock cycles (>> 8) every env's wave spent in its last position stage /
  // solver stage -- the cost predictor of the ordered launch
  int *cost_pos, *cost_sol;
  // capacity classes of the solver stage: when set, the position stage marks every env whose constraint system
  // fits rpk::LeanCaps as "light" (hdr[6] = 1); rp_lean_solver_kernel (two waves per SIMD) steps those and the
  // full-capacity solver stage skips them
  int lean;
  // may be null: the envs of this launch outside the light class, compacted (rp_order_kernel builds the list from
  // the hand-over headers).  The full-capacity solver stage then runs as a small grid that walks the list -- its
  // workgroups need a whole SIMD's registers each, and a full grid of them, although all but a few leave at
  // once, waited for the lean launch next to it to drain (measured: the join cost up to 200 us per substep)
  int* heavy_list;
  int *heavy_cnt, *heavy_done;   // entries in the list; finished workgroups (the last one clears both)
  int* heavy_peak;               // may be null: the longest list seen (the host sizes that stage's grid from it, one step late)
  // The position / velocity stage of a substep may run as two launches: the slice's launch skips the envs outside the
  // light class (skip_heavy: hdr[6] of the substep just solved), a small grid on the companion stream walks the list
  // right behind the full-capacity solver stage (which then leaves the list alone: heavy_keep) -- the light envs'
  // position stage no longer waits for the slowest heavy solve.
  int skip_heavy, heavy_keep;
  // (skip_heavy tests this snapshot -- 1 = rp_order_kernel put the env on the list of the substep just solved -- not
  // hdr[6], which the list's own position stage rewrites on the companion stream while the slice's launch reads it)
  const unsigned char* listed;
  // dm_control's legacy_step = False (physics.step() = mj_step = mj_step1; mj_step2: position-dependent data is NOT
  // refreshed after the last integration): the position stage that follows the LAST substep's solve keeps its outputs
  // to itself (site positions, contacts, actuator velocities stay those of the state before that integration; the
  // hand-over for the next step is written as always)
  int stale_outputs;
  // fused substeps (rp_fused_steps_kernel): may be null -- where the state before the last substep's solver
  // stage goes (the acceleration-stage sensors belong to that state)
  T *qpos_prev, *qvel_prev;
};

// One workgroup == one wavefront, and a wave's LDS instructions execute in issue
// order, so cross-lane LDS hand-offs need no s_barrier and no s_waitcnt drain: only
// the compiler must be kept from reordering LDS accesses across the hand-off.
#define WSYNC()                                              \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
  } while (0)
// ... and a hand-off through GLOBAL memory between lanes of one wave (the stages of the fused-substeps kernels; the
// position stage's contact overflow records).  Workgroup scope: writer and reader sit behind the same vector L1,
// which a CU's own stores keep coherent; an agent-scope fence writes the L2 back and invalidates it.
#define RPK_STAGE_FENCE()                                     \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");    \
    __builtin_amdgcn_wave_barrier();                          \
  } while (0)
// The LDS block of a stage: its own static allocation, or (EXT) a buffer the calling kernel owns -- the
// fused-substeps kernels run two stages in turn over one allocation.
template <typename S_, bool EXT> __device__ __forceinline__ S_& rp_smem(void* ext) {
  if constexpr (EXT) return *reinterpret_cast<S_*>(ext);
  else { __shared__ S_ own; return own; }
}
#define RPK_NPROF 48        // counters the stage kernels keep (slots 32.. belong to the lean solver stage)
#define RPK_NPROF_ALL 64    // ... and the host reads: 48 + t / 56 + t = shader cycles / chunks of pooled list t (rp_narrow_kernel)
#define RPK_NPROF_STAGE 32  // ... of which the stage kernels use the first 32 (their LDS is full: 20480 B at two waves per SIMD)
// Per-phase shader-clock counters of env 0 (debug aid), accumulated in LDS so a probe
// costs about one LDS round trip; flushed to global memory once at kernel exit.
#ifdef RPK_MARK  // static analysis aid: phase boundaries as comments in the ISA listing
#define PROF(i) asm volatile("; MARK " #i)
#else
#define PROF(i)                                                         \
  do {                                                                  \
    if (S.prof && env == 0) {                                           \
      long long t_ = (long long)__builtin_readcyclecounter();           \
      if (lane == 0) sm.prof[i] += (unsigned)(t_ - prof_t);                       \
      prof_t = t_;                                                      \
    }                                                                   \
  } while (0)
#endif

// Hand-over between the position/velocity kernel (MODE 0) and the solver kernel
// (MODE 1): everything `mj_step1` leaves behind for `mj_step2`, per env.  Lives in
// HBM but is L2 / Infinity-Cache resident (<= 25 KB per env).
#define RPK_NLF 29  // per-lane float fields
#define RPK_NLI 12  // per-lane int fields
template <typename T>
struct RpStage {
  T* RM;      // [E][RPK_NLX(MD)][MD+1] mass-matrix rows (MD = RPK_MAXD or RPK_MAXD_DEEP, the build in use)
  T* lanef;   // [E][RPK_NLF][64]
  int* lanei; // [E][RPK_NLI][64]
  int* hdr;   // [E][8]: ncon, nkt, dirty mask lo/hi, nent, max entries per contact, capacity class (1 = light)
  T* entJ;    // [E][RpCaps<T>::NE][3]  contact Jacobian entries: d(contact point velocity)/d(qvel of one dof)
  int* entM;  // [E][RpCaps<T>::NE][2]  lane | contact<<6 | column<<11 | cross<<15 ; base | count<<8 | rank<<16
  int* slots; // [E][64]: slotkey[16], slotlink[16], slotmask lo[16], hi[16]
  int* keyslot; // [E][RPK_NKEYS/4] (packed signed char)
  // contacts RPK_NCL.. of the position stage (rare: hand-on-hand pile-ups), staged here instead of in LDS between the
  // narrow phase that emits them and the contact lanes that take them into registers
  T* covf;      // [E][RPK_NC - RPK_NCL][12]: pos[3], normal[3], dist, mu, kterm, B, D
  int* covi;    // [E][RPK_NC - RPK_NCL][4]: link A, link B (or RPK_KEYBASE + key), model geom ids
  // ---- split position stage (all null when the engine runs the position stage as one kernel).  The front part
  // (kinematics, CRB, broad phase, fp32 prefilters) leaves link frames, geom frames and the candidate list; the pooled
  // narrow phase (lane = candidate, whatever env it belongs to, one routine per wave) fills the result records; the
  // back part (constraint rows, Jacobians, velocity stage) collects them in the one-kernel stage's emission order.
  T* frames;      // [E][RPK_NFRAME][64]   lane = link
  int* cand;      // [E][RPK_NCAND][2]     ga | gb << 16 ; first result record | type << 16
  int* ncand;     // [E]                   candidates of this mj_step; -1: the front part did not run for this env
  T* cres;        // [E][RPK_NRES][12]     pos[3], normal[3], dist, mu, kterm, B, D (as covf)
  int* cres_n;    // [E][RPK_NCAND]        contacts of candidate i (0 .. 8)
  int* tlist;     // [RPK_NTYPE][RPK_NSTRIPE][ceil(E / 8) * RPK_NCAND][4]  env, ga | gb << 16, first result record, candidate index
  int* tcount;    // [slices][RPK_NSTRIPE][RPK_NTYPE_PAD]   entries per list (the back part's first workgroup clears them)
  int tcount_off; // this launch's slice: tcount + tcount_off
  size_t tstride; // ceil(E / 8) * RPK_NCAND: one stripe of one type (a slice's entries start at (env_base / 8) * RPK_NCAND)
  int* split_dropped;   // may be null: counts the envs whose candidate / record lists overflowed (RP_WARN_SPLIT_FULL)
};
