// rp_solver2.hpp — the LEAN solver stage (mj_step2: actuation, Newton solve, Euler) for envs whose
// constraint system fits the "light" capacity class.  Same arithmetic as rp_stage_kernel<T, 1>
// (rp_kernels.hpp: that build keeps the full capacities and takes the other envs), restructured so that
// the per-lane state is half as large: <= 256 registers and <= 20 KB of LDS per env, i.e. two waves per
// SIMD in fp64 (the full-capacity build needs all 512 registers and 40 KB).
//
// What the reference does here: `physics.step()` inside composer.Environment.step,
// /root/reference/robopianist/suite/tasks/base.py:28,31,68-70 (mj_step2 half of every substep);
// MuJoCo's mj_fwdActuation / mj_fwdAcceleration / mj_solNewton (PrimalSearch) / mj_Euler restated.
//
// How the state shrinks:
//   * tree factor/solve with every row in ITS OWN lane (`Rr[10]`): the links of one chain position are
//     eliminated for all chains at once, their rows cross to their in-chain ancestors through LDS
//     (per-lane addresses), the trunk receives sum_v H[v][t] H[v][t'] / d_v through LDS adds; no lane
//     gathers a chain block (Ac[5][9] + At + dT of the full build: ~140 registers in all 64 lanes);
//   * contact Jacobian entries are rotated into their contact frame once per launch: J x gives
//     (normal, tangent 1, tangent 2) directly, the per-contact Hessian weight is 5 numbers, and the
//     contact frame (9 values per lane) never sits in registers;
//   * the line search recomputes a row's quadratic coefficients from (jar, jv) where it evaluates them
//     (same expressions, same bits) instead of holding 21 coefficients;
//   * M rows stay in registers, M x scatters its column part with LDS adds (no LDS copy of M);
//     key values cross through 16 slot-indexed cells instead of a 128-key table;
//   * state that the Newton loop does not touch (qpos, qvel, applied forces) is re-read at the end.
#pragma once
#ifdef RP_LEAN_TRACE
#include <cstdio>
#endif
#include "rp_model.hpp"
#include "rp_wave.hpp"
#include "rp_dense.hpp"

namespace rpk {
// capacities of the light class (what the position stage tests before it marks an env "light")
#ifndef RPK_LEAN_NE   // (tests shrink it to send envs through the full-capacity stage's compacted list)
#define RPK_LEAN_NE 184
#endif
struct LeanCaps {
  static constexpr int NC = 24;     // contacts
  static constexpr int NE = RPK_LEAN_NE;    // contact Jacobian entries (what the 20 KB of LDS per wave leave room for; the config-2 replay peaks at 166)
  static constexpr int HMAX = 36;   // rows of the dense (cross-chain) block
  static constexpr int NK = 12;     // touched keys (solver slots)
};
template <typename T>
struct SmemLean {
  T R[RPK_WAVE][RPK_MAXD + 1];   // tree rows: assembly target, then the rows published by the elimination
  T H[(LeanCaps::HMAX + 1) * (LeanCaps::HMAX + 2) / 2];   // packed dense block + rhs row
  T entJ[LeanCaps::NE][3];       // contact-frame Jacobian entries (normal, tangent 1, tangent 2)
  int entM[LeanCaps::NE][2];
  union {
    struct {
      T cC[LeanCaps::NC][5];     // per-contact Hessian weight in the contact frame: sn, a1, a2, b1, b2
      T cv[LeanCaps::NC][3];     // per-contact staging (J x, or the contact force), contact frame
    };
    T stage[2 * 5][16];          // tree solve / M x: per chain (tree * 5 + chain), what it leaves on its trunk
  };
  T vec[RPK_WAVE];               // per-dof staging
  T xs[RPK_WAVE];                // solve staging: right-hand sides / solution
  T jt[RPK_WAVE];                // J^T f staging; pivots' reciprocals during a tree solve
  T slotv[2][16];                // values of the touched keys, by solver slot
  unsigned long long csup[LeanCaps::NC];   // per contact: the lanes (links, solver slot) of its Jacobian entries
  int cinf[LeanCaps::NC];        // per contact: first entry | entries << 8 | cross-chain << 16
  unsigned prof[RPK_NPROF];
};
}  // namespace rpk

template <typename T, bool EXT = false>
__device__ __forceinline__ void rp_lean_solver_body(const RpModel<T>& M, const RpState<T>& S, const RpStage<T>& B, const int env, void* ext, const int lane) {
  using namespace rpk;
  using N = Num<T>;
  constexpr int MD = RPK_MAXD, TC = 4;
  // (the env's header in two wide scalar loads; everything the prologue reads from the hand-over is requested further down
  // in ONE batch before the first value is used: fetched where they were used, these were some twenty dependent trips to L2)
  const int4 hq0 = *(const int4*)(B.hdr + (size_t)env * 8), hq1 = *(const int4*)(B.hdr + (size_t)env * 8 + 4);
  if (S.active && S.active[env] == 0) return;
  if (hq1.z != 1) return;   // not a light env: the full-capacity build takes it
  SmemLean<T>& sm = rp_smem<SmemLean<T>, EXT>(ext);
#ifdef RP_LEAN_TRACE
  if (lane == 0) printf("lean kernel: env %d ncon %d\n", env, B.hdr[env * 8]);
#endif
#ifdef RPK_POISON_LDS  // debug build: nothing may depend on what a previous workgroup left in LDS
  {
    unsigned* w_ = reinterpret_cast<unsigned*>(&sm);
    for (int i = threadIdx.x; i < (int)(sizeof(sm) / 4); i += 64) w_[i] = 0xFFF4DEADu;
    WSYNC();
  }
#endif
  int warn = 0;
  if (S.prof && env == 0 && lane < RPK_NPROF) sm.prof[lane] = 0;
  long long prof_t = (long long)__builtin_readcyclecounter();
  const long long kernel_t0 = prof_t;
  const int nl = M.nlink, nk = M.nkey, nv = M.nv, nu = M.nu;
  const T h = M.timestep;
  const bool isl = lane < nl;
  const int L = isl ? lane : 0;
  // The per-lane integers of the solve live PACKED in three registers and are unpacked, from an opaque copy,
  // where they are used (v_bfe is one instruction; two dozen integers held through the Newton loop, and the
  // lane masks the compiler derives from them and hoists, are what pushed this stage into scratch):
  //   tpk: depth + 1 (0: not a link) | trunk length << 4 | trunk base lane << 7 | tree << 13 | links on my chain << 15
  //        | my chain << 18 | chains of my tree (bit mask) << 21
  //   kpk: solver slot + 1 of my key (5 bits) | of my key + 64 << 5 | touched keys hanging under me << 10
  //   spk: slot lanes: anchor link | its trunk length << 8 | its trunk base << 12 | its depth + 1 << 18; limit row signs + 1 << 22
  int tpk, kpk, spk;
  int ldof, lact;
  const bool isk[2] = {lane < nk, lane + 64 < nk};
  const int kid[2] = {lane, lane + 64};
  const size_t lf = (size_t)env * RPK_NLF * 64 + lane, li = (size_t)env * RPK_NLI * 64 + lane;
#define LF(i) B.lanef[lf + (size_t)(i) * 64]
#define LI(i) B.lanei[li + (size_t)(i) * 64]
  // ---- the prologue's reads: topology record, key tables, the hand-over's per-lane fields, the first round of entries
  const int4* rec = (const int4*)(M.lane_topo() + 16 * L);
  const int4 tr0 = rec[0], tr1 = rec[1], tr2 = rec[2], tr3 = rec[3];
  int kdof[2], kact[2], kslot_raw[2];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int K = isk[s] ? kid[s] : 0;
    kdof[s] = M.key_dof()[K]; kact[s] = M.key_act()[K];
    kslot_raw[s] = (int)((const signed char*)(B.keyslot + (size_t)env * (RPK_NKEYS / 4)))[K];
  }
  int pli[12];
#pragma unroll
  for (int i = 0; i < 12; i++) pli[i] = (i == 1 || i == 2) ? 0 : LI(i);
  T plf[17];   // LF(8 .. 10), LF(14 .. 15), LF(16 .. 24): limit / contact D, friction, contact frame
#pragma unroll
  for (int i = 0; i < 3; i++) plf[i] = LF(8 + i);
#pragma unroll
  for (int i = 0; i < 11; i++) plf[3 + i] = LF(14 + i);
  // this lane's first contact Jacobian entry (the list has room for RpCaps::NE entries per env: reading past its end is
  // reading this env's own, unused, slots)
  T pej[3]; int pem[2];
  {
    const size_t e = (size_t)env * RpCaps<T>::NE + lane;
    pej[0] = B.entJ[e * 3]; pej[1] = B.entJ[e * 3 + 1]; pej[2] = B.entJ[e * 3 + 2];
    pem[0] = B.entM[e * 2]; pem[1] = B.entM[e * 2 + 1];
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    ldof = isl ? tr1.y : 0; lact = isl ? tr2.z : -1;
    tpk = isl ? ((tr0.y + 1) | (tr1.w << 4) | (tr1.z << 7) | (tr1.x << 13) | ((tr2.w - tr1.w) << 15) | ((tr3.x & 7) << 18) | ((tr3.y & 31) << 21)) : 0;
  }
  struct Topo { int depth, TL, tbase, ltree, clen, mychain, chainmask; };
  auto topo = [&]() -> Topo {
    int t = tpk;
    asm volatile("" : "+v"(t));
    Topo o;
    o.depth = (t & 15) - 1; o.TL = (t >> 4) & 7; o.tbase = (t >> 7) & 63; o.ltree = (t >> 13) & 3; o.clen = (t >> 15) & 7;
    o.mychain = (t >> 18) & 7; o.chainmask = (t >> 21) & 31;
    return o;
  };
  // lane of my ancestor at depth e (lanes are in preorder: trunk chain, then the leaf chains)
  auto anc_at = [&](const Topo& tp, int e) -> int { return e < tp.TL ? tp.tbase + e : lane - (tp.depth - e); };
  auto anc_of = [](int lk, int dl, int tl, int tb, int e) -> int { return e < tl ? tb + e : lk - (dl - e); };
  kpk = 0;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    kdof[s] = isk[s] ? kdof[s] : 0;
    kact[s] = isk[s] ? kact[s] : -1;
    // solver slot of my key (-1: not touched)
    const int ks = isk[s] ? kslot_raw[s] : -1;
    kpk |= (ks + 1) << (5 * s);
  }
  auto myks = [&](int s) -> int {
    int t = kpk;
    asm volatile("" : "+v"(t));
    return ((t >> (5 * s)) & 31) - 1;
  };
  const size_t eo = (size_t)env * nv;
  // ---- what the position / velocity stage left behind
  const int ncon = hq0.x, nkt = hq0.y;
  const unsigned long long dirty_mask = ((unsigned long long)(unsigned)hq0.w << 32) | (unsigned)hq0.z;
  const int nent = hq1.x, maxm = hq1.y;
  // my mass-matrix row over my ancestors (diag at [depth]); re-read from the (L2-resident) hand-over where it
  // is used instead of holding 20 registers through the Newton loop
  auto Mrow = [&]() -> const T* { return fresh(B.RM) + ((size_t)env * RPK_NLX(MD) + L) * (MD + 1); };
  // ... in the local column layout of the tree solve (see there), plus `add` on my diagonal
  // (in two steps so that the Newton loop can request the row one phase ahead of its use)
  auto fetch_Mlocal = [&](T* Rl) {
    const Topo tp = topo();
    const int pos = isl ? tp.depth - tp.TL : -2;
    const int shift = (isl && pos >= 0) ? tp.TL - TC : 0;
    const T* row = Mrow();
#pragma unroll
    for (int k = 0; k <= MD; k++) Rl[k] = isl ? row[k < TC ? k : k + shift] : (T)0;
  };
  auto finish_Mlocal = [&](T* Rl, const T add) {
    const Topo tp = topo();
    const int pos = isl ? tp.depth - tp.TL : -2;
    const int kd = pos >= 0 ? TC + pos : tp.depth;
#pragma unroll
    for (int k = 0; k < MD; k++) if (isl && k == kd) Rl[k] += add;
  };
  // (one piece for the Newton loop's call: in two steps it cost the fused-substeps kernel 900 spilled registers)
  auto load_Mlocal = [&](T* Rl, const T add) {
    const Topo tp = topo();
    const int pos = isl ? tp.depth - tp.TL : -2;
    const int shift = (isl && pos >= 0) ? tp.TL - TC : 0, kd = pos >= 0 ? TC + pos : tp.depth;
    const T* row = Mrow();
#pragma unroll
    for (int k = 0; k <= MD; k++) Rl[k] = isl ? row[k < TC ? k : k + shift] : (T)0;
#pragma unroll
    for (int k = 0; k < MD; k++) if (isl && k == kd) Rl[k] += add;
  };
  auto load_Mr = [&](T* Mr) {
    const T* row = fresh(B.RM) + ((size_t)env * RPK_NLX(MD) + L) * (MD + 1);
#pragma unroll
    for (int e = 0; e <= MD; e++) Mr[e] = isl ? row[e] : (T)0;
  };
  spk = (pli[0] & 63) << 22;
  auto lim_sign = [&](int s) -> int {
    int t = spk;
    asm volatile("" : "+v"(t));
    return ((t >> (22 + 2 * s)) & 3) - 1;
  };
  T lim_D[3] = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < 3; k++) if (lim_sign(k) != 0) lim_D[k] = plf[k];
  T con_D = 0, con_mu = 0;
  int cinfo = 0;   // my contact: first entry | entries << 8 | cross-chain << 16
  if (lane < ncon) {
    con_D = plf[3]; con_mu = plf[4];
    {
      const int bc = pli[10];   // (hand-over layout: first entry | entries << 12; a light env's fit in 8 bits each)
      cinfo = (bc & 255) | (((bc >> 12) & 63) << 8) | ((pli[4] & 1) << 16);
      unsigned long long sup = 0;
      if ((bc >> 12) & 63) {   // (a contact dropped for capacity keeps no entries)
        sup = (((unsigned long long)(unsigned)pli[6] << 32) | (unsigned)pli[5]) | (((unsigned long long)(unsigned)pli[8] << 32) | (unsigned)pli[7]);
        const int slot = pli[3];
        if (slot >= 0) sup |= 1ull << (nl + slot);
      }
      sm.csup[lane] = sup; sm.cinf[lane] = cinfo;
    }
    // contact frame -> LDS (rows of sm.R are free until the first assembly), only to rotate the entries
#pragma unroll
    for (int k = 0; k < 9; k++) sm.R[lane][k] = plf[5 + k];
  }
  {
    const int sl_ = pli[11], sd_ = pli[9];
    spk |= (sl_ & 255) | (((sl_ >> 8) & 15) << 8) | (((sl_ >> 16) & 63) << 12) | (((sd_ + 1) & 15) << 18);
  }
  struct SlotI { int salink, sTL, sTB, sdepth; };
  auto slot_info = [&]() -> SlotI {
    int t = spk;
    asm volatile("" : "+v"(t));
    SlotI o;
    o.salink = t & 255; o.sTL = (t >> 8) & 15; o.sTB = (t >> 12) & 63; o.sdepth = ((t >> 18) & 15) - 1;
    return o;
  };
  // which touched keys hang under me (bit s: the anchor chain of slot s passes through this link)
  {
    const int* sl = B.slots + (size_t)env * 64;
    for (int s = 0; s < nkt; s++) {
      const unsigned long long am = ((unsigned long long)(unsigned)sl[48 + s] << 32) | (unsigned)sl[32 + s];
      if (isl && ((am >> lane) & 1)) kpk |= 1 << (10 + s);
    }
  }
  WSYNC();
  // the contacts that touch my row (bit c; `crossm`, uniform: the cross-chain ones), for the Hessian gather and
  // J^T f: a lane then walks ITS contacts, and the wave takes as many trips as the busiest lane has contacts (a
  // trunk link: the contacts of its hand) instead of one trip per contact of the env
  unsigned call = 0, crossm = 0;
  for (int c0 = 0; c0 < ncon; c0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int c = c0 + u < ncon ? c0 + u : c0;
      const int inf = sm.cinf[c];
      const unsigned long long sup = sm.csup[c];
      if (c0 + u < ncon && ((sup >> lane) & 1)) call |= 1u << c;
      if (c0 + u < ncon && ((inf >> 16) & 1)) crossm |= 1u << c;
    }
  }
  crossm = (unsigned)uni((int)crossm);
  // contact Jacobian entries, rotated into the contact frame of their contact
  auto put_entry = [&](const int i, const T j0, const T j1, const T j2, const int h0, const int h1) {
    // (the hand-over's records, rp_model.hpp, repacked into this stage's narrower fields: lane | contact << 6 |
    // column << 11 | cross << 15 and first entry | entries << 8 | rank << 16 -- a light env has < 32 contacts and
    // < 256 entries)
    // (round 6: ... | the row of the entry's dof in the packed dense block << 16 -- its rank among the dirty rows, which the
    // cross-contact pass used to recount, one 64-bit population count per visited entry and Newton iteration)
    const int m0 = RPK_EM_LANE(h0) | (RPK_EM_CON(h0) << 6) | (RPK_EM_COL(h0) << 11) | (RPK_EM_CROSS(h0) << 15) |
                   (__popcll(dirty_mask & lanemask_lt(RPK_EM_LANE(h0))) << 16);
    const int m1 = RPK_EM_BASE(h1) | (RPK_EM_CNT(h1) << 8) | (RPK_EM_RANK(h1) << 16);
    const T* fr = sm.R[(m0 >> 6) & 31];
    sm.entJ[i][0] = fr[0] * j0 + fr[1] * j1 + fr[2] * j2;
    sm.entJ[i][1] = fr[3] * j0 + fr[4] * j1 + fr[5] * j2;
    sm.entJ[i][2] = fr[6] * j0 + fr[7] * j1 + fr[8] * j2;
    sm.entM[i][0] = m0; sm.entM[i][1] = m1;
  };
  if (lane < nent) put_entry(lane, pej[0], pej[1], pej[2], pem[0], pem[1]);   // (fetched with the prologue's batch)
  for (int i = lane + 64; i < nent; i += 64) {
    const size_t e = (size_t)env * RpCaps<T>::NE + i;
    put_entry(i, B.entJ[e * 3], B.entJ[e * 3 + 1], B.entJ[e * 3 + 2], B.entM[e * 2], B.entM[e * 2 + 1]);
  }
  WSYNC();
  PROF(0);
  // ---- per-lane constants of the dynamics, actuator tables and state: every read of this phase requested in ONE batch
  // (clamped indices, values selected afterwards) -- read where they were used, under their predicates, they were some
  // fifteen dependent trips to L2
  const int A = lane < nu ? lane : 0;
  int KA[2], Kc[2];
#pragma unroll
  for (int s = 0; s < 2; s++) { Kc[s] = isk[s] ? kid[s] : 0; KA[s] = (isk[s] && kact[s] >= 0) ? kact[s] : 0; }
  const T ld_floss = M.link_floss()[L], ld_flR = M.link_fl_R()[L], ld_stiff = M.link_stiffness()[L], ld_sref = M.link_springref()[L],
          ld_actcoef = M.link_act_coef()[L], ld_damp = M.link_damping()[L];
  T kd_M[2], kd_stiff[2], kd_sref[2], kd_mass[2], kd_hx[2], kd_damp[2];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    kd_M[s] = M.key_M()[Kc[s]]; kd_stiff[s] = M.key_stiffness()[Kc[s]]; kd_sref[s] = M.key_springref()[Kc[s]];
    kd_mass[s] = M.key_mass()[Kc[s]]; kd_hx[s] = M.key_half()[3 * Kc[s]]; kd_damp[s] = M.key_damping()[Kc[s]];
  }
  const auto a_kind = M.act_kind()[A];
  const auto a_climited = M.act_ctrllimited()[A];
  const auto a_flimited = M.act_forcelimited()[A];
  const T a_clo = M.act_ctrlrange()[2 * A], a_chi = M.act_ctrlrange()[2 * A + 1], a_gain = M.act_gain()[A],
          a_b0 = M.act_bias()[3 * A], a_b1 = M.act_bias()[3 * A + 1], a_b2 = M.act_bias()[3 * A + 2],
          a_flo = M.act_forcerange()[2 * A], a_fhi = M.act_forcerange()[2 * A + 1];
  T k_clo[2], k_chi[2], k_gain[2], k_flo[2], k_fhi[2], k_coef[2], k_ctrl[2];
  bool k_climited[2], k_flimited[2];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    k_climited[s] = M.act_ctrllimited()[KA[s]] != 0; k_flimited[s] = M.act_forcelimited()[KA[s]] != 0;
    k_clo[s] = M.act_ctrlrange()[2 * KA[s]]; k_chi[s] = M.act_ctrlrange()[2 * KA[s] + 1]; k_gain[s] = M.act_gain()[KA[s]];
    k_flo[s] = M.act_forcerange()[2 * KA[s]]; k_fhi[s] = M.act_forcerange()[2 * KA[s] + 1]; k_coef[s] = M.act_coef()[2 * KA[s]];
    k_ctrl[s] = S.ctrl[(size_t)env * nu + KA[s]];
  }
  const T s_ctrl = S.ctrl[(size_t)env * nu + A];
  const T s_q0 = S.qpos[eo + ldof], s_qd0 = S.qvel[eo + ldof];
  T s_qk[2], s_qdk[2], s_qappk[2] = {0, 0};
#pragma unroll
  for (int s = 0; s < 2; s++) { s_qk[s] = S.qpos[eo + kdof[s]]; s_qdk[s] = S.qvel[eo + kdof[s]]; }
  T s_qapp0 = 0;
  if (S.qfrc_applied) {
    s_qapp0 = S.qfrc_applied[eo + ldof];
#pragma unroll
    for (int s = 0; s < 2; s++) s_qappk[s] = S.qfrc_applied[eo + kdof[s]];
  }
  T hf[7];
#pragma unroll
  for (int i = 0; i < 7; i++) hf[i] = LF(i);
  T Rr_s[MD + 1];   // (my mass-matrix row for the qacc_smooth solve that follows this phase)
  fetch_Mlocal(Rr_s);
  __builtin_amdgcn_sched_barrier(0);
  const T lfloss = isl ? ld_floss : (T)0;
  const T lflR = isl ? ld_flR : (T)1;
  const T lflD = (T)1 / lflR;
  T kM[2];
#pragma unroll
  for (int s = 0; s < 2; s++) kM[s] = isk[s] ? kd_M[s] : (T)1;
  // ---- actuation, passive forces, bias [MJ: mj_fwdActuation, mj_passive] -> qfrc_smooth
  T qfs[3], qs[3];
  {
    const bool isa = lane < nu && a_kind == 0;
    T ctrl = (lane < nu) ? s_ctrl : (T)0;
    {
      const T cl_ = fmin(a_chi, fmax(a_clo, ctrl));
      ctrl = (lane < nu && a_climited) ? cl_ : ctrl;
    }
    if (isa) {
      const T alen = hf[1], avel = hf[2];
      T aforce = a_gain * ctrl + a_b0 + a_b1 * alen +
                 a_b2 * avel;
      if (a_flimited)
        aforce = fmin(a_fhi, fmax(a_flo, aforce));
      sm.vec[lane] = aforce;
      S.act_force[(size_t)env * nu + lane] = aforce;
    }
    WSYNC();
    const T qbias = hf[0];
    const T q0 = isl ? s_q0 : (T)0, qd0 = isl ? s_qd0 : (T)0;
    const T qapp0 = isl ? s_qapp0 : (T)0;
    const T lstiff = isl ? ld_stiff : (T)0, lsref = isl ? ld_sref : (T)0;
    const T lactcoef = isl ? ld_actcoef : (T)0;
    const T qact = (isl && lact >= 0) ? lactcoef * sm.vec[lact >= 0 ? lact : 0] : (T)0;
    const T qpas = -lstiff * (q0 - lsref) - (isl ? ld_damp : (T)0) * qd0;
    qfs[0] = isl ? (qpas - qbias + qapp0 + qact) : (T)0;
    const T ksin[2] = {hf[3], isk[1] ? hf[4] : (T)0}, kcos[2] = {hf[5], isk[1] ? hf[6] : (T)1};
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const T kstiff = isk[s] ? kd_stiff[s] : (T)0, ksref = isk[s] ? kd_sref[s] : (T)0;
      const T kmass = isk[s] ? kd_mass[s] : (T)0, khx = isk[s] ? kd_hx[s] : (T)0;
      const T qk = isk[s] ? s_qk[s] : (T)0, qdk = isk[s] ? s_qdk[s] : (T)0;
      const T qappk = isk[s] ? s_qappk[s] : (T)0;
      const T grav = -kmass * M.gz * khx * kcos[s] - kmass * M.gx * khx * ksin[s];
      T f = -kstiff * (qk - ksref) - (isk[s] ? kd_damp[s] : (T)0) * qdk + grav + qappk;
      {
        const bool on = isk[s] && kact[s] >= 0;
        T c = k_ctrl[s];
        {
          const T cl_ = fmin(k_chi[s], fmax(k_clo[s], c));
          c = k_climited[s] ? cl_ : c;
        }
        T af = k_gain[s] * c;
        {
          const T al_ = fmin(k_fhi[s], fmax(k_flo[s], af));
          af = k_flimited[s] ? al_ : af;
        }
        const T f2 = f + k_coef[s] * af;
        f = on ? f2 : f;
        if (on) S.act_force[(size_t)env * nu + kact[s]] = af;
      }
      qfs[1 + s] = f;
      qs[1 + s] = f / kM[s];
    }
    WSYNC();
  }
  // kdof / kact are recomputed where the new state is stored

  PROF(1);

  // ---- hybrid tree-sparse / dense factor + solve [MJ: mj_factorM / mj_solveLD], rows in their lanes.
  // Row r of the symmetric system is held by lane r as Rr[e] = A[r][anc_e(r)] (diag at e = depth; slot
  // lanes: key leaf under its anchor link, diag at e = sdepth + 1).  Rows whose bit is clear in `dm`
  // ("clean") couple only to their ancestors and are eliminated leaf-to-root without fill-in: first the
  // key leaves, then chain position 4, 3, .. 0 for all chains of both trees at once (the row of an
  // eliminated link crosses to the links above it on its chain through LDS; what it leaves on the trunk
  // is added into a per-tree table), then trunk position 3 .. 0.  The rows in `dm` (supports of cross-chain
  // contacts: an ancestor-closed set) receive the Schur complement and are solved by the dense block
  // (sm.H already holds the cross-contact terms).  Back-substitution runs root to leaves, one level at a
  // time.  Returns x for this lane's row.
  auto tree_solve = [&](T* Rr, const T sdiag, T rhs, int nslots, unsigned long long dm) -> T {
    // (the lane predicates below -- pos == j, depth == j, ...: some fifty 64-bit masks -- are formed here,
    // from an opaque copy of the packed topology, so that the compiler does not hoist them out of the
    // Newton loop and then spill them: v_cmp is cheaper than a spilled SGPR pair)
    const Topo tp = topo();
    const int depth = tp.depth, TL = tp.TL, tbase = tp.tbase, ltree = tp.ltree, clen = tp.clen;
    const SlotI si = slot_info();
    const int sdepth = si.sdepth, salink = si.salink, sTL = si.sTL, sTB = si.sTB;
    const int foldmask = kpk >> 10;
    const int pos = isl ? depth - TL : -2;          // chain position (trunk links: negative)
    // Link lanes hold their row in a LOCAL column layout: columns 0 .. TC-1 = the trunk (the first TL of
    // them exist), TC + p = the link at position p of my chain -- so the diagonal of "position j" is a
    // compile-time register index for every trunk length.  Slot lanes keep the depth layout.
    // (slot lanes: Rr[e], e <= sdepth = my row over the anchor link's path, `sdiag` my diagonal)
    const int shift = (isl && pos >= 0) ? TL - TC : 0;
    const int kd = pos >= 0 ? TC + pos : depth;     // my diagonal (link lanes)
#ifdef RPK_X_NOTS   // compile-only experiment: register floor without the tree solve
    return rhs * Rr[0] + (T)(nslots + (int)dm);
#endif
    const bool isslot = !isl && lane < nl + nslots;
    const bool dirty = (dm >> lane) & 1;
    T mydinv = 0;   // reciprocal pivot of my (clean) row
    // ---- key leaves (they hang under chain / trunk links): the slot lanes publish their scaled rows,
    // the link lanes on the anchor's path fold them in
    T Dslot = 1;
    if (nslots > 0) {
      if (isslot) {
        T Dk = sdiag;
        if (!dirty) {
          if (!(Dk >= RPK_MINVAL)) { Dk = RPK_MINVAL; warn |= 4; }
          Dslot = Dk;
        }
        const T inv = dirty ? (T)1 : rcp_nr(Dk);
#pragma unroll
        for (int e = 0; e < MD; e++) sm.R[lane][e] = Rr[e] * inv;
        sm.jt[lane] = Dk;
        sm.xs[lane] = rhs;
      }
      WSYNC();
      if (isl) {
        for (int sidx = 0; sidx < nslots; sidx++) {
          const T* Lk = sm.R[nl + sidx];
          T lrow[MD];
#pragma unroll
          for (int k = 0; k < MD; k++) lrow[k] = Lk[k < TC ? k : k + shift];
          const T lk = Lk[depth], dk = sm.jt[nl + sidx], xk = sm.xs[nl + sidx];
          const bool fold = ((foldmask >> sidx) & 1) && !((dm >> (nl + sidx)) & 1);
          const T t = lk * dk;
#pragma unroll
          for (int k = 0; k < MD; k++) Rr[k] -= fold ? t * lrow[k] : (T)0;   // (selected: unwritten columns may hold NaN)
          rhs -= fold ? lk * xk : (T)0;
        }
      }
      WSYNC();
    }
    PROF(20);
    // ---- chains: position j = 4 .. 0.  (a) the links at position j publish their final row, right-hand
    // side (column MD of the record) and reciprocal pivot; (b) the links above them on the same chain
    // (position jp < j) take the update  R[k] -= (H[v][me] / d_v) H[v][k]  from the row of v = lane + (j - jp).
    // Whole records cross (one lane mask per step, wide LDS accesses): the columns a link does not own
    // carry garbage in both directions and are never used.
    // What the eliminated chain links leave on their trunk -- sum over v of H[v][t] H[v][t'] / d_v (lower triangle,
    // then the right-hand sides) -- is accumulated by the first link of every chain, which receives every row of
    // its chain anyway.  (No LDS adds: 22 lanes of a tree adding into one cell cost ~600 cycles per instruction,
    // scratch/ub/ldsadd_ub.hip; a separate gather of the published records cost a thousand VALU instructions.)
    T acc[14];
#pragma unroll
    for (int q = 0; q < 14; q++) acc[q] = 0;
    auto trunk_part = [&](const T* rowv, const T dinv, const T bv) {
#pragma unroll
      for (int t = 0; t < TC; t++) {
        const T lt = rowv[t] * dinv;
#pragma unroll
        for (int t2 = 0; t2 <= t; t2++) acc[t * (t + 1) / 2 + t2] -= lt * rowv[t2];
        acc[10 + t] -= lt * bv;
      }
    };
#pragma unroll
    for (int j = 4; j >= 0; j--) {
      if (isl && pos == j) {
        T dv = Rr[TC + j];
        if (!dirty) {
          if (!(dv >= RPK_MINVAL)) { dv = RPK_MINVAL; warn |= 4; }
          mydinv = rcp_nr(dv);
        }
#pragma unroll
        for (int k = 0; k < MD; k++) sm.R[lane][k] = Rr[k];
        sm.R[lane][MD] = rhs;
        sm.jt[lane] = mydinv;
      }
      if (j == 0) break;
      WSYNC();
      {
        const int v = lane + (j - pos);
        const bool recv = isl && pos >= 0 && pos < j && j < clen && !((dm >> (v & 63)) & 1);
        if (recv) {
          const T* Rv = sm.R[v];
          T rowv[MD + 1];
#pragma unroll
          for (int k = 0; k <= MD; k++) rowv[k] = Rv[k];
          const T dinv = sm.jt[v];
          const T l = Rv[kd] * dinv;
#pragma unroll
          for (int k = 0; k < MD; k++) Rr[k] -= l * rowv[k];
          rhs -= l * rowv[MD];
          trunk_part(rowv, dinv, rowv[MD]);   // (used by the lanes at position 0)
        }
      }
    }
    PROF(32);
    if (isl && pos == 0) {
      if (!dirty) trunk_part(Rr, mydinv, rhs);
      T* st = sm.stage[(ltree & 1) * 5 + tp.mychain];
#pragma unroll
      for (int q = 0; q < 14; q++) st[q] = acc[q];
    }
    PROF(33);
    WSYNC();
    if (isl && pos < 0) {
      // (the five chains' records read together, then added in the old order: a predicated block per chain was five
      // dependent round trips per solve)
      const int tro = depth * (depth + 1) / 2;
      T dv[5][TC], dr[5];
#pragma unroll
      for (int c = 0; c < 5; c++) {
        const T* st = sm.stage[(ltree & 1) * 5 + c];
#pragma unroll
        for (int e = 0; e < TC; e++) dv[c][e] = st[tro + e < 10 ? tro + e : 9];
        dr[c] = st[10 + depth];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 5; c++) {
        const bool on = (tp.chainmask >> c) & 1;
#pragma unroll
        for (int e = 0; e < TC; e++) { const T t_ = Rr[e] + dv[c][e]; Rr[e] = (on && e <= depth) ? t_ : Rr[e]; }
        { const T t_ = rhs + dr[c]; rhs = on ? t_ : rhs; }
      }
    }
    PROF(34);
    // ---- trunk: position j = 3 .. 0, the same way along the trunk chain
#pragma unroll
    for (int j = TC - 1; j >= 0; j--) {
      if (isl && pos < 0 && depth == j) {
        T dv = Rr[j];
        if (!dirty) {
          if (!(dv >= RPK_MINVAL)) { dv = RPK_MINVAL; warn |= 4; }
          mydinv = rcp_nr(dv);
        }
#pragma unroll
        for (int k = 0; k < TC; k++) sm.R[lane][k] = Rr[k];
        sm.R[lane][MD] = rhs;
        sm.jt[lane] = mydinv;
      }
      if (j == 0) break;
      WSYNC();
      {
        const int v = tbase + j;
        const bool recv = isl && pos < 0 && depth < j && j < TL && !((dm >> (v & 63)) & 1);
        if (recv) {
          const T* Rv = sm.R[v];
          T rowv[TC];
#pragma unroll
          for (int k = 0; k < TC; k++) rowv[k] = Rv[k];
          const T l = Rv[depth] * sm.jt[v];
#pragma unroll
          for (int k = 0; k < TC; k++) Rr[k] -= l * rowv[k];
          rhs -= l * Rv[MD];
        }
      }
    }
    WSYNC();
    PROF(21);
    // local column k of a link lane: does it exist, and which lane is that ancestor
    auto col_valid = [&](int k) -> bool { return k < TC ? (k < TL && k <= depth) : (pos >= 0 && k - TC <= pos); };
    auto col_lane = [&](int k) -> int { return k < TC ? tbase + k : lane - (pos - (k - TC)); };
    T xcur = 0;   // my row's x once its level has passed (dirty rows: from the dense block)
    // ---- dense block on the dirty rows (Schur complement + cross-contact terms)
    if (dm) {
      const int nD = __popcll(dm);
      auto cidx = [&](int l) -> int { return __popcll(dm & lanemask_lt(l)); };
      const int ci = cidx(lane);
      // every (row, ancestor) element has exactly one owner lane: plain read-modify-write
      if (dirty) {
        if (isl) {
#pragma unroll
          for (int k = 0; k < MD; k++) if (col_valid(k)) lds_add(&sm.H[tri(ci, cidx(col_lane(k)))], Rr[k]);   // (one owner per element: conflict-free, no round trip)
        } else {
          lds_add(&sm.H[tri(ci, ci)], sdiag);
#pragma unroll
          for (int e = 0; e < MD; e++) {
            if (e <= sdepth) {
              const int a_ = anc_of(salink, sdepth, sTL, sTB, e);
              if ((dm >> a_) & 1) lds_add(&sm.H[tri(ci, cidx(a_))], Rr[e]);
            }
          }
        }
        // the rhs of compact row r is row nD of the packed block
        sm.H[tri(nD, 0) + ci] = rhs;
      }
      WSYNC();
      PROF(22);
      // (four columns per step also in fp64: this build has the registers for it; +0.8 ... 1.7 %)
      const T xr = dense_factor_solve<T, true>(sm.H, nD, lane, &warn);
      WSYNC();
      if (lane < nD) sm.vec[lane] = xr;
      WSYNC();
      const T xd = sm.vec[dirty ? ci : 0];
      if (dirty) xcur = xd;
      WSYNC();
    }
    PROF(25);
    // ---- back-substitution, root to leaves: x_v = (b_v - sum_{a above v} H[v][a] x[a]) / d_v, one level per local
    // column (trunk 0 .. TC-1, then chain positions 0 .. 4).  Round 6: the levels no longer cross LDS (nine dependent
    // write -> read round trips per solve).  A trunk link's x goes to everything below it with v_readlane (one source
    // lane per tree, selected by the lane's tree); the x of the link at chain position p goes down its chain on a
    // wave_shr:1 DPP cascade -- after d shifts the lane at position p + d holds it (chain lanes are consecutive, and a
    // lane takes the value only at the shift that equals its distance, so what crosses into the next chain is never
    // used).  Same terms in the same order as the LDS version: bit-identical.
    T s_ = rhs;
    {
      const int tb0 = bcast(tbase, 0), tb1 = bcast(tbase, nl - 1);   // (trunk base lane of the first / last tree)
#pragma unroll
      for (int k = 0; k < TC; k++) {
        if (isl && kd == k && !dirty) xcur = s_ * mydinv;
        const T v0 = bcast(xcur, tb0 + k), v1 = bcast(xcur, tb1 + k);
        const T xa = (ltree & 1) ? v1 : v0;
        const bool below = isl && col_valid(k) && kd != k;
        if (below) s_ -= Rr[k] * xa;
      }
#pragma unroll
      for (int q = 0; q < MD - TC; q++) {
        const int k = TC + q;
        if (isl && kd == k && !dirty) xcur = s_ * mydinv;
        if (q == MD - TC - 1) break;
        T t = xcur, xa = 0;
#pragma unroll
        for (int d = 1; d < MD - TC - q; d++) {
          t = dpp_move<0x138>(t);   // wave_shr:1
          xa = (pos == q + d) ? t : xa;
        }
        const bool below = isl && col_valid(k) && kd != k;
        if (below) s_ -= Rr[k] * xa;
      }
    }
    T x = (isl || (isslot && dirty)) ? xcur : (T)0;
    if (nslots > 0) {
      // the key leaves read their anchor path's x from LDS
      if (isl) sm.xs[lane] = xcur;
      WSYNC();
      if (isslot && !dirty) {
        x = rhs / Dslot;
        // (reads together, then the terms in the old order: a predicated pair of reads per ancestor was nine dependent round trips)
        T rr_[MD], xx_[MD];
#pragma unroll
        for (int e = 0; e < MD; e++) {
          const int ee = e <= sdepth ? e : 0;
          rr_[e] = sm.R[lane][ee]; xx_[e] = sm.xs[anc_of(salink, sdepth, sTL, sTB, ee)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < MD; e++) { const T t_ = x - rr_[e] * xx_[e]; x = e <= sdepth ? t_ : x; }
      }
    }
    WSYNC();
    return x;
  };

  // ---- qacc_smooth = M^-1 qfrc_smooth (M is always tree-sparse)
  {
    finish_Mlocal(Rr_s, (T)0);
    qs[0] = tree_solve(Rr_s, (T)0, qfs[0], 0, 0ull);
  }
  PROF(2);


  // ---- constraint solve [MJ: mj_solNewton]
  // State of the iteration, per lane (dof slots: my hand dof, my key, my key + 64):
  //   dq = qacc - qacc_smooth,  r0 = (M qacc - qfrc_smooth) of my hand dof (keys: r = kM dq),  qfc = J^T f,
  //   jar = J qacc - aref for the rows I own.  qacc_smooth is parked in S.warm and qfrc_smooth in the
  //   hand-over slots it came from until the Euler step needs them again.
  const int nsys = nl + nkt;
  const bool hascon = lane < ncon && con_D > 0;
  const T scale = (T)1 / (M.meaninertia * (T)(nv > 1 ? nv : 1));
  // rows owned by this lane: friction loss of my hand dof, one limit row per dof slot, four pyramidal
  // rows of my contact
  struct RowsL { T fr, lim[3], con[4]; };
  T dq[3] = {0, 0, 0}, r0 = 0, qfc[3] = {0, 0, 0};
  RowsL jar;
  int act = 0;   // active-set bits: 0 friction row in its quadratic zone, 1..3 limit rows, 4..7 contact rows

  // y = J x for the rows owned by this lane (x in per-lane slot registers).  Every contact lane walks ITS
  // entries (no LDS adds: ten lanes adding into one contact's cell cost ~400 cycles per instruction).
  auto mulJ = [&](const T* x, RowsL& out) {
    // (round 6: a touched key's value goes to its SLOT LANE's cell of sm.vec -- entry lanes nl + slot -- so that an entry
    // reads sm.vec[its lane] whatever kind of dof it is.  Every lane writes its own cell first (unpredicated: `if (isl)`
    // around this store cost 2 % of the step, 631 against 644 k env-steps/s), then, behind a wave-level fence, the key lanes
    // overwrite their slots' cells)
    sm.vec[lane] = x[0];
    WSYNC();
#pragma unroll
    for (int s = 0; s < 2; s++) { const int ks = myks(s); if (ks >= 0) sm.vec[nl + ks] = x[1 + s]; }
    WSYNC();
    out.fr = x[0];
#pragma unroll
    for (int s = 0; s < 3; s++) out.lim[s] = (T)lim_sign(s) * x[s];
    T vc[3] = {0, 0, 0};
    {
      int ci_ = cinfo;
      asm volatile("" : "+v"(ci_));
      const int base = ci_ & 255, cnt = (ci_ >> 8) & 255;
      for (int k0 = 0; k0 < maxm; k0 += 4) {
        // (two LDS round trips per four entries: the entries' records and Jacobians together, then the values of their dofs.
        // The scheduling barriers keep the reads of a phase in flight together -- left alone, the compiler waited for
        // every entry's record and then for its value: eight dependent round trips per trip of this loop)
        int ln[4];
        T j0[4], j1[4], j2[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int e = min(base + k0 + u, nent - 1);   // (clamped: an entry past the list's end may hold NaN, and NaN x 0 is NaN)
          ln[u] = sm.entM[e][0];
          j0[u] = sm.entJ[e][0]; j1[u] = sm.entJ[e][1]; j2[u] = sm.entJ[e][2];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; u++) xv[u] = sm.vec[ln[u] & 63];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const T xm = k0 + u < cnt ? xv[u] : (T)0;   // (selected, not branched)
          vc[0] += j0[u] * xm; vc[1] += j1[u] * xm; vc[2] += j2[u] * xm;
        }
      }
    }
    const T vn = vc[0], v1 = con_mu * vc[1], v2 = con_mu * vc[2];
    out.con[0] = vn + v1; out.con[1] = vn - v1; out.con[2] = vn + v2; out.con[3] = vn - v2;
    WSYNC();
  };
  // active set from jar; returns this lane's share of the constraint cost [MJ: mj_constraintUpdate]
  auto update = [&](const RowsL& ja) -> T {
    T cost = 0;
    act = 0;
    if (isl && lfloss > 0) {
      const T x = ja.fr, rf = lflR * lfloss;
      if (x <= -rf) cost += -(T)0.5 * rf * lfloss - lfloss * x;
      else if (x >= rf) cost += -(T)0.5 * rf * lfloss + lfloss * x;
      else { act |= 1; cost += (T)0.5 * lflD * x * x; }
    }
#pragma unroll
    for (int s = 0; s < 3; s++) {
      if (lim_sign(s) != 0 && ja.lim[s] < 0) { act |= 2 << s; cost += (T)0.5 * lim_D[s] * ja.lim[s] * ja.lim[s]; }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (hascon && ja.con[r] < 0) { act |= 16 << r; cost += (T)0.5 * con_D * ja.con[r] * ja.con[r]; }
    }
    return cost;
  };
  // out = J^T f with the forces of the current active set: f = -D jar on active rows (friction rows in a
  // linear zone: -+ floss)
  auto mulJT = [&](const RowsL& ja, T* out) {
    T ffr = 0;
    if (isl && lfloss > 0) {
      const T rf = lflR * lfloss;
      ffr = ja.fr <= -rf ? lfloss : (ja.fr >= rf ? -lfloss : -lflD * ja.fr);
    }
    T fl[3], fc[4];
#pragma unroll
    for (int s = 0; s < 3; s++) fl[s] = ((act >> (1 + s)) & 1) ? -lim_D[s] * ja.lim[s] : (T)0;
#pragma unroll
    for (int r = 0; r < 4; r++) fc[r] = ((act >> (4 + r)) & 1) ? -con_D * ja.con[r] : (T)0;
    out[0] = ffr + (T)lim_sign(0) * fl[0];
    out[1] = (T)lim_sign(1) * fl[1];
    out[2] = (T)lim_sign(2) * fl[2];
    if (lane < ncon) {
      sm.cv[lane][0] = fc[0] + fc[1] + fc[2] + fc[3];
      sm.cv[lane][1] = con_mu * (fc[0] - fc[1]);
      sm.cv[lane][2] = con_mu * (fc[2] - fc[3]);
    }
    WSYNC();
    // every dof lane (link, solver slot) walks the contacts and takes its own entry of those that touch it
    T acc = 0;
    // (its OWN contacts, ascending, four per trip -- their headers, then their entries, travel together: one trip
    // for most envs instead of one per four contacts of the env)
    unsigned cm = call;
    asm volatile("" : "+v"(cm));
    while (__ballot(cm != 0u) != 0ull) {
      // (two LDS round trips per four contacts, kept together by scheduling barriers: the contacts' headers and force
      // vectors, then this lane's entry of each -- the compiler's own order waited for every header and every entry in turn,
      // twelve dependent round trips per trip)
      int c[4], e[4], inf[4];
      bool on[4];
      unsigned long long sup[4];
      T f[4][3], je[4][3];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        on[u] = cm != 0u;
        c[u] = on[u] ? __ffs((int)cm) - 1 : 0;
        cm &= cm - 1u;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        sup[u] = sm.csup[c[u]]; inf[u] = sm.cinf[c[u]];
        f[u][0] = sm.cv[c[u]][0]; f[u][1] = sm.cv[c[u]][1]; f[u][2] = sm.cv[c[u]][2];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int e_ = (inf[u] & 255) + __popcll(sup[u] & lanemask_lt(lane));
        e[u] = e_ < nent ? e_ : (nent > 0 ? nent - 1 : 0);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) { je[u][0] = sm.entJ[e[u]][0]; je[u][1] = sm.entJ[e[u]][1]; je[u][2] = sm.entJ[e[u]][2]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const T v = je[u][0] * f[u][0] + je[u][1] * f[u][1] + je[u][2] * f[u][2];
        acc += on[u] ? v : (T)0;
      }
    }
    if (isl) out[0] += acc;
    sm.jt[lane] = acc;
    WSYNC();
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int ks = myks(s);
      const T v = sm.jt[ks >= 0 ? nl + ks : 0];
      if (ks >= 0) out[1 + s] += v;
    }
    WSYNC();
  };
  // Gauss term of the cost: (1/2) (M qacc - qfrc_smooth) . (qacc - qacc_smooth)
  auto gauss = [&]() -> T { return (T)0.5 * (r0 * dq[0] + kM[0] * dq[1] * dq[1] + kM[1] * dq[2] * dq[2]); };
  // (M x) of my hand dof with the tree-sparse rows: the row part from this lane's own row, the column part
  // (descendants) scattered by the descendants with LDS adds.  Keys: M is diagonal (kM).
  auto mulM0 = [&](T x0, const T* Mr) -> T {   // (Mr: my mass-matrix row, load_Mr -- requested by the caller ahead of time)
    const Topo tp = topo();
    const int pos = isl ? tp.depth - tp.TL : -2;
    // every link publishes its products M[me][a] x_me (a = my ancestors; sm.R is free outside the factorisation)
    sm.vec[lane] = x0;
#pragma unroll
    for (int e = 0; e < MD; e++) sm.R[lane][e] = Mr[e] * x0;
    WSYNC();
    T y = 0;
    if (isl) {
      // (all reads of this part in flight together, then selected sums in the old order: a predicated read per ancestor was
      // nine dependent LDS round trips)
      T xa[MD];
#pragma unroll
      for (int e = 0; e < MD; e++) xa[e] = sm.vec[e <= tp.depth ? anc_at(tp, e) : lane];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < MD; e++) { const T t_ = y + Mr[e] * xa[e]; y = e <= tp.depth ? t_ : y; }
      // column part, in-chain: the links below me on my chain
      {
#pragma unroll
        for (int q = 1; q < 5; q++) {
          const bool on = pos >= 0 && pos + q < tp.clen;
          const T v = sm.R[on ? lane + q : lane][tp.depth];
          y += on ? v : (T)0;
        }
      }
      // the first link of every chain sums what its chain sends to the trunk
      if (pos == 0) {
        T acc[TC] = {0, 0, 0, 0};
#pragma unroll
        for (int p = 0; p < 5; p++) {
          const bool on = p < tp.clen;
          const T* row = sm.R[on ? lane + p : lane];
#pragma unroll
          for (int t = 0; t < TC; t++) { const T v = row[t]; acc[t] += on ? v : (T)0; }
        }
        T* st = sm.stage[(tp.ltree & 1) * 5 + tp.mychain];
#pragma unroll
        for (int t = 0; t < TC; t++) st[t] = acc[t];
      }
    }
    WSYNC();
    if (isl && pos < 0) {
      // trunk links: the chains' sums and the trunk links below me (reads together, then the sums in the old order)
      T vc_[5], vq[TC - 1];
#pragma unroll
      for (int c = 0; c < 5; c++) vc_[c] = sm.stage[(tp.ltree & 1) * 5 + c][tp.depth];
#pragma unroll
      for (int q = 1; q < TC; q++) {
        const bool on = tp.depth + q < tp.TL;
        vq[q - 1] = sm.R[on ? lane + q : lane][tp.depth];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 5; c++) { const T t_ = y + vc_[c]; y = ((tp.chainmask >> c) & 1) ? t_ : y; }
#pragma unroll
      for (int q = 1; q < TC; q++) {
        const bool on = tp.depth + q < tp.TL;
        y += on ? vq[q - 1] : (T)0;
      }
    }
    WSYNC();
    return y;
  };

  int anyrow = (isl && lfloss > 0) || lim_sign(0) || lim_sign(1) || lim_sign(2) || hascon;
  anyrow = __ballot(anyrow) != 0ull;
  int niter_last = 0;
  // park qfrc_smooth (the hand-over slots it was built from are dead) and qacc_smooth
  LF(0) = qfs[0]; LF(3) = qfs[1]; LF(4) = qfs[2];
  T qw[3];
  qw[0] = isl ? S.warm[eo + ldof] : (T)0;
#pragma unroll
  for (int s = 0; s < 2; s++) qw[1 + s] = isk[s] ? S.warm[eo + kdof[s]] : (T)0;
  if (isl) S.warm[eo + ldof] = qs[0];
  if (!anyrow) {
    jar.fr = 0;
#pragma unroll
    for (int s = 0; s < 3; s++) jar.lim[s] = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) jar.con[r] = 0;
  } else {
    // warmstart [MJ: warmstart()]: the cheaper of qacc_warmstart and qacc_smooth
    T cost;
    {
      // reference accelerations of my rows
      const T fr_aref = LF(7);
      T lim_aref[3] = {0, 0, 0}, con_aref[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 3; k++) if (lim_sign(k) != 0) lim_aref[k] = LF(11 + k);
      if (lane < ncon) {
#pragma unroll
        for (int k = 0; k < 4; k++) con_aref[k] = LF(25 + k);
      }
      auto sub_aref = [&](RowsL& r) {
        r.fr -= fr_aref;
#pragma unroll
        for (int s = 0; s < 3; s++) r.lim[s] -= lim_aref[s];
#pragma unroll
        for (int k = 0; k < 4; k++) r.con[k] -= con_aref[k];
      };
      RowsL jtmp;
      mulJ(qs, jtmp); sub_aref(jtmp);
      const T cost_smooth = uni(wave_sum(update(jtmp)));
      T Mrw[MD + 1];
      load_Mr(Mrw);
      const T Mw = mulM0(qw[0], Mrw);
      mulJ(qw, jar); sub_aref(jar);
      r0 = Mw - qfs[0];
#pragma unroll
      for (int s = 0; s < 3; s++) dq[s] = qw[s] - qs[s];
      // (keys: kM qw - qfs = kM (qw - qs))
      cost = uni(wave_sum(update(jar) + (T)0.5 * (r0 * dq[0] + (kM[0] * qw[1] - qfs[1]) * dq[1] + (kM[1] * qw[2] - qfs[2]) * dq[2])));
      if (cost > cost_smooth) {
#pragma unroll
        for (int s = 0; s < 3; s++) dq[s] = 0;
        r0 = 0;
        jar = jtmp;
        cost = uni(wave_sum(update(jar)));
      }
    }
    mulJT(jar, qfc);
    PROF(3);
    const int maxit = S.max_newton;
    for (int iter = 0; iter < maxit; iter++) {
      // gradient of the cost: M qacc - qfrc_smooth - J^T f
      const T grad[3] = {r0 - qfc[0], kM[0] * dq[1] - qfc[1], kM[1] * dq[2] - qfc[2]};
      // ---- H = M + J^T D J on the coupled system (hand dofs + touched keys)
      // rows of H in registers: links in the local column layout of the tree solve, solver slots over the
      // path of their anchor link (+ `sdiag`).  (My mass-matrix row is requested first: its trip to L2 runs under the
      // slot / weight hand-over below)
      T Rr[MD + 1];
      load_Mlocal(Rr, ((act & 1) ? lflD : (T)0) + (((act >> 1) & 1) ? lim_D[0] : (T)0));
      // key diagonals and gradients to the solver slots
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const int ks = myks(s);
        if (ks >= 0) {
          sm.slotv[0][ks] = kM[s] + (((act >> (2 + s)) & 1) ? lim_D[1 + s] : (T)0);
          sm.slotv[1][ks] = grad[1 + s];
        }
      }
      // per-contact weight in the contact frame: C = sum_r D_r w_r w_r^T with w = (1, +-mu, 0), (1, 0, +-mu)
      if (lane < ncon) {
        T Dr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) Dr[r] = ((act >> (4 + r)) & 1) ? con_D : (T)0;
        sm.cC[lane][0] = Dr[0] + Dr[1] + Dr[2] + Dr[3];
        sm.cC[lane][1] = con_mu * (Dr[0] - Dr[1]);
        sm.cC[lane][2] = con_mu * (Dr[2] - Dr[3]);
        sm.cC[lane][3] = con_mu * con_mu * (Dr[0] + Dr[1]);
        sm.cC[lane][4] = con_mu * con_mu * (Dr[2] + Dr[3]);
      }
      const bool isslot = !isl && lane < nsys;
      WSYNC();
      T sdiag = 0, rhs = isl ? grad[0] : (T)0;
      if (isslot) { sdiag = sm.slotv[0][lane - nl]; rhs = sm.slotv[1][lane - nl]; }
      const unsigned long long dmx = dirty_mask;
      if (dmx) {
        const int nD = __popcll(dmx);
        for (int i = lane; i < tri(nD + 1, 0); i += 64) sm.H[i] = 0;
      }
      // ... + J^T C J of every single-chain contact, gathered by the ROW lanes: the entries of such a
      // contact are the root path of its deepest link (entry e = the ancestor at depth e), then the key's
      // slot.  Row lane r of the contact forms u = C_c J_r and adds u . J_a for every ancestor a (and
      // itself): registers only, no LDS adds, fixed order.
      {
        const Topo tp = topo();
        const int pos = isl ? tp.depth - tp.TL : -2;
        const int shift = (isl && pos >= 0) ? tp.TL - TC : 0;
        // (every lane walks its own contacts, in ascending order: the same sums as a loop over all contacts,
        // in as many trips as the busiest lane needs; the header of a lane's next contact is fetched while
        // this one is worked on)
        unsigned cm = call;
        asm volatile("" : "+v"(cm));
        cm &= ~crossm;   // (cross-chain contacts: below)
        // (round 6) which local columns exist for this LINK lane, as one bit mask (bit k: column k): the per-column
        // predicate used to be rebuilt from depth / trunk length / chain position for every column of every contact
        // (six instructions per column, more than the column's arithmetic)
        unsigned hcol = 0;
#pragma unroll
        for (int k = 0; k < MD; k++)
          if (isl && (k < TC ? (k < tp.TL && k <= tp.depth) : (pos >= 0 && k - TC <= pos))) hcol |= 1u << k;
        // (one LDS round trip per contact, held together by a scheduling barrier: the header and weight of the lane's NEXT
        // contact are fetched with the entries of this one -- the compiler's own order took four to five dependent round
        // trips per contact: header, own entry, weight, two groups of columns)
        int c_n = cm ? __ffs(cm) - 1 : 0;
        int inf_n = sm.cinf[c_n];
        T Cn[5];
#pragma unroll
        for (int i = 0; i < 5; i++) Cn[i] = sm.cC[c_n][i];
        while (__ballot(cm != 0u) != 0ull) {
          const bool mem = cm != 0u;
          cm &= cm - 1u;   // (0 & 0xffffffff = 0)
          const int inf = inf_n;
          const T sn = Cn[0], a1 = Cn[1], a2 = Cn[2], b1 = Cn[3], b2 = Cn[4];
          c_n = cm ? __ffs(cm) - 1 : 0;
          const int base = inf & 255, cnt = (inf >> 8) & 255;
          const int eo_ = base + (isl ? tp.depth : cnt - 1);
          const int eo = (mem && eo_ < nent) ? eo_ : 0;
          // columns of this lane for this contact: a link's own columns, a slot lane's first npath columns
          const int npath = cnt - 1;   // slot lanes: the links of this contact
          const unsigned vm = mem ? (isl ? hcol : ((1u << npath) - 1u)) : 0u;
          // (branch-free: the reads of the columns go out together and the products are selected -- a
          // predicated block per column made every column wait for its own two LDS round trips: 2.2 k cycles per
          // contact, the largest single phase of the stage.  The ADDRESS is not selected any more (round 6): entry
          // base + depth-of-column of a column this lane does not have is some other entry of the list, or at worst a
          // few hundred bytes further inside this stage's own LDS block (base + 12 < LeanCaps::NE + 13); its product is
          // selected away.  base + k is then an immediate offset of the read for the trunk columns and base + shift + k
          // for the chain columns: no address arithmetic per column.)
          const T* const eT = &sm.entJ[base][0];            // trunk columns k < TC: entry base + k
          const T* const eC = &sm.entJ[base + shift][0];    // chain columns k >= TC: entry base + k + shift
          const T ja0 = sm.entJ[eo][0], ja1 = sm.entJ[eo][1], ja2 = sm.entJ[eo][2];
          T jb[MD][3];
#pragma unroll
          for (int k = 0; k < MD; k++) {
            const T* e = (k < TC ? eT : eC) + 3 * k;
            jb[k][0] = e[0]; jb[k][1] = e[1]; jb[k][2] = e[2];
          }
          inf_n = sm.cinf[c_n];
#pragma unroll
          for (int i = 0; i < 5; i++) Cn[i] = sm.cC[c_n][i];
          __builtin_amdgcn_sched_barrier(0);
          const T u0 = sn * ja0 + a1 * ja1 + a2 * ja2;
          const T u1 = a1 * ja0 + b1 * ja1;
          const T u2 = a2 * ja0 + b2 * ja2;
#pragma unroll
          for (int k = 0; k < MD; k++) {
            const T v = u0 * jb[k][0] + u1 * jb[k][1] + u2 * jb[k][2];
            Rr[k] += ((vm >> k) & 1u) ? v : (T)0;
          }
          if (mem && !isl) sdiag += u0 * ja0 + u1 * ja1 + u2 * ja2;
        }
      }
      // cross-chain contacts go to the packed dense block of the dirty rows (zeroed above; tree_solve adds the
      // Schur complement of the clean rows): entry lane a holds u = C_c J_a and walks the entries b <= a
      if (dmx) {
        auto cidx = [&](int l) -> int { return __popcll(dmx & lanemask_lt(l)); };
        WSYNC();
        for (int e0 = 0; e0 < nent; e0 += 64) {
          const bool valid = e0 + lane < nent;
          const int e = valid ? e0 + lane : nent - 1;
          const int m0 = sm.entM[e][0], m1 = sm.entM[e][1];
          const int ln = m0 & 63, c = (m0 >> 6) & 31;
          const int base = m1 & 255, rank = (m1 >> 16) & 255;
          const bool cross = ((m0 >> 15) & 1) != 0;
          if (__ballot(valid && cross) == 0ull) continue;
          const T ja0 = sm.entJ[e][0], ja1 = sm.entJ[e][1], ja2 = sm.entJ[e][2];
          const T* C = sm.cC[c];
          const T sn = C[0], a1 = C[1], a2 = C[2], b1 = C[3], b2 = C[4];
          const T u0 = sn * ja0 + a1 * ja1 + a2 * ja2;
          const T u1 = a1 * ja0 + b1 * ja1;
          const T u2 = a2 * ja0 + b2 * ja2;
#ifdef RPK_NO_T7
          const int cia = cross ? cidx(ln) : 0;
#else
          const int cia = cross ? ((m0 >> 16) & 63) : 0;
#endif
          for (int k0 = 0; k0 < maxm; k0 += 4) {
            int mb[4];
            T val[4], jq[4][3];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int b = base + k0 + u < nent ? base + k0 + u : nent - 1;
              mb[u] = sm.entM[b][0];
              jq[u][0] = sm.entJ[b][0]; jq[u][1] = sm.entJ[b][1]; jq[u][2] = sm.entJ[b][2];
            }
            __builtin_amdgcn_sched_barrier(0);   // (the four entries' reads in flight together)
#pragma unroll
            for (int u = 0; u < 4; u++) val[u] = u0 * jq[u][0] + u1 * jq[u][1] + u2 * jq[u][2];
#pragma unroll
            for (int u = 0; u < 4; u++) {
#ifdef RPK_NO_T7
              if (valid && cross && k0 + u <= rank) lds_add(&sm.H[tri(cia, cidx(mb[u] & 63))], val[u]);
#else
              if (valid && cross && k0 + u <= rank) lds_add(&sm.H[tri(cia, (mb[u] >> 16) & 63)], val[u]);
#endif
            }
          }
        }
      }
      WSYNC();
      PROF(4);
      const T x = tree_solve(Rr, sdiag, rhs, nkt, dmx);
      // (the mass-matrix row of M * search: its trip to L2 runs under the slot exchange and the two wave sums below)
      T Mrs[MD + 1];
      load_Mr(Mrs);
      T search[3];
      search[0] = isl ? -x : (T)0;
      if (isslot) sm.slotv[0][lane - nl] = -x;
      WSYNC();
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const int ks = myks(s);
        const T via_slot = sm.slotv[0][ks >= 0 ? ks : 0];
        const T own = -grad[1 + s] / (kM[s] + (((act >> (2 + s)) & 1) ? lim_D[1 + s] : (T)0));
        search[1 + s] = isk[s] ? (ks >= 0 ? via_slot : own) : (T)0;
      }
      WSYNC();
      PROF(5);
      const T snorm = uni(N::sqrt(wave_sum(search[0] * search[0] + search[1] * search[1] + search[2] * search[2])));
      if (!(snorm >= RPK_MINVAL)) break;
      // phi'(0) = grad . search (before grad goes out of use)
      const T f0g = uni(wave_sum(grad[0] * search[0] + grad[1] * search[1] + grad[2] * search[2]));
      const T Mv0 = mulM0(search[0], Mrs);
      PROF(35);
      RowsL jv;
      mulJ(search, jv);
      PROF(36);
      T g0, g1, g2;
      {
        const T rk1 = kM[0] * dq[1], rk2 = kM[1] * dq[2];
        g0 = uni((T)0.5 * wave_sum(r0 * dq[0] + rk1 * dq[1] + rk2 * dq[2]));
        g1 = uni(wave_sum(search[0] * r0 + search[1] * rk1 + search[2] * rk2));
        g2 = uni((T)0.5 * wave_sum(search[0] * Mv0 + search[1] * (kM[0] * search[1]) + search[2] * (kM[1] * search[2])));
      }
      PROF(6);
      // ---- exact line search [MJ: PrimalSearch].  Along the search direction every row's cost is a
      // quadratic in alpha while the row stays in one zone, q0 + alpha q1 + alpha^2 q2 with
      // q = D (jar^2 / 2, jar jv, jv^2 / 2) [MJ: PrimalPrepare]; an evaluation tests each row's zone at
      // alpha and sums the coefficients of the active rows [MJ: PrimalEval] -- formed where they are used.
      const T frf = (isl && lfloss > 0) ? lflR * lfloss : (T)-1;
      const bool any_con = ncon > 0;
      auto ls_eval = [&](T alpha, T& d1, T& d2, const bool with_cost) -> T {
        T s0 = 0, s1 = 0, s2 = 0;
        {
          // (round 6: the friction row's three zones as factors -- sg = -1 / 0 / +1 for the two linear zones, q = 1 in the
          // quadratic one, all 0 without a friction row -- on the same products in the same association: same bits, a
          // third of the instructions of the nested selects)
          const T xx = jar.fr + alpha * jv.fr;
          const bool has = frf >= 0, lo_ = xx <= -frf, hi_ = xx >= frf;
          const T sg = !has ? (T)0 : (lo_ ? (T)-1 : (hi_ ? (T)1 : (T)0));
          const T q = (!has || lo_ || hi_) ? (T)0 : (T)1;
          const T ql1 = lfloss * jar.fr, frs = lfloss * jv.fr;
          s1 = sg * frs + q * (lflD * jar.fr * jv.fr);
          s2 = q * ((T)0.5 * lflD * jv.fr * jv.fr);
          if (with_cost) {
            const T ql0 = -(T)0.5 * frf * lfloss;
            s0 = N::abs(sg) * ql0 + sg * ql1 + q * ((T)0.5 * lflD * jar.fr * jar.fr);
          }
        }
        // limit rows (a weight each) and the four pyramid rows of my contact (one weight: applied once to the
        // sums of the active rows' products) -- the stage is VALU-issue bound and the line search is its largest
        // consumer, so an evaluation forms as few products per row as the algebra allows
        // (round 6: a row's zone test is a 0 / 1 factor on ITS weighted row values, not a predicated block -- seven
        // blocks per evaluation, each a compare, two exec-mask writes and a branch around three or four multiply-adds,
        // were a quarter of this stage's instructions.  The factor multiplies ONE operand of every product, exactly
        // (x * 1 = x), so an active row contributes the same bits as before and an inactive one +-0.)
        T h0_ = 0, h2_ = 0;   // twice the alpha^0 / alpha^2 coefficients
#pragma unroll
        for (int s = 0; s < 3; s++) {
          const T xx = jar.lim[s] + alpha * jv.lim[s];
          const T m = xx < 0 ? (T)1 : (T)0;
          const T t = (lim_D[s] * jar.lim[s]) * m, u = (lim_D[s] * jv.lim[s]) * m;   // (weight 0 without a limit row)
          s1 += t * jv.lim[s]; h2_ += u * jv.lim[s];
          if (with_cost) h0_ += t * jar.lim[s];
        }
        if (any_con) {
          T c0_ = 0, c1_ = 0, c2_ = 0;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const T xx = jar.con[r] + alpha * jv.con[r];
            const T m = xx < 0 ? (T)1 : (T)0;
            const T ja = jar.con[r] * m, jb = jv.con[r] * m;
            c1_ += ja * jv.con[r]; c2_ += jb * jv.con[r];
            if (with_cost) c0_ += ja * jar.con[r];
          }
          const T Dp = hascon ? con_D : (T)0;
          s1 += Dp * c1_; h2_ += Dp * c2_;
          if (with_cost) h0_ += Dp * c0_;
        }
        s2 += (T)0.5 * h2_;
        if (with_cost) s0 += (T)0.5 * h0_;
        s1 = wave_sum(s1) + g1; s2 = wave_sum(s2) + g2;
        d1 = uni(s1 + (T)2 * alpha * s2);
        d2 = uni((T)2 * s2);
        if (!with_cost) return (T)0;
        s0 = wave_sum(s0) + g0;
        return uni(s0 + alpha * (s1 + alpha * s2));
      };
      const T gtol = uni(M.tolerance * M.ls_tolerance * snorm / scale);
      // phi(0) is the current cost, phi'(0) = grad . search, phi''(0) = -phi'(0) (H search = -grad)
      T f0, h0, c0;
      if constexpr (sizeof(T) == 8) {
        f0 = f0g; h0 = -f0; c0 = cost;
      } else {
        c0 = ls_eval((T)0, f0, h0, true);
      }
      struct LsPnt { T a, c, d0, d1; };
      auto ls_point = [&](T al, const bool with_cost) -> LsPnt {
        LsPnt p; p.a = uni(al); p.c = ls_eval(p.a, p.d0, p.d1, with_cost); return p;
      };
      T alpha = 0;
      if (h0 > 0) {
        const int max_ls = S.max_ls;
        int evals = 2;
        const LsPnt p0 = {(T)0, c0, f0, h0};
        LsPnt p1 = ls_point(-f0 / h0, true);
        if (p0.c < p1.c) p1 = p0;
        alpha = p1.a;
#ifndef RPK_X_LS1   // compile-only experiment: the line search cut down to its first Newton point
        if (!(N::abs(p1.d0) < gtol)) {
          const T dir = p1.d0 < 0 ? (T)1 : (T)-1;
          LsPnt p2 = p1;
          bool p2update = false, converged = false;
          while (p1.d0 * dir <= -gtol && evals < max_ls) {
            p2 = p1; p2update = true;
            p1 = ls_point(p1.a - p1.d0 / p1.d1, false); evals++;
            if (N::abs(p1.d0) < gtol) { converged = true; break; }
          }
          alpha = p1.a;
          if (!converged && evals < max_ls && p2update) {
            // bracketed search: per round the midpoint and the Newton successors of both bracket ends are the
            // candidates; the cheapest one with |phi'| < gtol wins, otherwise each end moves to the candidate on
            // its side of the root whose slope is closest to zero [MJ: updateBracket].  (Both ends choose among
            // the round's candidates first and their successors are evaluated afterwards, in the same order as
            // the sequential procedure: no copies of the candidates are held.)
            LsPnt p2next = p1;
            LsPnt p1next = ls_point(p1.a - p1.d0 / p1.d1, true); evals++;
            auto choose = [&](LsPnt& p, const LsPnt& ca, const LsPnt& cb, const LsPnt& cc) -> bool {
              bool moved = false;
              T a_ = p.a, d0_ = p.d0, d1_ = p.d1;
              auto consider = [&](const LsPnt& c) {
                if ((d0_ < 0 && c.d0 < 0 && d0_ < c.d0) || (d0_ > 0 && c.d0 > 0 && d0_ > c.d0)) { a_ = c.a; d0_ = c.d0; d1_ = c.d1; moved = true; }
              };
              consider(ca); consider(cb); consider(cc);
              p.a = a_; p.d0 = d0_; p.d1 = d1_;
              return moved;
            };
            bool settled = false;
            while (evals < max_ls) {
              const LsPnt pmid = ls_point((T)0.5 * (p1.a + p2.a), true); evals++;
              T bestc = 0; bool found = false;
              if (N::abs(p1next.d0) < gtol) { found = true; bestc = p1next.c; alpha = p1next.a; }
              if (N::abs(p2next.d0) < gtol && (!found || p2next.c < bestc)) { found = true; bestc = p2next.c; alpha = p2next.a; }
              if (N::abs(pmid.d0) < gtol && (!found || pmid.c < bestc)) { found = true; alpha = pmid.a; }
              if (found) { settled = true; break; }
              const bool b1 = choose(p1, p1next, p2next, pmid);
              const bool b2 = choose(p2, p1next, p2next, pmid);
              if (b1) { p1next = ls_point(p1.a - p1.d0 / p1.d1, true); evals++; }
              if (b2) { p2next = ls_point(p2.a - p2.d0 / p2.d1, true); evals++; }
              if (!b1 && !b2) { alpha = pmid.a; settled = true; break; }
            }
            if (!settled) {
              T t1, t2;
              const T c1 = ls_eval(p1.a, t1, t2, true), c2 = ls_eval(p2.a, t1, t2, true);
              alpha = (c1 <= c2 && c1 < p0.c) ? p1.a : ((c2 <= c1 && c2 < p0.c) ? p2.a : (T)0);
            }
          }
        }
#endif
      }
      PROF(7);
      if (!(alpha > 0)) break;
#pragma unroll
      for (int s = 0; s < 3; s++) dq[s] += alpha * search[s];
      r0 += alpha * Mv0;
      jar.fr += alpha * jv.fr;
#pragma unroll
      for (int s = 0; s < 3; s++) jar.lim[s] += alpha * jv.lim[s];
#pragma unroll
      for (int r = 0; r < 4; r++) jar.con[r] += alpha * jv.con[r];
      const T oldcost = cost;
      cost = uni(wave_sum(update(jar) + gauss()));
      mulJT(jar, qfc);
      niter_last = iter + 1;
      T gn;
      {
        const T ga = r0 - qfc[0], gb = kM[0] * dq[1] - qfc[1], gc = kM[1] * dq[2] - qfc[2];
        gn = uni(N::sqrt(wave_sum(ga * ga + gb * gb + gc * gc)));
      }
      PROF(8);
      if (scale * (oldcost - cost) < M.tolerance || scale * gn < M.tolerance) break;
    }
  }
  // forces of the pyramidal contact rows, for the acceleration-stage sensors (MODE 2)
  if (S.con_force && lane < RPK_NC) {
#pragma unroll
    for (int r = 0; r < 4; r++)
      S.con_force[((size_t)env * RPK_NC + lane) * 4 + r] =
          (anyrow && lane < ncon && ((act >> (4 + r)) & 1)) ? -con_D * jar.con[r] : (T)0;
  }
  PROF(8);
  // ---- Euler with implicit joint damping [MJ: mj_Euler, eulerdamp]; qfrc_smooth comes back from its parking slots
  // (the epilogue's reads in two batches IN FRONT of the last solve -- dof indices, parked forces, damping, my mass-matrix
  // row; then the state at those indices -- so that their trips to L2 run under the solve: read where they were used they
  // were nine dependent trips behind it)
  T qe[3];
  const int e_ld = fresh(M.lane_topo())[16 * L + 5];   // my dof
  int e_kd[2];
  T e_fk[2], e_kdamp[2];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int K = isk[s] ? kid[s] : 0;
    e_kd[s] = M.key_dof()[K]; e_kdamp[s] = M.key_damping()[K];
  }
  e_fk[0] = LF(3); e_fk[1] = LF(4);
  const T e_f0 = LF(0);
  const T e_ldamp = M.link_damping()[L];
  T Rr_e[MD + 1];
  fetch_Mlocal(Rr_e);
  __builtin_amdgcn_sched_barrier(0);
  const T e_qv0 = S.qvel[eo + e_ld], e_qp0 = S.qpos[eo + e_ld], e_w0 = S.warm[eo + e_ld];
  T e_qvk[2], e_qpk[2];
#pragma unroll
  for (int s = 0; s < 2; s++) { e_qvk[s] = S.qvel[eo + e_kd[s]]; e_qpk[s] = S.qpos[eo + e_kd[s]]; }
  __builtin_amdgcn_sched_barrier(0);
  {
    const T f0_ = e_f0;
    const T ldamp = isl ? e_ldamp : (T)0;
    finish_Mlocal(Rr_e, h * ldamp);
    qe[0] = tree_solve(Rr_e, (T)0, f0_ + qfc[0], 0, 0ull);
  }
  PROF(9);
  // ---- new state (S.warm holds qacc_smooth of my hand dof, the keys' is qfrc_smooth / kM)
  if (isl) {
    const int ld = e_ld;
    const T qd0 = e_qv0 + h * qe[0];
    S.qvel[eo + ld] = qd0;
    S.qpos[eo + ld] = e_qp0 + h * qd0;
    S.warm[eo + ld] = e_w0 + dq[0];
  }
#pragma unroll
  for (int s = 0; s < 2; s++) if (isk[s]) {
    const int kd = e_kd[s];
    const T fk = e_fk[s];
    const T qek = (fk + qfc[1 + s]) / (kM[s] + h * e_kdamp[s]);
    const T qdk = e_qvk[s] + h * qek;
    S.qvel[eo + kd] = qdk;
    S.qpos[eo + kd] = e_qpk[s] + h * qdk;
    S.warm[eo + kd] = fk / kM[s] + dq[1 + s];
  }
  {
    int w = wave_or(warn);
    if (lane == 0) {
      if (w) S.warn[env] |= w;
      S.solver_iter[env] = (niter_last & 255) | ((__popcll(dirty_mask) & 255) << 8) | ((nkt & 255) << 16);
      S.time[env] += h;
    }
  }
  if (S.prof && env == 0 && lane < RPK_NPROF) {
    WSYNC();
    atomicAdd((unsigned long long*)&S.prof[lane], (unsigned long long)sm.prof[lane]);
  }
  if (S.cost_sol && lane == 0) S.cost_sol[env] = (int)(((long long)__builtin_readcyclecounter() - kernel_t0) >> 8);
#undef LF
#undef LI
}

template <typename T>
__global__ __launch_bounds__(64, 2) void rp_lean_solver_kernel(RpModel<T> M, RpState<T> S, RpStage<T> B) {
  const int env = S.order ? S.order[S.env_base + blockIdx.x] : S.env_base + (int)blockIdx.x;
  rp_lean_solver_body<T, false>(M, S, B, env, nullptr, (int)threadIdx.x);
}
