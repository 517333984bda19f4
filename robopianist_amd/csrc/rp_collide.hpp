// rp_collide.hpp -- the POOLED narrow phase of the split position stage (round 5).
//
// The one-kernel position stage (rp_stage_body<T, 0>, PART 0) runs its narrow phase with one wave per env: the ~25
// candidates an env has after the fp32 prefilters sit in ~25 lanes, and the wave walks EVERY routine a candidate of
// its pass needs -- capsule-capsule, capsule-box, box-box, the portal refinement of the hull pairs -- one after the
// other with ~10 of 64 lanes working (measured: a third of the stage's shader cycles).  Here the candidates of ALL envs
// of a launch are pooled by pair type (rp_stage_body<T, 0, ..., PART 1> appends them to RpStage::tlist): a wave takes 64
// candidates of ONE type, whatever envs they came from, and walks one routine with every lane busy.  Results go back
// to the candidate's own records (RpStage::cres / cres_n), so where a candidate landed on the pooled list -- which
// depends on the order in which the front part's waves arrive -- never shows in the output.
//
// Reference: the work is MuJoCo's mj_collision narrow phase (mjc_Capsule / mjc_CapsuleBox / mjc_BoxBox / mjc_Convex)
// + mj_contactParam / mj_makeImpedance, reached from physics.step() of the task
// (/root/reference/robopianist/suite/tasks/base.py:28,31,68-70); the arithmetic per candidate is exactly the one-kernel
// stage's (same routines of rp_narrow.hpp on the same inputs: same bits).
#pragma once
#include "rp_model.hpp"
#include "rp_wave.hpp"
#include "rp_narrow.hpp"

// The scanned hulls' vertex table in LDS (hull_support_wave<.., LDSV>): one variable, named from the kernel (which fills
// it) and from the hull routine (which reads it) through this accessor, so that both see an LDS address, not a generic one.
template <typename T> __device__ __forceinline__ T* rp_lds_vert() {
  __shared__ T v[3 * RPK_MAXMESHV];
  return v;
}
// One pair type per routine, each a REAL call (noinline): the kernel then needs the registers of its hungriest routine,
// not of all four inlined side by side (256 VGPRs + 96 AGPRs, one wave per SIMD -- a workgroup that has to wait for a
// whole idle SIMD while the other slice's kernels hold the chip), and nothing of the caller lives across the call.
// TYPE: 0 capsule-capsule, 1 capsule-box, 2 box-box, 3 .. 6 hull pairs (MPR; MESH builds; the buckets of rp_model.hpp).  `e` = this lane's entry on the
// type's pooled list, `in` = it exists.
// (LDSV: the scanned hulls' vertex table is the copy in LDS, rp_lds_vert -- the pooled kernel; false: the model's table
// through the scalar cache, as in the one-kernel stage -- the fused schedule, whose LDS belongs to the stage bodies)
template <typename T, int MESH, int TYPE, bool LDSV = true>
__device__ __noinline__ void rp_narrow_pair(const RpModel<T>& M, const RpState<T>& S, const RpStage<T>& B, const int env_, const int pair_, const int rb_,
                                            const int ci_, const bool in) {
  using namespace rpk;
  using N = Num<T>;
  const T h = M.timestep;
  const int env = in ? env_ : 0, pair = in ? pair_ : 0, rb = in ? rb_ : 0, ci = in ? ci_ : 0;
  const int ga = pair & 0xffff, gb = (pair >> 16) & 0xffff;
  const bool key = gb >= RPK_KEYBASE;
  const int kk = key ? gb - RPK_KEYBASE : 0, gbi = key ? 0 : gb;
  // side A: a hand geom; side B: a hand geom, or the key's box (moving with the key's hinge angle)
  // (a geom's world frame from its link's frame, RpStage::frames, with the one-kernel stage's expressions: the front
  // part would pay 9 k cycles per wave to write world frames of all its geoms; this kernel's waves mostly wait anyway)
  auto geom_frame = [&](const int g, T* pos, T* mat) {
    const int gl = M.geom_link()[g];
    const T* gp = M.geom_pos() + 3 * g;
    const T* gm = M.geom_mat() + 9 * g;
    if (gl >= 0) {
      const T* fr = B.frames + (size_t)env * RPK_NFRAME * 64 + gl;
      T xp[3], xm[9], t[3];
#pragma unroll
      for (int k = 0; k < 3; k++) xp[k] = fr[(size_t)k * 64];
#pragma unroll
      for (int k = 0; k < 9; k++) xm[k] = fr[(size_t)(3 + k) * 64];
      mat_vec(t, xm, gp);
      pos[0] = xp[0] + t[0]; pos[1] = xp[1] + t[1]; pos[2] = xp[2] + t[2];
      mat_mul(mat, xm, gm);
    } else {
#pragma unroll
      for (int k = 0; k < 3; k++) pos[k] = gp[k];
#pragma unroll
      for (int k = 0; k < 9; k++) mat[k] = gm[k];
    }
  };
  T posA[3], mA[9], posB[3], mB[9];
  const T* sB;
  geom_frame(ga, posA, mA);
  T invw = M.geom_invw()[ga];
  const T* pB;
  if (key) {
    T s, c;
    N::sincos(S.qpos[(size_t)env * M.nv + M.key_dof()[kk]], &s, &c);
    const T hx = M.key_half()[3 * kk];
    posB[0] = M.key_pos()[3 * kk] - hx + hx * c; posB[1] = M.key_pos()[3 * kk + 1]; posB[2] = M.key_pos()[3 * kk + 2] - hx * s;
    mB[0] = c; mB[1] = 0; mB[2] = s; mB[3] = 0; mB[4] = 1; mB[5] = 0; mB[6] = -s; mB[7] = 0; mB[8] = c;
    sB = M.key_half() + 3 * kk;
    pB = M.key_cparam();
    invw += M.key_invw_body()[kk];
  } else {
    geom_frame(gbi, posB, mB);
    sB = M.geom_size() + 3 * gbi;
    pB = M.geom_cparam() + 8 * gbi;
    invw += M.geom_invw()[gbi];
  }
  RawCon<T> rc[3];
  RawCon<T> bbx[TYPE == 2 ? 5 : 1];   // (box-box points four to eight)
  int n = 0;
  if constexpr (TYPE == 0) {
    if (in) n = capsule_capsule(rc, posA, mA, M.geom_size() + 3 * ga, posB, mB, sB);
  } else if constexpr (TYPE == 1) {
    if (in) n = capsule_box(rc, posA, mA, M.geom_size() + 3 * ga, posB, mB, sB);
  } else if constexpr (TYPE == 2) {
    if (in) n = box_box(rc, bbx, posA, mA, M.geom_size() + 3 * ga, posB, mB, sB);
  } else if constexpr (MESH != 0) {
    // (box, hull) in geom-type order when the partner is a key: the key is geom 1 of the pair; the engine keeps the
    // hand geom as side A of the contact, so the normal is turned around afterwards
    CGeom<T> a_, b_;
    auto hull_of = [&](CGeom<T>& g, const int type, const int geom) {
      const bool hull = type == GEOM_MESH_ && geom >= 0;
      const int gi = geom >= 0 ? geom : 0;
      g.type = type;
      g.nvert = hull ? M.geom_vertnum()[gi] : 0;
      g.vadr = hull ? M.geom_vertadr()[gi] : 0;
      g.flip = hull ? M.geom_vertflip()[gi] : 0;
      g.graph = (MESH > 1 && hull) ? M.geom_vertgraph()[gi] : 0;
    };
    const bool cylkey = MESH > 1 && key && M.geom_type()[ga] == GEOM_CYL_;   // (cylinder, box): the hand geom IS geom 1
    if (key && !cylkey) {
      hull_of(a_, GEOM_BOX_, -1); hull_of(b_, GEOM_MESH_, ga);
#pragma unroll
      for (int i = 0; i < 3; i++) { a_.pos[i] = posB[i]; a_.size[i] = sB[i]; b_.pos[i] = posA[i]; b_.size[i] = M.geom_size()[3 * ga + i]; }
#pragma unroll
      for (int i = 0; i < 9; i++) { a_.mat[i] = mB[i]; b_.mat[i] = mA[i]; }
    } else {
      hull_of(a_, M.geom_type()[ga], ga); hull_of(b_, key ? GEOM_BOX_ : (MESH > 1 ? M.geom_type()[gbi] : GEOM_MESH_), key ? -1 : gbi);
#pragma unroll
      for (int i = 0; i < 3; i++) { a_.pos[i] = posA[i]; a_.size[i] = M.geom_size()[3 * ga + i]; b_.pos[i] = posB[i]; b_.size[i] = sB[i]; }
#pragma unroll
      for (int i = 0; i < 9; i++) { a_.mat[i] = mA[i]; b_.mat[i] = mB[i]; }
    }
    RawCon<T> rcm[1];
    const T* vtab;
    if constexpr (LDSV) vtab = rp_lds_vert<T>(); else vtab = M.mesh_vert();
    const int nm = convex_mpr_wave<T, (MESH > 1), LDSV>(rcm, &a_, &b_, vtab, M.hull_vert, M.hull_graph, in, M.mpr_tol, M.mpr_tol_poly);
    if (in) {
      n = nm; rc[0] = rcm[0];
      if (key && !cylkey) { rc[0].n[0] = -rc[0].n[0]; rc[0].n[1] = -rc[0].n[1]; rc[0].n[2] = -rc[0].n[2]; }
    }
  }
  // ---- the records: contact + its parameters [MJ: mj_contactParam, mj_makeImpedance] (as the one-kernel stage's `emit`)
  if (in) {
    const T* pA = M.geom_cparam() + 8 * ga;
    T solref0 = (T)0.5 * (pA[0] + pB[0]);
    const T solref1 = (T)0.5 * (pA[1] + pB[1]);
    T solimp[5];
#pragma unroll
    for (int q = 0; q < 5; q++) solimp[q] = (T)0.5 * (pA[2 + q] + pB[2 + q]);
    const T mu = fmax(pA[7], pB[7]);
    if (solref0 > 0) solref0 = fmax(solref0, (T)2 * h);
    const T dmax = fmin((T)0.9999, fmax((T)0.0001, solimp[1]));
    const T Kc = (T)1 / fmax(RPK_MINVAL, dmax * dmax * solref0 * solref0 * solref1 * solref1);
    const T Bc = (T)2 / fmax(RPK_MINVAL, dmax * solref0);
    T* const res = B.cres + ((size_t)env * RPK_NRES + rb) * 12;
    auto put = [&](const int slot, const RawCon<T>& r) {
      const T imp = impedance(solimp, r.dist);
      const T Rn = fmax(RPK_MINVAL, ((T)1 - imp) * invw * ((T)1 + mu * mu) / imp);
      const T mur = mu * M.mu_scale;   // (opt.impratio: the regularised friction coefficient)
      const T Rpy = fmax(RPK_MINVAL, (T)2 * mur * mur * Rn);
      T* o = res + (size_t)slot * 12;
#pragma unroll
      for (int k = 0; k < 3; k++) { o[k] = r.pos[k]; o[3 + k] = r.n[k]; }
      o[6] = r.dist; o[7] = mu; o[8] = Kc * imp * r.dist; o[9] = Bc; o[10] = (T)1 / Rpy;
    };
#pragma unroll
    for (int slot = 0; slot < (TYPE >= 3 ? 1 : (TYPE == 2 ? 3 : 2)); slot++) if (n > slot) put(slot, rc[slot]);
    if constexpr (TYPE == 2) {
      if (RPK_BOXBOX_MAX > 3 && n > 3) {
        for (int slot = 3; slot < n && slot < 8; slot++) put(slot, bbx[slot - 3]);
      }
    }
    B.cres_n[(size_t)env * RPK_NCAND + ci] = n;
  }
}

// a chunk of one pooled list: lane's entry e of the type's list (its env, pair, result records, candidate index)
template <typename T, int MESH, int TYPE>
__device__ __forceinline__ void rp_narrow_chunk(const RpModel<T>& M, const RpState<T>& S, const RpStage<T>& B, const int stripe, const int e, const bool in) {
  int env = 0, pair = 0, rb = 0, ci = 0;
  if (in) {
    const int4 rec = *(const int4*)(B.tlist + (((size_t)TYPE * RPK_NSTRIPE + stripe) * B.tstride + (size_t)(S.env_base / RPK_NSTRIPE) * RPK_NCAND + e) * 4);
    env = rec.x; pair = rec.y; rb = rec.z; ci = rec.w;
  }
  rp_narrow_pair<T, MESH, TYPE, true>(M, S, B, env, pair, rb, ci, in);
}
// The narrow phase of ONE env inside its own wave (fused substeps with the split stage's bodies, rp_fused_split_kernel):
// lane = candidate of this env, one real call per pair type present -- the same routines on the same inputs as the pooled
// kernel's, so the same bits; the hull buckets 3 .. 6 exist for pooling only and share routine 3 here.
template <typename T, int MESH>
__device__ __forceinline__ void rp_narrow_env(const RpModel<T>& M, const RpState<T>& S, const RpStage<T>& B, const int env, const int lane) {
  const int nc = B.ncand[env];
  const int* const cl = B.cand + (size_t)env * RPK_NCAND * 2;
  for (int c0 = 0; c0 < nc; c0 += 64) {
    const int i = c0 + lane;
    const bool in = i < nc;
    const int pair = in ? cl[2 * i] : 0, meta = in ? cl[2 * i + 1] : 0;
    const int rb = meta & 0xffff, ty = (meta >> 16) & 15;
    const bool ok = in && ((meta >> 20) & 1);   // (its result records fit: RPK_NRES)
    if constexpr (MESH != 0) {
      if (__ballot(ok && ty >= 3) != 0ull) rp_narrow_pair<T, MESH, 3, false>(M, S, B, env, pair, rb, i, ok && ty >= 3);
    }
    if (__ballot(ok && ty == 2) != 0ull) rp_narrow_pair<T, MESH, 2, false>(M, S, B, env, pair, rb, i, ok && ty == 2);
    if (__ballot(ok && ty == 1) != 0ull) rp_narrow_pair<T, MESH, 1, false>(M, S, B, env, pair, rb, i, ok && ty == 1);
    if (__ballot(ok && ty == 0) != 0ull) rp_narrow_pair<T, MESH, 0, false>(M, S, B, env, pair, rb, i, ok && ty == 0);
  }
}

#ifndef RPK_NARROW_WAVES
#define RPK_NARROW_WAVES 2
#endif
template <typename T, int MESH>
__global__ __launch_bounds__(64, RPK_NARROW_WAVES) void rp_narrow_kernel(RpModel<T> M, RpState<T> S, RpStage<T> B) {
  const int lane = (int)threadIdx.x;
  const int* const tc = B.tcount + B.tcount_off;
  // the scanned hulls' vertex table, staged in LDS by the waves that walk a hull list (hull_support_wave<.., LDSV>)
  bool staged = false;
  // chunks of candidates of one list: the long routines first (hull buckets, box-box, capsule-box, capsule-capsule)
  int first = 0;   // (first chunk of the list under test)
#define RP_NARROW_LIST(TYPE_, LANES_)                                                                  \
  for (int st_ = 0; st_ < RPK_NSTRIPE; st_++) {                                                        \
    const int cnt_ = tc[st_ * RPK_NTYPE_PAD + TYPE_], nch_ = (cnt_ + (LANES_) - 1) / (LANES_);         \
    int ch = (int)blockIdx.x;                                                                          \
    if (ch < first) ch += (first - ch + (int)gridDim.x - 1) / (int)gridDim.x * (int)gridDim.x;         \
    for (; ch < first + nch_; ch += (int)gridDim.x) {                                                  \
      const int e = (ch - first) * (LANES_) + lane;                                                    \
      const long long t0_ = S.prof ? (long long)__builtin_readcyclecounter() : 0;                      \
      if (MESH != 0 && (TYPE_) >= 3 && !staged) {                                                      \
        for (int i_ = lane; i_ < 3 * RPK_MAXMESHV; i_ += 64) rp_lds_vert<T>()[i_] = M.mesh_vert()[i_]; \
        staged = true;                                                                                 \
        __syncthreads();                                                                               \
      }                                                                                                \
      rp_narrow_chunk<T, MESH, TYPE_>(M, S, B, st_, e, lane < (LANES_) && e < cnt_);                   \
      if (S.prof && lane == 0) {   /* (debug aid: shader cycles and chunks per list) */                \
        atomicAdd((unsigned long long*)&S.prof[48 + (TYPE_)], (unsigned long long)((long long)__builtin_readcyclecounter() - t0_)); \
        atomicAdd((unsigned long long*)&S.prof[56 + (TYPE_)], 1ull);                                   \
      }                                                                                                \
    }                                                                                                  \
    first += nch_;                                                                                     \
  }
  if constexpr (MESH != 0) {
    RP_NARROW_LIST(6, RPK_MPR_LANES) RP_NARROW_LIST(5, RPK_MPR_LANES) RP_NARROW_LIST(4, RPK_MPR_LANES) RP_NARROW_LIST(3, RPK_MPR_LANES)
  }
  RP_NARROW_LIST(2, 64) RP_NARROW_LIST(1, 64) RP_NARROW_LIST(0, 64)
#undef RP_NARROW_LIST
}
