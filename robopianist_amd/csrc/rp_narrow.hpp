// rp_narrow.hpp — narrow-phase collision primitives (sphere/capsule/box), one candidate
// pair per lane [MJ: engine_collision_primitive.c / engine_collision_box.c, restated].
#pragma once
#ifdef RP_MPR_COUNT
#include <cstdio>
#endif
#include "rp_wave.hpp"

namespace rpk {
// ----------------------------------------------------------------- narrow phase
template <typename T> struct RawCon { T dist, pos[3], n[3]; };

template <typename T>
__device__ __forceinline__ int sphere_sphere(RawCon<T>* c, const T* c1, T r1, const T* c2, T r2) {
  T v[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
  T len = Num<T>::sqrt(dot3(v, v)), dist = len - r1 - r2;
  if (dist > (T)0) return 0;
  if (len < RPK_MINVAL) { v[0] = 1; v[1] = 0; v[2] = 0; }
  else { T inv = (T)1 / len; v[0] *= inv; v[1] *= inv; v[2] *= inv; }
  c->dist = dist;
#pragma unroll
  for (int k = 0; k < 3; k++) { c->n[k] = v[k]; c->pos[k] = c1[k] + v[k] * (r1 + (T)0.5 * dist); }
  return 1;
}

// out[n++] = c with a static register index (out[] must never be indexed dynamically,
// or the compiler places it in scratch memory)
template <typename T>
__device__ __forceinline__ void put_con(RawCon<T>* out, int& n, const RawCon<T>& c) {
#pragma unroll
  for (int i = 0; i < 3; i++) if (n == i) out[i] = c;
  n++;
}

template <typename T>
__device__ int capsule_capsule(RawCon<T>* out, const T* p1, const T* m1, const T* s1, const T* p2,
                               const T* m2, const T* s2) {
  T a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]};
  T r1 = s1[0], l1 = s1[1], r2 = s2[0], l2 = s2[1];
  T dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  T b = dot3(a1, a2), u = -dot3(a1, dif), v = dot3(a2, dif);
  T det = (T)1 - b * b;
  int n = 0;
  T c1[3], c2[3];
  if (det > (T)1e-10) {
    T x1 = (u + b * v) / det, x2 = (v + b * u) / det;
    if (x1 > l1) { x1 = l1; x2 = v + b * x1; } else if (x1 < -l1) { x1 = -l1; x2 = v + b * x1; }
    if (x2 > l2) { x2 = l2; x1 = fmin(l1, fmax(-l1, u + b * x2)); }
    else if (x2 < -l2) { x2 = -l2; x1 = fmin(l1, fmax(-l1, u + b * x2)); }
#pragma unroll
    for (int k = 0; k < 3; k++) { c1[k] = p1[k] + a1[k] * x1; c2[k] = p2[k] + a2[k] * x2; }
    RawCon<T> rc;
    if (sphere_sphere(&rc, c1, r1, c2, r2)) put_con(out, n, rc);
  } else {
    T sgn = b >= 0 ? (T)1 : (T)-1, mid = u;
    T lo = fmax(-l1, mid - l2), hi = fmin(l1, mid + l2);
    if (lo <= hi) {
      int cnt = (hi - lo > (T)1e-12) ? 2 : 1;
      for (int q = 0; q < cnt; q++) {
        T x1 = q == 0 ? lo : hi, x2 = sgn * (x1 - mid);
#pragma unroll
        for (int k = 0; k < 3; k++) { c1[k] = p1[k] + a1[k] * x1; c2[k] = p2[k] + a2[k] * x2; }
        RawCon<T> rc;
        if (sphere_sphere(&rc, c1, r1, c2, r2)) put_con(out, n, rc);
      }
    } else {
      T x1 = mid > 0 ? l1 : -l1;
      T x2 = fmin(l2, fmax(-l2, sgn * (x1 - mid)));
#pragma unroll
      for (int k = 0; k < 3; k++) { c1[k] = p1[k] + a1[k] * x1; c2[k] = p2[k] + a2[k] * x2; }
      RawCon<T> rc;
      if (sphere_sphere(&rc, c1, r1, c2, r2)) put_con(out, n, rc);
    }
  }
  return n;
}

template <typename T>
__device__ __forceinline__ T seg_box_g(const T* c, const T* a, const T* h, T t) {
  T g = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    T p = c[k] + t * a[k];
    if (p > h[k]) g += a[k] * (p - h[k]); else if (p < -h[k]) g += a[k] * (p + h[k]);
  }
  return g;
}

template <typename T>
__device__ __forceinline__ int sphere_box_local(RawCon<T>* c, const T* p, T r, const T* h) {
  T q[3], v[3], d2 = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    q[k] = p[k] > h[k] ? h[k] : (p[k] < -h[k] ? -h[k] : p[k]);
    v[k] = p[k] - q[k]; d2 += v[k] * v[k];
  }
  T nbs[3], dist;
  if (d2 > 0) {
    T dd = Num<T>::sqrt(d2);
    dist = dd - r;
    T inv = (T)1 / dd;
    nbs[0] = v[0] * inv; nbs[1] = v[1] * inv; nbs[2] = v[2] * inv;
  } else {
    int ax = 0; T best = (T)-1e30;
#pragma unroll
    for (int k = 0; k < 3; k++) { T pen = Num<T>::abs(p[k]) - h[k]; if (pen > best) { best = pen; ax = k; } }
    nbs[0] = nbs[1] = nbs[2] = 0;
    T sg = p[ax] >= 0 ? (T)1 : (T)-1;
    nbs[ax] = sg; q[ax] = sg * h[ax];
    dist = best - r;
  }
  if (dist > 0) return 0;
  c->dist = dist;
#pragma unroll
  for (int k = 0; k < 3; k++) { c->pos[k] = q[k] + nbs[k] * (T)0.5 * dist; c->n[k] = -nbs[k]; }
  return 1;
}

// signed distance of a sphere (local centre p, radius r) to the box of half sizes h
template <typename T>
__device__ __forceinline__ T sphere_box_dist(const T* p, T r, const T* h) {
  T d2 = 0, best = (T)-1e30;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const T q = p[k] > h[k] ? h[k] : (p[k] < -h[k] ? -h[k] : p[k]);
    d2 += (p[k] - q) * (p[k] - q);
    const T pen = Num<T>::abs(p[k]) - h[k];
    if (pen > best) best = pen;
  }
  return (d2 > 0 ? Num<T>::sqrt(d2) : best) - r;
}

// capsule (geom1) vs box (geom2), at most two contacts [MJ: mjc_CapsuleBox's contract]: the axis point
// closest to the box (exact root of the piecewise linear distance derivative), then the segment end
// that lies deeper in / closer to the box.
template <typename T>
__device__ int capsule_box(RawCon<T>* out, const T* cp, const T* cm, const T* cs, const T* bp,
                           const T* bm, const T* bs) {
  T r = cs[0], l = cs[1];
  T ax[3] = {cm[2], cm[5], cm[8]}, rel[3] = {cp[0] - bp[0], cp[1] - bp[1], cp[2] - bp[2]};
  T c[3], a[3];
  matT_vec(c, bm, rel);
  matT_vec(a, bm, ax);
  a[0] *= l; a[1] *= l; a[2] *= l;
  T tstar;
  T gm = seg_box_g(c, a, bs, (T)-1), gp = seg_box_g(c, a, bs, (T)1);
  if (gm >= 0) tstar = -1;
  else if (gp <= 0) tstar = 1;
  else {
    T tl = -1, gl = gm, tr = 1, gr = gp;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (Num<T>::abs(a[k]) > RPK_MINVAL) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
          T tt = ((s == 0 ? bs[k] : -bs[k]) - c[k]) / a[k];
          if (tt > -1 && tt < 1) {
            T g = seg_box_g(c, a, bs, tt);
            if (g <= 0 && tt > tl) { tl = tt; gl = g; }
            if (g >= 0 && tt < tr) { tr = tt; gr = g; }
          }
        }
      }
    }
    if (tr <= tl) tstar = tl;
    else if (gr - gl > 0) tstar = tl + (tr - tl) * (-gl) / (gr - gl);
    else tstar = tl;
  }
  // the deeper end (ties: the -1 end); if it is t* itself, the other one
  T dend[2];
#pragma unroll
  for (int e = 0; e < 2; e++) {
    const T te = e ? (T)1 : (T)-1;
    T p[3] = {c[0] + te * a[0], c[1] + te * a[1], c[2] + te * a[2]};
    dend[e] = sphere_box_dist(p, r, bs);
  }
  T tend = dend[0] <= dend[1] ? (T)-1 : (T)1;
  if (Num<T>::abs(tend - tstar) < (T)1e-9) tend = -tend;
  int n = 0;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    T tc = i == 0 ? tstar : tend;
    if (i > 0 && Num<T>::abs(tc - tstar) < (T)1e-9) continue;
    T p[3] = {c[0] + tc * a[0], c[1] + tc * a[1], c[2] + tc * a[2]};
    RawCon<T> rc;
    if (sphere_box_local(&rc, p, r, bs)) {
      T w[3];
      RawCon<T> wc;
      mat_vec(w, bm, rc.pos);
      wc.pos[0] = bp[0] + w[0]; wc.pos[1] = bp[1] + w[1]; wc.pos[2] = bp[2] + w[2];
      mat_vec(wc.n, bm, rc.n);
      wc.dist = rc.dist;
      put_con(out, n, wc);
    }
  }
  return n;
}
// ---------------------------------------------------------------------------------------------
// box (geom1) vs box (geom2) [MJ: mjc_BoxBox's contract]: separating-axis test over the 15 axes; a
// face axis of least penetration -> the part of the other box's facing face inside the reference
// face's prism (its vertices, the reference corners under it, the edge crossings), all with the face
// normal; an edge-edge axis -> one contact at the closest points of the two edges.  Position midway
// between the surfaces, dist = -penetration, normal from geom1 to geom2.  A quadrilateral clipped by a
// rectangle has at most eight corners: all of them are kept, in emission order (up to RPK_BOXBOX_MAX = 8,
// [MJ: mjc_BoxBox's maximum]).  The first three go to out[0..2] through registers (static indices: selects), points
// four to eight -- two faces resting on each other -- to `extra[0..4]`, memory-resident and indexed dynamically
// (measured: all eight through memory cost the hull position stage 10 %: 16 candidate sites x 7 scratch stores on
// every mj_step of the stand-in hand, whose forearm box always touches its palm boxes).
#ifndef RPK_BOXBOX_MAX   // (experiments: 3 = the contract of rounds 1-3 without the depth ordering)
#define RPK_BOXBOX_MAX 8
#endif
template <typename T> __device__ __forceinline__ T pick3(T a0, T a1, T a2, int i) { return i == 0 ? a0 : (i == 1 ? a1 : a2); }

#ifndef RPK_BOXBOX_INLINE
#define RPK_BOXBOX_INLINE __forceinline__   // (measured: +1.6 % hull mode, +0.5 % capsule mode over a real call)
#endif
template <typename T>
__device__ RPK_BOXBOX_INLINE int box_box(RawCon<T>* out, RawCon<T>* extra, const T* p1, const T* m1, const T* s1, const T* p2,
                                    const T* m2, const T* s2) {
  using N = Num<T>;
  T R[3][3], Q[3][3], t[3], tb[3];
  const T d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
#pragma unroll
  for (int i = 0; i < 3; i++) {
    t[i] = m1[i] * d[0] + m1[3 + i] * d[1] + m1[6 + i] * d[2];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      R[i][j] = m1[i] * m2[j] + m1[3 + i] * m2[3 + j] + m1[6 + i] * m2[6 + j];
      Q[i][j] = N::abs(R[i][j]) + (T)1e-12;
    }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) tb[j] = t[0] * R[0][j] + t[1] * R[1][j] + t[2] * R[2][j];
  T best = (T)-1e30, sgn = 1;
  int code = -1;
  bool separated = false;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const T sep = N::abs(t[i]) - (s1[i] + s2[0] * Q[i][0] + s2[1] * Q[i][1] + s2[2] * Q[i][2]);
    separated = separated || sep > (T)0;
    if (sep > best) { best = sep; code = i; sgn = t[i] >= 0 ? (T)1 : (T)-1; }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const T sep = N::abs(tb[j]) - (s2[j] + s1[0] * Q[0][j] + s1[1] * Q[1][j] + s1[2] * Q[2][j]);
    separated = separated || sep > (T)0;
    // (parallel faces tie exactly: geom1's face stays the reference unless geom2's is clearly better)
    if (sep > best + (T)1e-10) { best = sep; code = 3 + j; sgn = tb[j] >= 0 ? (T)1 : (T)-1; }
  }
  T ebest = (T)-1e30, esgn = 1, einv = 0;
  int ecode = -1;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      constexpr int dummy = 0; (void)dummy;
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const T l2 = R[i1][j] * R[i1][j] + R[i2][j] * R[i2][j];
      const bool ok = l2 >= (T)1e-12;            // parallel edges are covered by the face axes
      const T inv = (T)1 / N::sqrt(ok ? l2 : (T)1);
      const T tl = (t[i2] * R[i1][j] - t[i1] * R[i2][j]) * inv;
      const T ra = (s1[i1] * Q[i2][j] + s1[i2] * Q[i1][j]) * inv;
      const T rb = (s2[j1] * Q[i][j2] + s2[j2] * Q[i][j1]) * inv;
      const T sep = N::abs(tl) - ra - rb;
      separated = separated || (ok && sep > (T)0);
      if (ok && sep > ebest) { ebest = sep; ecode = 3 * i + j; esgn = tl >= 0 ? (T)1 : (T)-1; einv = inv; }
    }
  }
  if (separated) return 0;
  int n = 0;
  // a face axis wins unless the edge axis is clearly better (5 % bias, as is customary)
  if (ecode >= 0 && ebest > best + (T)1e-9 + (T)0.05 * N::abs(best)) {
    const int i = ecode / 3, j = ecode - 3 * i;
    const T ua[3] = {pick3(m1[0], m1[1], m1[2], i), pick3(m1[3], m1[4], m1[5], i), pick3(m1[6], m1[7], m1[8], i)};
    const T ub[3] = {pick3(m2[0], m2[1], m2[2], j), pick3(m2[3], m2[4], m2[5], j), pick3(m2[6], m2[7], m2[8], j)};
    T nrm[3];
    cross3(nrm, ua, ub);
#pragma unroll
    for (int k = 0; k < 3; k++) nrm[k] *= esgn * einv;    // from geom1 to geom2
    T pa[3] = {p1[0], p1[1], p1[2]}, pb[3] = {p2[0], p2[1], p2[2]};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const T ak[3] = {m1[k], m1[3 + k], m1[6 + k]}, bk[3] = {m2[k], m2[3 + k], m2[6 + k]};
      const T sa = k != i ? (dot3(nrm, ak) >= 0 ? s1[k] : -s1[k]) : (T)0;
      const T sb = k != j ? (dot3(nrm, bk) >= 0 ? -s2[k] : s2[k]) : (T)0;
#pragma unroll
      for (int c = 0; c < 3; c++) { pa[c] += sa * ak[c]; pb[c] += sb * bk[c]; }
    }
    const T w[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    const T uaub = dot3(ua, ub), q1 = dot3(ua, w), q2 = -dot3(ub, w), den = (T)1 - uaub * uaub;
    T alpha = 0, beta = 0;
    if (den > (T)1e-12) { alpha = (q1 + uaub * q2) / den; beta = (uaub * q1 + q2) / den; }
    const T la = pick3(s1[0], s1[1], s1[2], i), lb = pick3(s2[0], s2[1], s2[2], j);
    alpha = fmin(la, fmax(-la, alpha)); beta = fmin(lb, fmax(-lb, beta));
    T pos[3];
#pragma unroll
    for (int c = 0; c < 3; c++) pos[c] = (T)0.5 * ((pa[c] + alpha * ua[c]) + (pb[c] + beta * ub[c]));
    out[0].dist = ebest;
#pragma unroll
    for (int c = 0; c < 3; c++) { out[0].pos[c] = pos[c]; out[0].n[c] = nrm[c]; }
    return 1;
  }
  // ---- face contact.  Reference frame (zr = fs * face axis towards the other box, ur, vr) and the
  // incident box (centre ci, scaled axes) are built with selects, never with dynamic indices.
  const bool refA = code < 3;
  const int ax = refA ? code : code - 3;
  const T* pr = refA ? p1 : p2; const T* mr = refA ? m1 : m2; const T* sr = refA ? s1 : s2;
  const T* pi = refA ? p2 : p1; const T* mi = refA ? m2 : m1; const T* si = refA ? s2 : s1;
  const T fs = refA ? sgn : -sgn;
  const int au = (ax + 1) % 3, av = (ax + 2) % 3;
  T zr[3], ur[3], vr[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    zr[c] = fs * pick3(mr[3 * c], mr[3 * c + 1], mr[3 * c + 2], ax);
    ur[c] = pick3(mr[3 * c], mr[3 * c + 1], mr[3 * c + 2], au);
    vr[c] = pick3(mr[3 * c], mr[3 * c + 1], mr[3 * c + 2], av);
  }
  const T h = pick3(sr[0], sr[1], sr[2], ax), hu = pick3(sr[0], sr[1], sr[2], au), hv = pick3(sr[0], sr[1], sr[2], av);
  const T dd[3] = {pi[0] - pr[0], pi[1] - pr[1], pi[2] - pr[2]};
  const T ci[3] = {dot3(ur, dd), dot3(vr, dd), dot3(zr, dd)};   // (u, v, z) of the incident centre
  // incident box axes in (u, v, z), scaled by the half sizes
  T E[3][3];
  T zabs[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const T ek[3] = {mi[k], mi[3 + k], mi[6 + k]};
    E[k][0] = si[k] * dot3(ur, ek); E[k][1] = si[k] * dot3(vr, ek); E[k][2] = si[k] * dot3(zr, ek);
    zabs[k] = N::abs(dot3(zr, ek));
  }
  // incident face: the face most anti-parallel to zr (first maximum)
  const int ia = (zabs[0] >= zabs[1] && zabs[0] >= zabs[2]) ? 0 : (zabs[1] >= zabs[2] ? 1 : 2);
  const int iu = (ia + 1) % 3, iv = (ia + 2) % 3;
  T Ea[3], Eu[3], Ev[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    Ea[c] = pick3(E[0][c], E[1][c], E[2][c], ia);
    Eu[c] = pick3(E[0][c], E[1][c], E[2][c], iu);
    Ev[c] = pick3(E[0][c], E[1][c], E[2][c], iv);
  }
  const T isg = Ea[2] > 0 ? (T)-1 : (T)1;       // the face whose outward normal opposes zr
  const T fc[3] = {ci[0] + isg * Ea[0], ci[1] + isg * Ea[1], ci[2] + isg * Ea[2]};
  T nrm[3];
#pragma unroll
  for (int k = 0; k < 3; k++) nrm[k] = refA ? zr[k] : -zr[k];
  // every candidate that penetrates is a contact.  Candidates are collected as (depth, u, v): the first three in
  // registers (static indices), the rest in the memory-resident `extra` (its first three words per record); the
  // world-frame points are formed after the last candidate
  T bd[3] = {(T)-1, (T)-1, (T)-1}, bu[3] = {0, 0, 0}, bv[3] = {0, 0, 0};
  auto emit = [&](T cu, T cv, T zz) {
    const T depth = h - zz;
    if (depth >= (T)0 && n < RPK_BOXBOX_MAX) {
      if (n < 3) {
#pragma unroll
        for (int i = 0; i < 3; i++) { const bool me = n == i; bd[i] = me ? depth : bd[i]; bu[i] = me ? cu : bu[i]; bv[i] = me ? cv : bv[i]; }
      } else if (RPK_BOXBOX_MAX > 3) {
        extra[n - 3].dist = depth; extra[n - 3].pos[0] = cu; extra[n - 3].pos[1] = cv;
      }
      n++;
    }
  };
  T q[4][3];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const T su = (c == 0 || c == 3) ? (T)-1 : (T)1, sv = c < 2 ? (T)-1 : (T)1;
#pragma unroll
    for (int a = 0; a < 3; a++) q[c][a] = fc[a] + su * Eu[a] + sv * Ev[a];
  }
  // (a) incident vertices inside the reference face's prism
#pragma unroll
  for (int c = 0; c < 4; c++)
    if (N::abs(q[c][0]) <= hu && N::abs(q[c][1]) <= hv) emit(q[c][0], q[c][1], q[c][2]);
  // (b) reference corners under the incident face (a parallelogram in (u, v))
  const T det = Eu[0] * Ev[1] - Eu[1] * Ev[0];
  if (N::abs(det) > (T)1e-14) {
    const T idet = (T)1 / det;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const T cu = (c == 0 || c == 3) ? -hu : hu, cv = c < 2 ? -hv : hv;
      const T ru = cu - fc[0], rv = cv - fc[1];
      const T a = (ru * Ev[1] - rv * Ev[0]) * idet, b = (Eu[0] * rv - Eu[1] * ru) * idet;
      if (N::abs(a) <= (T)1 && N::abs(b) <= (T)1) emit(cu, cv, fc[2] + a * Eu[2] + b * Ev[2]);
    }
  }
  // (c) crossings of the incident face's edges with the reference rectangle's edges (one
  // reciprocal per edge and axis)
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const T* qa = q[c];
    const T* qb = q[(c + 1) % 4];
#pragma unroll
    for (int cax = 0; cax < 2; cax++) {
      const int oax = 1 - cax;
      const T hh = cax == 0 ? hu : hv, ho = cax == 0 ? hv : hu;
      const T dq = qb[cax] - qa[cax];
      const T idq = (T)1 / (dq != (T)0 ? dq : (T)1);
#pragma unroll
      for (int sd = 0; sd < 2; sd++) {
        const T lim = sd ? hh : -hh;
        const T da = qa[cax] - lim, db = qb[cax] - lim;
        if ((da < 0) != (db < 0) && dq != (T)0) {
          const T tt = -da * idq;
          const T oc = qa[oax] + tt * (qb[oax] - qa[oax]);
          if (N::abs(oc) <= ho) {
            const T zz = qa[2] + tt * (qb[2] - qa[2]);
            if (cax == 0) emit(lim, oc, zz); else emit(oc, lim, zz);
          }
        }
      }
    }
  }
  // position midway between the surfaces
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const T zc = h - (T)0.5 * bd[i];
#pragma unroll
    for (int k = 0; k < 3; k++) { out[i].pos[k] = pr[k] + ur[k] * bu[i] + vr[k] * bv[i] + zr[k] * zc; out[i].n[k] = nrm[k]; }
    out[i].dist = -bd[i];
  }
  if (RPK_BOXBOX_MAX > 3 && n > 3) {   // (two faces resting on each other: rare)
    for (int i = 3; i < RPK_BOXBOX_MAX; i++) {
      if (i < n) {
        const T depth = extra[i - 3].dist, cu = extra[i - 3].pos[0], cv = extra[i - 3].pos[1];
        const T zc = h - (T)0.5 * depth;
        RawCon<T> c;
#pragma unroll
        for (int k = 0; k < 3; k++) { c.pos[k] = pr[k] + ur[k] * cu + vr[k] * cv + zr[k] * zc; c.n[k] = nrm[k]; }
        c.dist = -depth;
        extra[i - 3] = c;
      }
    }
  }
  return n;
}
// ---------------------------------------------------------------------------------------------
// Convex pairs with a hull (GEOM_MESH_) [MJ: mjc_Convex -> libccd MPR; restated, see the oracle]:
// Minkowski Portal Refinement on B - A, one contact: depth along the final portal's normal,
// normal from A to B, position = witness midpoints blended with the origin ray's barycentric
// weights.  Tolerance 1e-6, at most 50 refinement steps.
//
// The whole wave walks the procedure together (wave-uniform control flow, one support evaluation per
// trip for every lane that still has a pair in flight; a lane's own sequence of supports and decisions
// is the sequential procedure's).  That is what makes the hull support cheap: the lanes that use the
// same vertex set scan it together through the SCALAR cache (uniform addresses), instead of every lane
// pulling 26 vertices through its own vector loads (7 dependent round trips per support: the narrow
// phase was 38 % of all cycles of an mj_step in hull-fingertip mode).
template <typename T> struct CGeom { int type, nvert, vadr, flip, graph; T pos[3], mat[9], size[3]; };   // flip: bit k = the stored vertex set is this hull's mirror image in coordinate k; graph: the set has a vertex graph (walked, not scanned)
template <typename T> struct MPoint { T v[3], m[3]; };   // a point of B - A and the midpoint of its two witness points

// support point of a capsule / box (hulls: hull_support_wave); CYL: ... or a cylinder [MJ: mjc_support, mjGEOM_CYLINDER:
// in the geom frame the direction's xy part scaled to the radius (nothing when it vanishes), sign(z) * half height]
// (size = (radius, radius, half height): the cylinder's bounding box, which the fp32 culls use as they use a hull's)
template <typename T, bool CYL = false>
__device__ __forceinline__ void prim_support(const CGeom<T>& g, const T* d, T* out) {
  if (CYL && g.type == GEOM_CYL_) {
    T dl[3], r[3], w[3];
    matT_vec(dl, g.mat, d);
    const T t = Num<T>::sqrt(dl[0] * dl[0] + dl[1] * dl[1]);
    const T ts = t > RPK_MINVAL ? t : (T)1;   // (divisions, as the oracle writes them: the same roundings)
    r[0] = t > RPK_MINVAL ? dl[0] / ts * g.size[0] : (T)0;
    r[1] = t > RPK_MINVAL ? dl[1] / ts * g.size[0] : (T)0;
    r[2] = dl[2] > 0 ? g.size[2] : (dl[2] < 0 ? -g.size[2] : (T)0);
    mat_vec(w, g.mat, r);
    out[0] = g.pos[0] + w[0]; out[1] = g.pos[1] + w[1]; out[2] = g.pos[2] + w[2];
  } else if (g.type == GEOM_CAPSULE_) {
    const T ax[3] = {g.mat[2], g.mat[5], g.mat[8]};
    const T sl = dot3(ax, d) >= 0 ? g.size[1] : -g.size[1];
#pragma unroll
    for (int k = 0; k < 3; k++) out[k] = g.pos[k] + sl * ax[k] + g.size[0] * d[k];
  } else {
    T o[3] = {g.pos[0], g.pos[1], g.pos[2]};
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const T ax[3] = {g.mat[a], g.mat[3 + a], g.mat[6 + a]};
      const T sh = dot3(ax, d) >= 0 ? g.size[a] : -g.size[a];
#pragma unroll
      for (int k = 0; k < 3; k++) o[k] += sh * ax[k];
    }
    out[0] = o[0]; out[1] = o[1]; out[2] = o[2];
  }
}
// Support point of hull g in direction d for every lane with `need` (called by the whole wave).  First
// maximum wins, as in the sequential scan.  mv = the model's hull vertex table (uniform pointer).
// Hulls with a vertex graph (more than 32 vertices; hv / hg = RpModel::hull_vert / hull_graph): every such lane walks
// ITS hull from vertex 0 to the neighbour with the largest dot product while that is strictly larger -- the oracle's
// walk (geom_support), vertex for vertex.  Per-lane loads; the wave loops until its last walker has arrived.
// (GRAPH = false: the builds for scenes without such hulls -- the benchmark's -- carry none of this: with the walk
// compiled in, the position stage of the stand-in scene lost 4 %)
// (LDSV: `mv` is a copy of the vertex table in LDS -- the pooled narrow phase, whose waves own no other LDS: the scan
// reads it with uniform addresses (broadcast reads) and the winner's coordinates come back in ~130 cycles instead of
// a dependent global round trip per support; the one-kernel stage's LDS is full and keeps the scalar-cache scan)
template <typename T, bool GRAPH, bool LDSV = false>
__device__ __forceinline__ void hull_support_wave(const T* mv, const T* hv, const int* hg, const CGeom<T>& g, const T* d, const bool need_any, T* out) {
  T dl[3];
  matT_vec(dl, g.mat, d);
#pragma unroll
  for (int k = 0; k < 3; k++) if ((g.flip >> k) & 1) dl[k] = -dl[k];
  T bv = (T)-1e30;
  int bi = 0;
  const bool walk = GRAPH && need_any && g.graph != 0;
  const bool need = need_any && !walk;
  if (GRAPH && __ballot(walk) != 0ull) {
    bool going = walk;
    if (walk) { const T* v0 = hv + 3 * (size_t)g.vadr; bv = dl[0] * v0[0] + dl[1] * v0[1] + dl[2] * v0[2]; }
    while (__ballot(going) != 0ull) {
      if (going) {
        const int* row = hg + (size_t)RPK_HULL_GRAPH_ROW * (g.vadr + bi);
        const int deg = row[0];
        int next = -1;
        T nv_ = bv;
        for (int j = 0; j < deg; j++) {
          const int nb = row[1 + j];
          const T* vn = hv + 3 * (size_t)(g.vadr + nb);
          const T v = dl[0] * vn[0] + dl[1] * vn[1] + dl[2] * vn[2];
          if (v > nv_) { nv_ = v; next = nb; }
        }
        if (next < 0) going = false;
        else { bi = next; bv = nv_; }
      }
    }
  }
  unsigned long long todo = __ballot(need);
  while (todo) {   // one trip per distinct vertex set among the lanes that need a support
    const int L0 = __ffsll((long long)todo) - 1;
    const int base = bcast(g.vadr, L0), nv = bcast(g.nvert, L0);
    const bool mine = need && g.vadr == base;
    todo &= ~__ballot(mine);
    // (eight vertices per trip, so that their scalar loads are in flight together; the table pads every set to a
    // multiple of eight with copies of its last vertex, which cannot win a strict comparison against itself)
    if constexpr (LDSV) {
      const T* vb = mv + 3 * base;
      for (int i = 0; i < nv; i += 8) {
        T x[8], y[8], z[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { x[u] = vb[3 * (i + u)]; y[u] = vb[3 * (i + u) + 1]; z[u] = vb[3 * (i + u) + 2]; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const T v = dl[0] * x[u] + dl[1] * y[u] + dl[2] * z[u];
          if (mine && v > bv) { bv = v; bi = i + u; }
        }
      }
    } else {
    const T RPK_CONST_AS* vb = uniform_const(mv + 3 * base);
#ifdef RPK_X_MPR_SHORTSCAN   // timing experiment only (wrong results): what the vertex scan costs
    for (int i = 0; i < 8; i += 8) {
#else
    for (int i = 0; i < nv; i += 8) {
#endif
      T x[8], y[8], z[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { x[u] = vb[3 * (i + u)]; y[u] = vb[3 * (i + u) + 1]; z[u] = vb[3 * (i + u) + 2]; }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const T v = dl[0] * x[u] + dl[1] * y[u] + dl[2] * z[u];
        if (mine && v > bv) { bv = v; bi = i + u; }
      }
    }
    }
  }
  const int a = need_any ? 3 * (g.vadr + bi) : 0;
  const T* src = (GRAPH && walk) ? hv : mv;
  T bl[3] = {src[a], src[a + 1], src[a + 2]};
#pragma unroll
  for (int k = 0; k < 3; k++) if ((g.flip >> k) & 1) bl[k] = -bl[k];
  T w[3];
  mat_vec(w, g.mat, bl);
  out[0] = g.pos[0] + w[0]; out[1] = g.pos[1] + w[1]; out[2] = g.pos[2] + w[2];
}
template <typename T> __device__ __forceinline__ bool normalize3(T* v) {
  const T n = Num<T>::sqrt(dot3(v, v));
  if (n < (T)1e-14) return false;
  const T inv = (T)1 / n;
  v[0] *= inv; v[1] *= inv; v[2] *= inv;
  return true;
}
template <typename T> __device__ __forceinline__ void portal_dir(T* dir, const MPoint<T>& a, const MPoint<T>& b, const MPoint<T>& o) {
  const T t1[3] = {a.v[0] - o.v[0], a.v[1] - o.v[1], a.v[2] - o.v[2]};
  const T t2[3] = {b.v[0] - o.v[0], b.v[1] - o.v[1], b.v[2] - o.v[2]};
  cross3(dir, t1, t2);
}

// Called by the whole wave; lanes with `active` hold a pair (A, B).  Returns the number of contacts (0 / 1).
#ifndef RPK_MPR_INLINE
#define RPK_MPR_INLINE __forceinline__   // (inlined: as a real call it cost the position stage 290 more scratch operations, 19 of them in the drain loop)
#endif
template <typename T, bool GRAPH, bool LDSV = false>
__device__ RPK_MPR_INLINE int convex_mpr_wave(RawCon<T>* __restrict__ out, const CGeom<T>* __restrict__ Ap,
                                            const CGeom<T>* __restrict__ Bp, const T* __restrict__ mv, const T* __restrict__ hv,
                                            const int* __restrict__ hg, const bool active, const T tol, const T tol_poly) {
  // (the two geoms by value: re-reading them through the pointers on every trip -- scratch memory, a
  // dependent round trip each -- was most of this routine's time)
  const CGeom<T> A = *Ap, B = *Bp;
  MPoint<T> v0, v1, v2, v3, S;
  RawCon<T> res;
  res.dist = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) { res.pos[k] = 0; res.n[k] = 0; }
  T dir[3], t1[3];
  int phase = active ? 0 : 4;   // 0: first support, 1: second, 2: portal discovery, 3: refinement, 4: done
  int it = 0, result = 0;
  bool hit = false;
#pragma unroll
  for (int k = 0; k < 3; k++) { v0.m[k] = (T)0.5 * (A.pos[k] + B.pos[k]); v0.v[k] = B.pos[k] - A.pos[k]; }
  if (Num<T>::sqrt(dot3(v0.v, v0.v)) < (T)1e-10) v0.v[0] = (T)1e-5;
  dir[0] = -v0.v[0]; dir[1] = -v0.v[1]; dir[2] = -v0.v[2];
  normalize3(dir);
  v1 = v0; v2 = v0; v3 = v0;
#ifdef RP_MPR_COUNT   // emulator-only diagnostics: trips per call, lanes in flight
  int trips_ = 0; const int lanes_ = __popcll(__ballot(active));
#endif
  while (__ballot(phase != 4) != 0ull) {
#ifdef RP_MPR_COUNT
    trips_++;
#endif
    const bool run = phase != 4;
    // ---- S = support of B - A along dir
    {
      const T nd[3] = {-dir[0], -dir[1], -dir[2]};
      T p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0};
      const bool hA = run && A.type == GEOM_MESH_, hB = run && B.type == GEOM_MESH_;
      if (__ballot(hA) != 0ull) hull_support_wave<T, GRAPH, LDSV>(mv, hv, hg, A, nd, hA, p1);
      if (__ballot(hB) != 0ull) hull_support_wave<T, GRAPH, LDSV>(mv, hv, hg, B, dir, hB, p2);
      if (run && A.type != GEOM_MESH_) prim_support<T, GRAPH>(A, nd, p1);
      if (run && B.type != GEOM_MESH_) prim_support<T, GRAPH>(B, dir, p2);
#pragma unroll
      for (int k = 0; k < 3; k++) { S.v[k] = p2[k] - p1[k]; S.m[k] = (T)0.5 * (p1[k] + p2[k]); }
    }
    // Every branch below that needs a new search direction only FORMS it (unnormalised) and names what has to happen
    // once it is a unit vector (`post`); the normalisation -- an fp64 square root and a division, ~400 cycles of
    // dependent chain -- then runs ONCE per trip for the whole wave instead of once per branch (a pooled chunk holds
    // lanes in every phase, so every branch's code is walked on every trip: up to six normalisations).  A lane's own
    // arithmetic is what it was, operation for operation.
    int post = 0;
    if (phase == 0) {
      v1 = S;
      if (dot3(v1.v, dir) <= 0) phase = 4;
      else { cross3(dir, v0.v, v1.v); post = 1; }
    } else if (phase == 1) {
      v2 = S;
      if (dot3(v2.v, dir) <= 0) phase = 4;
      else { portal_dir(dir, v1, v2, v0); post = 2; }
    } else if (phase == 2) {
      // ---- portal discovery
      v3 = S;
      if (dot3(v3.v, dir) <= 0) phase = 4;
      else {
        cross3(t1, v1.v, v3.v);
        if (dot3(t1, v0.v) < 0) { v2 = v3; portal_dir(dir, v1, v3, v0); post = 3; }
        else {
          cross3(t1, v3.v, v2.v);
          if (dot3(t1, v0.v) < 0) { v1 = v3; portal_dir(dir, v3, v2, v0); post = 3; }
          else { it = 0; hit = false; portal_dir(dir, v2, v3, v1); post = 4; }   // (refinement starts: direction of the first portal)
        }
      }
    } else if (phase == 3) {
      const T reach = dot3(S.v, dir) - dot3(v1.v, dir);
      if (!hit && dot3(S.v, dir) < 0) phase = 4;
      // (MuJoCo's uniform rule by default: tol = tol_poly = 1e-6.  rp_set_mpr_tolerance gives POLYTOPE pairs -- box / hull on
      // both sides -- their own: refined to 1e-10 the result does not depend on which of several tied support vertices
      // rounding put first; with the 1e-6 rule the engine stops 1.7e-7 short of the oracle at one mj_step in 1580 of the
      // hull replay, 1.6 % of that step's velocity change.  oracle/rp_oracle.c: g_mpr_tol_poly.)
      else if (reach <= (((A.type == GEOM_BOX_ || A.type == GEOM_MESH_) && (B.type == GEOM_BOX_ || B.type == GEOM_MESH_)) ? tol_poly : tol) || it == 50) {
        if (hit) {
          const T depth = dot3(v1.v, dir);
          T b[4], c[3];
          cross3(c, v1.v, v2.v); b[0] = dot3(c, v3.v);
          cross3(c, v3.v, v2.v); b[1] = dot3(c, v0.v);
          cross3(c, v0.v, v1.v); b[2] = dot3(c, v3.v);
          cross3(c, v2.v, v1.v); b[3] = dot3(c, v0.v);
          T sum = b[0] + b[1] + b[2] + b[3];
          if (sum <= 0) {
            b[0] = 0;
            cross3(c, v2.v, v3.v); b[1] = dot3(c, dir);
            cross3(c, v3.v, v1.v); b[2] = dot3(c, dir);
            cross3(c, v1.v, v2.v); b[3] = dot3(c, dir);
            sum = b[1] + b[2] + b[3];
          }
          const T inv = (T)1 / sum;
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const T acc = b[0] * v0.m[k] + b[1] * v1.m[k] + b[2] * v2.m[k] + b[3] * v3.m[k];
            res.pos[k] = acc * inv;
            res.n[k] = -dir[k];   // (the portal faces away from the centre of B - A: A -> B is -dir)
          }
          res.dist = -depth;
          result = 1;
        }
        phase = 4;
      } else {
        cross3(t1, S.v, v0.v);
        if (dot3(v1.v, t1) > 0) { if (dot3(v2.v, t1) > 0) v1 = S; else v3 = S; }
        else { if (dot3(v3.v, t1) > 0) v2 = S; else v1 = S; }
        it++;
        portal_dir(dir, v2, v3, v1);
        post = 5;
      }
    }
    // ---- the one normalisation of the trip, then what each branch had to do with its unit direction
    bool ok = true;
    if (post != 0) ok = normalize3(dir);
    if (post == 1) {
      if (!ok) {   // the origin lies on the ray v0 -> v1
        T n[3] = {v1.v[0] - v0.v[0], v1.v[1] - v0.v[1], v1.v[2] - v0.v[2]};
        normalize3(n);
        res.dist = -dot3(v1.v, n);
#pragma unroll
        for (int k = 0; k < 3; k++) { res.n[k] = -n[k]; res.pos[k] = v1.m[k]; }
        result = res.dist <= 0 ? 1 : 0;
        phase = 4;
      } else phase = 1;
    } else if (post == 2) {
      if (dot3(dir, v0.v) > 0) {
        const MPoint<T> tmp = v1; v1 = v2; v2 = tmp;
        dir[0] = -dir[0]; dir[1] = -dir[1]; dir[2] = -dir[2];
      }
      phase = 2; it = 0;
    } else if (post == 3) {
      if (++it > 50) phase = 4;
    } else if (post == 4) {
      if (!ok) phase = 4;
      else { if (dot3(dir, v1.v) >= 0) hit = true; phase = 3; }
    } else if (post == 5) {
      if (!ok) phase = 4;
      else if (dot3(dir, v1.v) >= 0) hit = true;
    }
  }
  *out = res;
#ifdef RP_MPR_COUNT
  { const int hits_ = __popcll(__ballot(result != 0)); if ((threadIdx.x & 63) == 0) printf("MPR call: %d pairs, %d trips, %d contacts\n", lanes_, trips_, hits_); }
#endif
  return result;
}
}  // namespace rpk
