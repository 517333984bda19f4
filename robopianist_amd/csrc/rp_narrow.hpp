// rp_narrow.hpp — narrow-phase collision primitives (sphere/capsule/box), one candidate
// pair per lane [MJ: engine_collision_primitive.c / engine_collision_box.c, restated].
#pragma once
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

// capsule (geom1) vs box (geom2): closest axis point (exact root of the piecewise
// linear distance derivative) plus both segment ends.
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
  int n = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    T tc = i == 0 ? tstar : (i == 1 ? (T)-1 : (T)1);
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
}  // namespace rpk
