// rp_dense.hpp — dense SPD factor + solve of the cross-coupled block of the Newton
// Hessian, one row per lane, matrix packed in LDS.
#pragma once
#include "rp_wave.hpp"

namespace rpk {
// 1/sqrt(x) to working precision: hardware estimate + Newton steps (no division on the
// pivot chain of the dense factorisation).
template <typename T> __device__ __forceinline__ T rsqrt_nr(T x);
template <> __device__ __forceinline__ double rsqrt_nr<double>(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = y * __builtin_fma(-h * y, y, 1.5);
  y = y * __builtin_fma(-h * y, y, 1.5);
  return y;
}
template <> __device__ __forceinline__ float rsqrt_nr<float>(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  const float h = 0.5f * x;
  y = y * __builtin_fmaf(-h * y, y, 1.5f);
  return y;
}

// 1/x to working precision for the pivots of the tree elimination (x > 0, normal range): hardware
// estimate + Newton steps instead of the IEEE division sequence (v_div_scale / v_div_fmas / v_div_fixup:
// twice the dependent chain, and the pivot chain is what the chain leaders wait on).
template <typename T> __device__ __forceinline__ T rcp_nr(T x);
template <> __device__ __forceinline__ double rcp_nr<double>(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  return y;
}
template <> __device__ __forceinline__ float rcp_nr<float>(float x) {
  float y = __builtin_amdgcn_rcpf(x);
  const float e = __builtin_fmaf(-x, y, 1.0f);
  return __builtin_fmaf(y, e, y);
}

// Dense solve of the packed lower-triangular SPD system H (n rows, n uniform) with the
// right-hand side stored as row n of H; returns x_i in lane i < n.
//   * left-looking L L^T, lane = row, four columns per step (one LDS hand-over per four
//     pivots; a pair and a single column finish the remainder);
//   * inner products read the lane's own row and rows j, j+1 (uniform address =
//     LDS broadcast) with paired 64-bit reads, no v_readlane in the loop;
//   * the rhs row takes part in the factorisation like any other row, which performs
//     the forward substitution for free; only the backward pass is a serial chain.
// Measured on gfx950, one wave per SIMD, n = 20, fp64: 12 k cycles (the first version,
// one column per step with sqrt/divide/readlane, took 34 k).
#ifndef RPK_DENSE_PTRIP
#define RPK_DENSE_PTRIP 4
#endif
template <typename T, bool WIDE = (sizeof(T) == 4)>
__device__ T dense_factor_solve(T* H, int n, int lane, int* warn) {
  T invd_me = 0;
  int j = 0;
  // four columns per step while they last (one LDS hand-over per four pivots) ...
  // (fp32 only: the fp64 solver kernel is at its register limit and the four-column step
  // pushes it into scratch spills, measured -7 %)
  for (; WIDE && j + 4 <= n; j += 4) {
    const bool act = lane >= j && lane <= n;
    const T* ri = H + tri(act ? lane : 0, 0);
    const T* r0 = H + tri(j, 0);
    const T* r1 = H + tri(j + 1, 0);
    const T* r2 = H + tri(j + 2, 0);
    const T* r3 = H + tri(j + 3, 0);
    T s0 = ri[j], s1 = ri[j + 1], s2 = ri[j + 2], s3 = ri[j + 3];
    // (fp64 -- the lean solver stage: four p per trip, all reads in flight before the first multiply-add waits; the fp32
    // build's solver kernel is over its register budget already and keeps two)
    constexpr int PT = sizeof(T) == 8 ? RPK_DENSE_PTRIP : 2;
    for (int p = 0; p < j; p += PT) {  // j is a multiple of 4 here; p per trip: LDS round trips against live registers
      T a[PT], b0[PT], b1[PT], b2[PT], b3[PT];
#pragma unroll
      for (int u = 0; u < PT; u++) { a[u] = ri[p + u]; b0[u] = r0[p + u]; b1[u] = r1[p + u]; b2[u] = r2[p + u]; b3[u] = r3[p + u]; }
      if constexpr (PT > 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < PT; u++) {
        s0 -= a[u] * b0[u]; s1 -= a[u] * b1[u]; s2 -= a[u] * b2[u]; s3 -= a[u] * b3[u];
      }
    }
    T d0 = bcast(s0, j);
    if (!(d0 >= RPK_MINVAL)) { d0 = RPK_MINVAL; *warn |= 4; }
    const T q0 = rsqrt_nr(d0);
    const T l0 = s0 * q0;
    s1 -= l0 * bcast(l0, j + 1);
    T d1 = bcast(s1, j + 1);
    if (!(d1 >= RPK_MINVAL)) { d1 = RPK_MINVAL; *warn |= 4; }
    const T q1 = rsqrt_nr(d1);
    const T l1 = s1 * q1;
    s2 -= l0 * bcast(l0, j + 2); s2 -= l1 * bcast(l1, j + 2);
    T d2 = bcast(s2, j + 2);
    if (!(d2 >= RPK_MINVAL)) { d2 = RPK_MINVAL; *warn |= 4; }
    const T q2 = rsqrt_nr(d2);
    const T l2 = s2 * q2;
    s3 -= l0 * bcast(l0, j + 3); s3 -= l1 * bcast(l1, j + 3); s3 -= l2 * bcast(l2, j + 3);
    T d3 = bcast(s3, j + 3);
    if (!(d3 >= RPK_MINVAL)) { d3 = RPK_MINVAL; *warn |= 4; }
    const T q3 = rsqrt_nr(d3);
    const T l3 = s3 * q3;
    if (lane == j) invd_me = q0;
    if (lane == j + 1) invd_me = q1;
    if (lane == j + 2) invd_me = q2;
    if (lane == j + 3) invd_me = q3;
    if (act) H[tri(lane, j)] = l0;
    if (act && lane > j) H[tri(lane, j + 1)] = l1;
    if (act && lane > j + 1) H[tri(lane, j + 2)] = l2;
    if (act && lane > j + 2) H[tri(lane, j + 3)] = l3;
    WSYNC();
  }
  // ... then a pair, then a single column
  for (; j + 2 <= n; j += 2) {
    const int j1 = j + 1;
    const bool act = lane >= j && lane <= n;
    const T* ri = H + tri(act ? lane : 0, 0);
    const T* rj = H + tri(j, 0);
    const T* rk = H + tri(j1, 0);
    T s = ri[j], t = ri[j1];
    int p = 0;
    for (; p + 4 <= j; p += 4) {
      T a0 = ri[p], a1 = ri[p + 1], a2 = ri[p + 2], a3 = ri[p + 3];
      T b0 = rj[p], b1 = rj[p + 1], b2 = rj[p + 2], b3 = rj[p + 3];
      T c0 = rk[p], c1 = rk[p + 1], c2 = rk[p + 2], c3 = rk[p + 3];
#if RPK_DENSE_PTRIP > 2
      __builtin_amdgcn_sched_barrier(0);
#endif
      s -= a0 * b0; t -= a0 * c0; s -= a1 * b1; t -= a1 * c1;
      s -= a2 * b2; t -= a2 * c2; s -= a3 * b3; t -= a3 * c3;
    }
    for (; p < j; p++) { T a0 = ri[p]; s -= a0 * rj[p]; t -= a0 * rk[p]; }
    T dj = bcast(s, j);
    if (!(dj >= RPK_MINVAL)) { dj = RPK_MINVAL; *warn |= 4; }
    const T rs = rsqrt_nr(dj);
    const T lij = s * rs;               // L[i][j] for lanes > j (lane j: sqrt(dj))
    const T lkj = bcast(lij, j1);
    t -= lij * lkj;
    T dk = bcast(t, j1);
    if (!(dk >= RPK_MINVAL)) { dk = RPK_MINVAL; *warn |= 4; }
    const T rs2 = rsqrt_nr(dk);
    const T lik = t * rs2;
    if (lane == j) invd_me = rs;
    if (lane == j1) invd_me = rs2;
    if (act) H[tri(lane, j)] = lij;
    if (act && lane > j) H[tri(lane, j1)] = lik;
    WSYNC();
  }
  if (j < n) {
    const bool act = lane >= j && lane <= n;
    const T* ri = H + tri(act ? lane : 0, 0);
    const T* rj = H + tri(j, 0);
    T s = ri[j];
    int p = 0;
#if RPK_DENSE_PTRIP > 2
    // (the reads of a trip in flight together: the plain loop waited for every pair of them, j / 2 LDS round trips)
    for (; p + 8 <= j; p += 8) {
      T a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { a[u] = ri[p + u]; b[u] = rj[p + u]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; u++) s -= a[u] * b[u];
    }
    for (; p + 4 <= j; p += 4) {
      T a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { a[u] = ri[p + u]; b[u] = rj[p + u]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; u++) s -= a[u] * b[u];
    }
#endif
    for (; p < j; p++) s -= ri[p] * rj[p];
    T dj = bcast(s, j);
    if (!(dj >= RPK_MINVAL)) { dj = RPK_MINVAL; *warn |= 4; }
    const T rs = rsqrt_nr(dj);
    if (lane == j) invd_me = rs;
    if (act) H[tri(lane, j)] = s * rs;
    WSYNC();
  }
  // row n now holds y = L^-1 b; backward pass L^T x = y
  T x = lane < n ? H[tri(n, 0) + lane] : (T)0;
  int p = n - 1;
  // (the rows of the NEXT group of four are requested before this group's serial chain starts: their LDS round trip runs
  // under it instead of in front of the next one)
  T nx0 = 0, nx1 = 0, nx2 = 0, nx3 = 0;
  if (p - 3 >= 0) {
    nx0 = H[tri(p, 0) + (lane < p ? lane : 0)]; nx1 = H[tri(p - 1, 0) + (lane < p - 1 ? lane : 0)];
    nx2 = H[tri(p - 2, 0) + (lane < p - 2 ? lane : 0)]; nx3 = H[tri(p - 3, 0) + (lane < p - 3 ? lane : 0)];
  }
  for (; p - 3 >= 0; p -= 4) {
    T l0 = nx0, l1 = nx1, l2 = nx2, l3 = nx3;
    if (p - 7 >= 0) {
      const int p4 = p - 4;
      nx0 = H[tri(p4, 0) + (lane < p4 ? lane : 0)]; nx1 = H[tri(p4 - 1, 0) + (lane < p4 - 1 ? lane : 0)];
      nx2 = H[tri(p4 - 2, 0) + (lane < p4 - 2 ? lane : 0)]; nx3 = H[tri(p4 - 3, 0) + (lane < p4 - 3 ? lane : 0)];
    }
    l0 = lane < p ? l0 : (T)0; l1 = lane < p - 1 ? l1 : (T)0; l2 = lane < p - 2 ? l2 : (T)0; l3 = lane < p - 3 ? l3 : (T)0;
    if (lane == p) x *= invd_me;
    x -= l0 * bcast(x, p);
    if (lane == p - 1) x *= invd_me;
    x -= l1 * bcast(x, p - 1);
    if (lane == p - 2) x *= invd_me;
    x -= l2 * bcast(x, p - 2);
    if (lane == p - 3) x *= invd_me;
    x -= l3 * bcast(x, p - 3);
  }
  for (; p >= 0; p--) {
    T l0 = H[tri(p, 0) + (lane < p ? lane : 0)];
    l0 = lane < p ? l0 : (T)0;
    if (lane == p) x *= invd_me;
    x -= l0 * bcast(x, p);
  }
  return x;
}
}  // namespace rpk
