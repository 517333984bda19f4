// experimental dense-factor variants for the microbenchmark (factor only; timing)
template <typename T>
__device__ T chol_solve_v6(T* H, int n, int lane, T x, int* warn) {
  T invd_me = 0;
  int j = 0;
  for (; j + 2 <= n; j += 2) {
    const int j1 = j + 1;
    const bool act = lane >= j && lane < n;
    const T* ri = H + tri(act ? lane : 0, 0);
    const T* rj = H + tri(j, 0);
    const T* rk = H + tri(j1, 0);
    T s = ri[j], t = ri[j1];
    int p = 0;
    for (; p + 4 <= j; p += 4) {
      T a0 = ri[p], a1 = ri[p + 1], a2 = ri[p + 2], a3 = ri[p + 3];
      T b0 = rj[p], b1 = rj[p + 1], b2 = rj[p + 2], b3 = rj[p + 3];
      T c0 = rk[p], c1 = rk[p + 1], c2 = rk[p + 2], c3 = rk[p + 3];
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
    const bool act = lane >= j && lane < n;
    const T* ri = H + tri(act ? lane : 0, 0);
    const T* rj = H + tri(j, 0);
    T s = ri[j];
    for (int p = 0; p < j; p++) s -= ri[p] * rj[p];
    T dj = bcast(s, j);
    if (!(dj >= RPK_MINVAL)) { dj = RPK_MINVAL; *warn |= 4; }
    const T rs = rsqrt_nr(dj);
    if (lane == j) invd_me = rs;
    if (act) H[tri(lane, j)] = s * rs;
    WSYNC();
  }
  if (n > 0) return invd_me;  // factor only
  // solve L L^T x = b
  const bool in = lane < n;
  if (!in) x = 0;
  const T* ri = H + tri(in ? lane : 0, 0);
  int p = 0;
  for (; p + 4 <= n; p += 4) {
    T l0 = ri[p], l1 = ri[p + 1], l2 = ri[p + 2], l3 = ri[p + 3];
    l0 = (in && lane > p) ? l0 : (T)0; l1 = (in && lane > p + 1) ? l1 : (T)0;
    l2 = (in && lane > p + 2) ? l2 : (T)0; l3 = (in && lane > p + 3) ? l3 : (T)0;
    if (lane == p) x *= invd_me;
    x -= l0 * bcast(x, p);
    if (lane == p + 1) x *= invd_me;
    x -= l1 * bcast(x, p + 1);
    if (lane == p + 2) x *= invd_me;
    x -= l2 * bcast(x, p + 2);
    if (lane == p + 3) x *= invd_me;
    x -= l3 * bcast(x, p + 3);
  }
  for (; p < n; p++) { T l0 = ri[p]; l0 = (in && lane > p) ? l0 : (T)0; if (lane == p) x *= invd_me; x -= l0 * bcast(x, p); }
  p = n - 1;
  for (; p - 3 >= 0; p -= 4) {
    T l0 = H[tri(p, 0) + (lane < p ? lane : 0)], l1 = H[tri(p - 1, 0) + (lane < p - 1 ? lane : 0)];
    T l2 = H[tri(p - 2, 0) + (lane < p - 2 ? lane : 0)], l3 = H[tri(p - 3, 0) + (lane < p - 3 ? lane : 0)];
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
  for (; p >= 0; p--) { T l0 = H[tri(p, 0) + (lane < p ? lane : 0)]; l0 = lane < p ? l0 : (T)0; if (lane == p) x *= invd_me; x -= l0 * bcast(x, p); }
  return x;
}
// V8: four columns per step
template <typename T>
__device__ T chol_solve_v8(T* H, int n, int lane, T x, int* warn) {
  T invd_me = 0;
  int j = 0;
  for (; j + 4 <= n; j += 4) {
    const bool act = lane >= j && lane < n;
    const T* ri = H + tri(act ? lane : 0, 0);
    const T* r0 = H + tri(j, 0); const T* r1 = H + tri(j + 1, 0); const T* r2 = H + tri(j + 2, 0); const T* r3 = H + tri(j + 3, 0);
    T s0 = ri[j], s1 = ri[j + 1], s2 = ri[j + 2], s3 = ri[j + 3];
    for (int p = 0; p < j; p += 4) {   // j is a multiple of 4
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const T a = ri[p + u];
        s0 -= a * r0[p + u]; s1 -= a * r1[p + u]; s2 -= a * r2[p + u]; s3 -= a * r3[p + u];
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
  for (; j < n; j++) {
    const bool act = lane >= j && lane < n;
    const T* ri = H + tri(act ? lane : 0, 0);
    const T* rj = H + tri(j, 0);
    T s = ri[j];
    for (int p = 0; p < j; p++) s -= ri[p] * rj[p];
    T dj = bcast(s, j);
    if (!(dj >= RPK_MINVAL)) { dj = RPK_MINVAL; *warn |= 4; }
    const T rs = rsqrt_nr(dj);
    if (lane == j) invd_me = rs;
    if (act) H[tri(lane, j)] = s * rs;
    WSYNC();
  }
  if (n > 0) return invd_me;  // factor only (timing)
  return x;
}

template <typename T, int V>
__device__ T chol_solve_variant(T* H, int n, int lane, T x, int* warn) {
  if constexpr (V == 6) return chol_solve_v6(H, n, lane, x, warn);
  else return chol_solve_v8(H, n, lane, x, warn);
}
