// experimental variants; V selects
template <typename T> __device__ __forceinline__ T rsqrt_nr(T x);
template <> __device__ __forceinline__ double rsqrt_nr<double>(double x) {
  double y = __builtin_amdgcn_rsq(x);
  // two Newton steps: y <- y * (1.5 - 0.5 x y^2)
  double h = 0.5 * x;
  y = y * __builtin_fma(-h * y, y, 1.5);
  y = y * __builtin_fma(-h * y, y, 1.5);
  return y;
}
template <> __device__ __forceinline__ float rsqrt_nr<float>(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  float h = 0.5f * x;
  y = y * __builtin_fmaf(-h * y, y, 1.5f);
  return y;
}
// LL^T, two columns per step, inner products from LDS (row i per lane, rows j, j+1 as
// uniform-address broadcast reads), pivots through rsqrt + Newton.
template <typename T>
__device__ T chol_solve_v4(T* H, int n, int lane, T x, int* warn) {
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
template <typename T>
__device__ T chol_solve_v7(T* H, int n, int lane, T x, int* warn) {
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
    const T rs = dj * (T)0.5;
    const T lij = s * rs;               // L[i][j] for lanes > j (lane j: sqrt(dj))
    const T lkj = bcast(lij, j1);
    t -= lij * lkj;
    T dk = bcast(t, j1);
    if (!(dk >= RPK_MINVAL)) { dk = RPK_MINVAL; *warn |= 4; }
    const T rs2 = dk * (T)0.5;
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
    const T rs = dj * (T)0.5;
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
template <typename T>
__device__ T chol_solve_v5(T* H, int n, int lane, T x, int* warn) {
  T invd_me = 0;
  int j = 0;
  for (; j + 2 <= n; j += 2) {
    const int j1 = j + 1;
    const bool act = lane >= j && lane < n;
    const T* ri = H + tri(act ? lane : 0, 0);
    const T* rj = H + tri(j, 0);
    const T* rk = H + tri(j1, 0);
    T s = ri[j], t = ri[j1];
    // (reads past column j stay inside H, which is padded; those terms are masked)
    for (int p = 0; p < j; p += 4) {
      T a0 = ri[p], a1 = ri[p + 1], a2 = ri[p + 2], a3 = ri[p + 3];
      T b0 = rj[p], b1 = rj[p + 1], b2 = rj[p + 2], b3 = rj[p + 3];
      T c0 = rk[p], c1 = rk[p + 1], c2 = rk[p + 2], c3 = rk[p + 3];
      if (p + 1 >= j) { a1 = 0; } if (p + 2 >= j) { a2 = 0; } if (p + 3 >= j) { a3 = 0; }
      s -= a0 * b0; t -= a0 * c0; s -= a1 * b1; t -= a1 * c1;
      s -= a2 * b2; t -= a2 * c2; s -= a3 * b3; t -= a3 * c3;
    }
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
    for (int p = 0; p < j; p += 4) {
      T a0 = ri[p], a1 = ri[p + 1], a2 = ri[p + 2], a3 = ri[p + 3];
      T b0 = rj[p], b1 = rj[p + 1], b2 = rj[p + 2], b3 = rj[p + 3];
      if (p + 1 >= j) { a1 = 0; } if (p + 2 >= j) { a2 = 0; } if (p + 3 >= j) { a3 = 0; }
      s -= a0 * b0; s -= a1 * b1; s -= a2 * b2; s -= a3 * b3;
    }
    T dj = bcast(s, j);
    if (!(dj >= RPK_MINVAL)) { dj = RPK_MINVAL; *warn |= 4; }
    const T rs = rsqrt_nr(dj);
    if (lane == j) invd_me = rs;
    if (act) H[tri(lane, j)] = s * rs;
    WSYNC();
  }
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
template <typename T, int V>
__device__ T chol_solve_variant(T* H, int n, int lane, T x, int* warn) {
  if constexpr (V == 4) { return chol_solve_v4(H, n, lane, x, warn); }
  else if constexpr (V == 6) { return chol_solve_v6(H, n, lane, x, warn); }
  else if constexpr (V == 7) { return chol_solve_v7(H, n, lane, x, warn); }
  else if constexpr (V == 5) { return chol_solve_v5(H, n, lane, x, warn); }
  else if constexpr (V == 1) {
    // LDL^T, left-looking, lane = row.  Stores L (unit diag implied) below diag, D on diag.
    // inner product uses D-scaled row j: w_p = L[j][p]*D[p]
    T dinv_me = 0;   // 1/D[lane] once known
    for (int j = 0; j < n; j++) {
      T rj = (lane < j) ? H[tri(j, 0) + lane] : (T)0;   // L[j][lane]
      T dl = (lane < j) ? H[tri(lane, lane)] : (T)0;    // D[lane]
      T wj = rj * dl;
      const bool act = lane >= j && lane < n;
      const T* ri = H + tri(act ? lane : 0, 0);
      T s = ri[j];
      int p = 0;
      for (; p + 8 <= j; p += 8) {
        T a0 = ri[p], a1 = ri[p + 1], a2 = ri[p + 2], a3 = ri[p + 3], a4 = ri[p + 4], a5 = ri[p + 5], a6 = ri[p + 6], a7 = ri[p + 7];
        s -= a0 * bcast(wj, p); s -= a1 * bcast(wj, p + 1); s -= a2 * bcast(wj, p + 2);
        s -= a3 * bcast(wj, p + 3); s -= a4 * bcast(wj, p + 4); s -= a5 * bcast(wj, p + 5);
        s -= a6 * bcast(wj, p + 6); s -= a7 * bcast(wj, p + 7);
      }
      for (; p < j; p++) { T a0 = ri[p]; s -= a0 * bcast(wj, p); }
      T dj = bcast(s, j);
      if (dj < RPK_MINVAL) { dj = RPK_MINVAL; *warn |= 4; }
      T inv = (T)1 / dj;
      if (lane == j) { H[tri(j, j)] = dj; dinv_me = inv; }
      else if (act) H[tri(lane, j)] = s * inv;
      WSYNC();
    }
    // solve L D L^T x = b
    const bool in = lane < n;
    if (!in) x = 0;
    const T* ri = H + tri(in ? lane : 0, 0);
    int p = 0;
    for (; p + 4 <= n; p += 4) {
      T l0 = ri[p], l1 = ri[p + 1], l2 = ri[p + 2], l3 = ri[p + 3];
      l0 = (in && lane > p) ? l0 : (T)0; l1 = (in && lane > p + 1) ? l1 : (T)0;
      l2 = (in && lane > p + 2) ? l2 : (T)0; l3 = (in && lane > p + 3) ? l3 : (T)0;
      x -= l0 * bcast(x, p); x -= l1 * bcast(x, p + 1); x -= l2 * bcast(x, p + 2); x -= l3 * bcast(x, p + 3);
    }
    for (; p < n; p++) { T l0 = ri[p]; l0 = (in && lane > p) ? l0 : (T)0; x -= l0 * bcast(x, p); }
    x *= dinv_me;
    p = n - 1;
    for (; p - 3 >= 0; p -= 4) {
      T l0 = H[tri(p, 0) + (lane < p ? lane : 0)], l1 = H[tri(p - 1, 0) + (lane < p - 1 ? lane : 0)];
      T l2 = H[tri(p - 2, 0) + (lane < p - 2 ? lane : 0)], l3 = H[tri(p - 3, 0) + (lane < p - 3 ? lane : 0)];
      l0 = lane < p ? l0 : (T)0; l1 = lane < p - 1 ? l1 : (T)0; l2 = lane < p - 2 ? l2 : (T)0; l3 = lane < p - 3 ? l3 : (T)0;
      x -= l0 * bcast(x, p); x -= l1 * bcast(x, p - 1); x -= l2 * bcast(x, p - 2); x -= l3 * bcast(x, p - 3);
    }
    for (; p >= 0; p--) { T l0 = H[tri(p, 0) + (lane < p ? lane : 0)]; l0 = lane < p ? l0 : (T)0; x -= l0 * bcast(x, p); }
    return x;
  } else if constexpr (V == 2) {
    // right-looking LDL^T in LDS with the trailing update spread over all 64 lanes is
    // not obviously better; instead: row-in-registers for n <= 16, static unroll.
    if (n > 16) { chol_packed(H, n, lane, warn); return solve_packed(H, n, lane, x); }
    T r[16];
    const bool in = lane < n;
    const T* ri = H + tri(in ? lane : 0, 0);
#pragma unroll
    for (int k = 0; k < 16; k++) { T v = ri[k < 16 ? k : 0]; r[k] = (in && k <= lane) ? v : (T)0; }
    // make it symmetric-full in registers: r[k] for k>lane = H[k][lane]
#pragma unroll
    for (int k = 1; k < 16; k++) { T v = H[tri(k, 0) + (lane < k ? lane : 0)]; if (lane < k && k < n) r[k] = v; }
    // right-looking LDL^T on full symmetric rows: after step j, r[j] of lanes > j holds L[i][j]
    T dinv_me = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      if (j < n) {
        T dj = bcast(r[j], j);
        if (dj < RPK_MINVAL) { dj = RPK_MINVAL; *warn |= 4; }
        T inv = (T)1 / dj;
        if (lane == j) dinv_me = inv;
        T lij = r[j] * inv;          // L[i][j] for lanes i > j
        // forward-solve the rhs on the fly: x_i -= L[i][j] x_j
        T xj = bcast(x, j);
        if (lane > j) x -= lij * xj;
        // trailing update: r[k] -= L[i][j] * (L[k][j] * dj) = lij * r_k_of_lane... need H[k][j] = bcast(r[j], k)
#pragma unroll
        for (int k = j + 1; k < 16; k++) {
          T hkj = bcast(r[j], k);    // H[k][j] (pre-scale), uniform
          if (lane > j) r[k] -= lij * hkj;
        }
        if (lane > j) r[j] = lij;
      }
    }
    x *= dinv_me;
    // backward: x_j -= sum_{i>j} L[i][j] x_i ; lane i holds L[i][j] in r[j] -> for i = n-1..1: bcast x_i, lane j<i needs L[i][j] = lane i's r[j]: transpose access.
    // use symmetric trick: lane j's r[i] (i>j) was updated as H[j][i] trailing entries, not L. So write L to LDS and do the LDS backward pass.
#pragma unroll
    for (int k = 0; k < 16; k++) if (in && k < lane) H[tri(lane, k)] = r[k];
    WSYNC();
    int p = n - 1;
    for (; p >= 1; p--) { T l0 = H[tri(p, 0) + (lane < p ? lane : 0)]; l0 = lane < p ? l0 : (T)0; x -= l0 * bcast(x, p); }
    return x;
  } else {
    // V3: like V1 but two columns per step (shared LDS reads, one sync per pair)
    T dinv_me = 0;
    int j = 0;
    for (; j + 2 <= n; j += 2) {
      const int j1 = j + 1;
      T rj = (lane < j) ? H[tri(j, 0) + lane] : (T)0;
      T rk = (lane < j) ? H[tri(j1, 0) + lane] : (T)0;
      T dl = (lane < j) ? H[tri(lane, lane)] : (T)0;
      T wj = rj * dl, wk = rk * dl;
      const bool act = lane >= j && lane < n;
      const T* ri = H + tri(act ? lane : 0, 0);
      T s = ri[j], t = ri[j1 <= (act ? lane : 0) ? j1 : 0];
      int p = 0;
      for (; p + 4 <= j; p += 4) {
        T a0 = ri[p], a1 = ri[p + 1], a2 = ri[p + 2], a3 = ri[p + 3];
        s -= a0 * bcast(wj, p); t -= a0 * bcast(wk, p);
        s -= a1 * bcast(wj, p + 1); t -= a1 * bcast(wk, p + 1);
        s -= a2 * bcast(wj, p + 2); t -= a2 * bcast(wk, p + 2);
        s -= a3 * bcast(wj, p + 3); t -= a3 * bcast(wk, p + 3);
      }
      for (; p < j; p++) { T a0 = ri[p]; s -= a0 * bcast(wj, p); t -= a0 * bcast(wk, p); }
      T dj = bcast(s, j);
      if (dj < RPK_MINVAL) { dj = RPK_MINVAL; *warn |= 4; }
      T invj = (T)1 / dj;
      T lij = s * invj;                 // L[i][j], lanes > j
      T lkj = bcast(lij, j1);           // L[j1][j]
      t -= lij * (lkj * dj);            // column j1 after removing column j (lanes >= j1)
      T dk = bcast(t, j1);
      if (dk < RPK_MINVAL) { dk = RPK_MINVAL; *warn |= 4; }
      T invk = (T)1 / dk;
      if (lane == j) { H[tri(j, j)] = dj; dinv_me = invj; }
      else if (act) {
        H[tri(lane, j)] = lij;
        if (lane == j1) { H[tri(j1, j1)] = dk; dinv_me = invk; }
        else H[tri(lane, j1)] = t * invk;
      }
      WSYNC();
    }
    for (; j < n; j++) {
      T rj = (lane < j) ? H[tri(j, 0) + lane] : (T)0;
      T dl = (lane < j) ? H[tri(lane, lane)] : (T)0;
      T wj = rj * dl;
      const bool act = lane >= j && lane < n;
      const T* ri = H + tri(act ? lane : 0, 0);
      T s = ri[j];
      for (int p = 0; p < j; p++) { T a0 = ri[p]; s -= a0 * bcast(wj, p); }
      T dj = bcast(s, j);
      if (dj < RPK_MINVAL) { dj = RPK_MINVAL; *warn |= 4; }
      T inv = (T)1 / dj;
      if (lane == j) { H[tri(j, j)] = dj; dinv_me = inv; }
      else if (act) H[tri(lane, j)] = s * inv;
      WSYNC();
    }
    const bool in = lane < n;
    if (!in) x = 0;
    const T* ri = H + tri(in ? lane : 0, 0);
    int p = 0;
    for (; p + 4 <= n; p += 4) {
      T l0 = ri[p], l1 = ri[p + 1], l2 = ri[p + 2], l3 = ri[p + 3];
      l0 = (in && lane > p) ? l0 : (T)0; l1 = (in && lane > p + 1) ? l1 : (T)0;
      l2 = (in && lane > p + 2) ? l2 : (T)0; l3 = (in && lane > p + 3) ? l3 : (T)0;
      x -= l0 * bcast(x, p); x -= l1 * bcast(x, p + 1); x -= l2 * bcast(x, p + 2); x -= l3 * bcast(x, p + 3);
    }
    for (; p < n; p++) { T l0 = ri[p]; l0 = (in && lane > p) ? l0 : (T)0; x -= l0 * bcast(x, p); }
    x *= dinv_me;
    p = n - 1;
    for (; p - 3 >= 0; p -= 4) {
      T l0 = H[tri(p, 0) + (lane < p ? lane : 0)], l1 = H[tri(p - 1, 0) + (lane < p - 1 ? lane : 0)];
      T l2 = H[tri(p - 2, 0) + (lane < p - 2 ? lane : 0)], l3 = H[tri(p - 3, 0) + (lane < p - 3 ? lane : 0)];
      l0 = lane < p ? l0 : (T)0; l1 = lane < p - 1 ? l1 : (T)0; l2 = lane < p - 2 ? l2 : (T)0; l3 = lane < p - 3 ? l3 : (T)0;
      x -= l0 * bcast(x, p); x -= l1 * bcast(x, p - 1); x -= l2 * bcast(x, p - 2); x -= l3 * bcast(x, p - 3);
    }
    for (; p >= 0; p--) { T l0 = H[tri(p, 0) + (lane < p ? lane : 0)]; l0 = lane < p ? l0 : (T)0; x -= l0 * bcast(x, p); }
    return x;
  }
}
