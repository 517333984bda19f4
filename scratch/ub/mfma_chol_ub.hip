// microbenchmark: does v_mfma_f64_16x16x4_f64 pay on the solver's dense cross-chain block?
//   production: rpk::dense_factor_solve (rp_dense.hpp) -- packed L L^T in LDS, lane = row, two columns per step, inner
//               products over ALL previous columns as LDS broadcast reads;
//   MFMA:       the same factorisation in panels of 16 columns: the contribution of all PREVIOUS panels to a panel is one
//               batch of 16x16x4 MFMA tiles (A = rows of L, B = the panel's rows of L, both gathered from the packed
//               LDS matrix), the inner products inside the panel (<= 15 terms) stay lane = row; same backward pass.
// n = rows of the block (the solver's blocks: 17 on average, 34-39 at p99.9, 57 at the capacity), + the rhs row.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on mfma_chol_ub.hip -o mfma_chol_ub
#include "../../robopianist_amd/csrc/rp_kernels.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace rpk;
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ double dense_factor_solve_mfma(double* H, int n, int lane, int* warn) {
  using T = double;
  const int N = n + 1;   // rows incl. the rhs row
  T invd_me = 0;
  for (int j0 = 0; j0 < n; j0 += 16) {
    const int jend = j0 + 16 < n ? j0 + 16 : n;
    if (j0 > 0) {
      // ---- all previous panels' contribution to this panel, row tile by row tile
      const int c = lane & 15, k = lane >> 4;
      for (int r0 = j0; r0 < N; r0 += 16) {
        d4 acc = {0, 0, 0, 0};
        const int ra = r0 + c, rb = j0 + c;
        const T* pa = H + tri(ra < N ? ra : 0, 0);
        const T* pb = H + tri(rb < jend ? rb : 0, 0);
        const bool va = ra < N, vb = rb < jend;
        for (int p0 = 0; p0 < j0; p0 += 4) {
          const T a = va ? pa[p0 + k] : 0.0, b = vb ? pb[p0 + k] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int row = r0 + k + 4 * q, col = j0 + c;
          if (row < N && col < jend && col <= row) H[tri(row, col)] -= acc[q];
        }
      }
      WSYNC();
    }
    // ---- the panel itself: lane = row, two columns per step, inner products over the panel's own columns only
    int j = j0;
    for (; j + 2 <= jend; j += 2) {
      const int j1 = j + 1;
      const bool act = lane >= j && lane <= n;
      const T* ri = H + tri(act ? lane : 0, 0);
      const T* rj = H + tri(j, 0);
      const T* rk = H + tri(j1, 0);
      T s = ri[j], t = ri[j1];
      for (int p = j0; p < j; p++) { const T a0 = ri[p]; s -= a0 * rj[p]; t -= a0 * rk[p]; }
      T dj = bcast(s, j);
      if (!(dj >= RPK_MINVAL)) { dj = RPK_MINVAL; *warn |= 4; }
      const T rs = rsqrt_nr(dj);
      const T lij = s * rs;
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
    if (j < jend) {
      const bool act = lane >= j && lane <= n;
      const T* ri = H + tri(act ? lane : 0, 0);
      const T* rj = H + tri(j, 0);
      T s = ri[j];
      for (int p = j0; p < j; p++) s -= ri[p] * rj[p];
      T dj = bcast(s, j);
      if (!(dj >= RPK_MINVAL)) { dj = RPK_MINVAL; *warn |= 4; }
      const T rs = rsqrt_nr(dj);
      if (lane == j) invd_me = rs;
      if (act) H[tri(lane, j)] = s * rs;
      WSYNC();
    }
  }
  // backward pass L^T x = y (row n holds y), as in production
  T x = lane < n ? H[tri(n, 0) + lane] : (T)0;
  for (int p = n - 1; p >= 0; p--) {
    T l0 = H[tri(p, 0) + (lane < p ? lane : 0)];
    l0 = lane < p ? l0 : (T)0;
    if (lane == p) x *= invd_me;
    x -= l0 * bcast(x, p);
  }
  return x;
}

template <int V>
__global__ __launch_bounds__(64) void kb(const double* A, const double* b, double* xout, long long* cyc, int n, int reps) {
  __shared__ double H[(RPK_HMAX + 1) * (RPK_HMAX + 2) / 2 + 8];
  const int lane = threadIdx.x;
  int warn = 0;
  long long tot = 0;
  double x = 0;
  for (int r = 0; r < reps; r++) {
    for (int i = lane; i < n * (n + 1) / 2; i += 64) H[i] = A[i];
    x = lane < n ? b[lane] : 0.0;
    WSYNC();
    if (lane < n) H[tri(n, 0) + lane] = x;
    WSYNC();
    const long long t0 = (long long)__builtin_readcyclecounter();
    x = V == 0 ? dense_factor_solve<double>(H, n, lane, &warn) : dense_factor_solve_mfma(H, n, lane, &warn);
    WSYNC();
    tot += (long long)__builtin_readcyclecounter() - t0;
  }
  if (lane < n) xout[blockIdx.x * 64 + lane] = x;
  if (lane == 0) cyc[blockIdx.x] = tot / reps;
}

template <int V> double run(int n, int NB, const char* name) {
  std::vector<double> A(n * (n + 1) / 2), b(n), B(n * n);
  srand(1);
  for (auto& v : B) v = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) {
    double s = i == j ? 1.0 : 0.0;
    for (int k = 0; k < n; k++) s += B[i * n + k] * B[j * n + k];
    A[i * (i + 1) / 2 + j] = s;
  }
  for (int i = 0; i < n; i++) b[i] = i + 1;
  double *dA, *db, *dx; long long* dc;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&db, n * 8); hipMalloc(&dx, NB * 64 * 8); hipMalloc(&dc, NB * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((kb<V>), dim3(NB), dim3(64), 0, 0, dA, db, dx, dc, n, 20);
  hipDeviceSynchronize();
  std::vector<long long> c(NB); std::vector<double> x(64);
  hipMemcpy(c.data(), dc, NB * 8, hipMemcpyDeviceToHost); hipMemcpy(x.data(), dx, 64 * 8, hipMemcpyDeviceToHost);
  double res = 0;
  for (int i = 0; i < n; i++) {
    double s = 0;
    for (int j = 0; j < n; j++) s += A[i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i] * x[j];
    res = fmax(res, fabs(s - b[i]));
  }
  double avg = 0; for (auto v : c) avg += v; avg /= NB;
  printf("%-34s n=%2d waves/SIMD=%d  %8.0f cycles  resid %.2e\n", name, n, NB >= 2048 ? 2 : 1, avg, res);
  hipFree(dA); hipFree(db); hipFree(dx); hipFree(dc);
  return avg;
}

int main() {
  for (int NB : {1024, 2048}) for (int n : {12, 17, 24, 36, 48, 57}) {
    const double a = run<0>(n, NB, "production (lane = row, 2 columns)");
    const double m = run<1>(n, NB, "MFMA f64 16x16x4 panel updates");
    printf("   -> MFMA / production = %.2f\n", m / a);
  }
  return 0;
}
