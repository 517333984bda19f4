// microbenchmark: dense packed Cholesky + solve variants, one wave per block
#include "../../robopianist_amd/csrc/rp_kernels.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace rpk;

#include "chol_variants.hpp"

template <typename T, int V>
__global__ __launch_bounds__(64) void kb(const T* A, const T* b, T* xout, long long* cyc, int n, int reps) {
  __shared__ T H[(RPK_HMAX + 1) * (RPK_HMAX + 2) / 2 + 8];
  __shared__ T pad[2500];  // mimic the LDS footprint (occupancy 4 WG/CU for fp64)
  const int lane = threadIdx.x;
  int warn = 0;
  pad[lane] = 0;
  long long tot = 0;
  T x = 0;
  for (int r = 0; r < reps; r++) {
    for (int i = lane; i < n * (n + 1) / 2; i += 64) H[i] = A[i];
    x = lane < n ? b[lane] : (T)0;
    WSYNC();
    long long t0 = (long long)__builtin_readcyclecounter();
    if (V == 0) { if (lane < n) H[tri(n, 0) + lane] = x; WSYNC(); x = dense_factor_solve(H, n, lane, &warn); }
    else x = chol_solve_variant<T, V>(H, n, lane, x, &warn);
    WSYNC();
    long long t1 = (long long)__builtin_readcyclecounter();
    tot += t1 - t0;
  }
  if (lane < n) xout[blockIdx.x * 64 + lane] = x + pad[lane];
  if (lane == 0) cyc[blockIdx.x] = tot / reps;
}

template <typename T, int V>
void run(int n, const char* name) {
  std::vector<T> A(n * (n + 1) / 2), b(n), B(n * n);
  srand(1);
  for (auto& v : B) v = (T)(rand() / (double)RAND_MAX - 0.5);
  for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) {
    double s = i == j ? 1.0 : 0.0;
    for (int k = 0; k < n; k++) s += B[i * n + k] * B[j * n + k];
    A[i * (i + 1) / 2 + j] = (T)s;
  }
  for (int i = 0; i < n; i++) b[i] = (T)(i + 1);
  T *dA, *db, *dx; long long* dc;
  const int NB = 1024;
  hipMalloc(&dA, A.size() * sizeof(T)); hipMalloc(&db, n * sizeof(T));
  hipMalloc(&dx, NB * 64 * sizeof(T)); hipMalloc(&dc, NB * sizeof(long long));
  hipMemcpy(dA, A.data(), A.size() * sizeof(T), hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), n * sizeof(T), hipMemcpyHostToDevice);
  hipLaunchKernelGGL((kb<T, V>), dim3(NB), dim3(64), 0, 0, dA, db, dx, dc, n, 20);
  hipDeviceSynchronize();
  std::vector<long long> c(NB); std::vector<T> x(64);
  hipMemcpy(c.data(), dc, NB * sizeof(long long), hipMemcpyDeviceToHost);
  hipMemcpy(x.data(), dx, 64 * sizeof(T), hipMemcpyDeviceToHost);
  // residual
  double res = 0;
  for (int i = 0; i < n; i++) {
    double s = 0;
    for (int j = 0; j < n; j++) s += (double)A[i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i] * x[j];
    res = fmax(res, fabs(s - b[i]));
  }
  double avg = 0; for (auto v : c) avg += v; avg /= NB;
  printf("%-28s n=%2d  %8.0f cycles  (%.0f /col)  resid %.2e\n", name, n, avg, avg / n, res);
  hipFree(dA); hipFree(db); hipFree(dx); hipFree(dc);
}

int main() {
  for (int n : {9, 13, 16, 20, 24, 28, 34}) {
    run<double, 0>(n, "f64 production (factor+solve)");
    run<double, 6>(n, "f64 2-col factor only");
    run<double, 8>(n, "f64 v8 4-col factor only");
  }
  return 0;
}
