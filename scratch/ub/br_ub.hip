#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(long long* out, double* sink, int n, double seed, int zero) {
  const int lane = threadIdx.x;
  long long t[16];
  // 0: rolled fma chain
  double a = seed * lane;
  t[0] = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < n; i++) a = a * 1.0000001 + 0.5;
  t[1] = __builtin_readcyclecounter();
  // 1: unrolled x16 fma chain
  double b = seed * lane;
#pragma unroll 1
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int u = 0; u < 16; u++) b = b * 1.0000001 + 0.5;
  }
  t[2] = __builtin_readcyclecounter();
  // 2: 16 independent fmas per iteration (throughput)
  double c[16];
#pragma unroll
  for (int u = 0; u < 16; u++) c[u] = seed + u;
#pragma unroll 1
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int u = 0; u < 16; u++) c[u] = c[u] * 1.0000001 + 0.5;
  }
  t[3] = __builtin_readcyclecounter();
  // 3: divergent if (exec mask branch, not taken path skipping) inside unrolled code
  double d = seed;
#pragma unroll 1
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int u = 0; u < 16; u++) { if ((lane + zero) > 70 + u) d = d * 1.01 + 0.5; else d += 1.0; }
  }
  t[4] = __builtin_readcyclecounter();
  // 4: uniform branch skip (scalar condition false -> jump over block)
  double e = seed;
#pragma unroll 1
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int u = 0; u < 16; u++) { if (zero > u) { e = e * 1.01 + 0.5; e = sqrt(e); } e += 1.0; }
  }
  t[5] = __builtin_readcyclecounter();
  // 5: 16 float independent fmas
  float f[16];
#pragma unroll
  for (int u = 0; u < 16; u++) f[u] = (float)seed + u;
#pragma unroll 1
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int u = 0; u < 16; u++) f[u] = f[u] * 1.0000001f + 0.5f;
  }
  t[6] = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) for (int i = 0; i < 6; i++) out[i] = (t[i + 1] - t[i]);
  double s = a + b + d + e;
#pragma unroll
  for (int u = 0; u < 16; u++) s += c[u] + f[u];
  if (s == -12345.0) sink[lane] = s;
}
int main() {
  long long* dout; double* dsink;
  hipMalloc(&dout, 16 * 8); hipMalloc(&dsink, 4096 * 8);
  const int n = 1024;
  for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, dout, dsink, n, 1.25, 0); hipDeviceSynchronize(); }
  long long o[16]; hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  const char* names[] = {"rolled fp64 fma chain (1/iter)", "unrolled x16 dependent fp64 fma", "16 independent fp64 fma", "divergent if/else x16 (exec)", "uniform skip branch x16", "16 independent fp32 fma"};
  for (int i = 0; i < 6; i++) printf("%-36s %7.2f cycles per op\n", names[i], (double)o[i] / n);
  return 0;
}
