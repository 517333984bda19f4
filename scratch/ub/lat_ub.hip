// primitive latencies on one wave per SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
__device__ __forceinline__ double bcastd(double v, int l) {
  long long b = __builtin_bit_cast(long long, v);
  int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  lo = __builtin_amdgcn_readlane(lo, l); hi = __builtin_amdgcn_readlane(hi, l);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__global__ __launch_bounds__(64) void k(long long* out, double* sink, int n, double seed) {
  __shared__ double L[4096];
  __shared__ int I[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) L[i] = seed + i;
  for (int i = lane; i < 1024; i += 64) I[i] = (i * 7 + 3) & 1023;
  WSYNC();
  long long t[16];
  double acc = seed;
  int idx = lane;
  // 0: dependent LDS read chain (b32 index chase)
  t[0] = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) idx = I[idx];
  t[1] = __builtin_readcyclecounter();
  // 1: dependent fp64 fma chain
  double a = seed * lane;
  for (int i = 0; i < n; i++) a = a * 1.0000001 + 0.5;
  t[2] = __builtin_readcyclecounter();
  // 2: readlane (dynamic lane) + fma chain
  double s = seed;
  for (int i = 0; i < n; i++) s -= a * bcastd(a + s, i & 63);
  t[3] = __builtin_readcyclecounter();
  // 3: fp64 divide chain
  double d = seed + 2.0;
  for (int i = 0; i < n; i++) d = 1.0 / (d + 1.5);
  t[4] = __builtin_readcyclecounter();
  // 4: fp64 sqrt chain
  double q = seed + 2.0;
  for (int i = 0; i < n; i++) q = sqrt(q + 1.5);
  t[5] = __builtin_readcyclecounter();
  // 5: LDS write -> sync -> read roundtrip (b64)
  double w = seed;
  for (int i = 0; i < n; i++) { L[lane + (i & 7) * 64] = w; WSYNC(); w = L[((lane + 1) & 63) + (i & 7) * 64] + 1.0; WSYNC(); }
  t[6] = __builtin_readcyclecounter();
  // 6: 8 independent LDS b64 reads then sum
  double r8 = 0;
  for (int i = 0; i < n; i++) {
    const double* p = L + ((idx + i) & 1023);
    r8 += p[0] + p[64] + p[128] + p[192] + p[256] + p[320] + p[384] + p[448];
  }
  t[7] = __builtin_readcyclecounter();
  // 7: readlane with constant lane + fma
  double s2 = seed;
  for (int i = 0; i < n; i++) s2 -= a * bcastd(a + s2, 5);
  t[8] = __builtin_readcyclecounter();
  // 8: fp32 fma chain
  float f = (float)seed * lane;
  for (int i = 0; i < n; i++) f = f * 1.0000001f + 0.5f;
  t[9] = __builtin_readcyclecounter();
  // 9: s_memtime overhead itself
  for (int i = 0; i < n; i++) { acc += (double)(__builtin_readcyclecounter() & 1); }
  t[10] = __builtin_readcyclecounter();
  // 10: LDS atomic add f64 (no return), then one sync
  for (int i = 0; i < n; i++) __hip_atomic_fetch_add(&L[lane + (i & 7) * 64], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  WSYNC();
  t[11] = __builtin_readcyclecounter();
  // 11: global load chain (L2 hit)
  int gi = lane;
  const int* gI = (const int*)sink;  // reuse buffer as index table (host fills)
  for (int i = 0; i < n; i++) gi = gI[gi & 1023];
  t[12] = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) for (int i = 0; i < 12; i++) out[i] = (t[i + 1] - t[i]);
  if (idx + gi == -12345) sink[2000 + lane] = acc + a + s + d + q + w + r8 + s2 + f;
}
int main() {
  long long* dout; double* dsink;
  hipMalloc(&dout, 16 * 8); hipMalloc(&dsink, 4096 * 8);
  int h[2048]; for (int i = 0; i < 2048; i++) h[i] = (i * 13 + 5) & 1023;
  hipMemcpy(dsink, h, sizeof(h), hipMemcpyHostToDevice);
  const int n = 256;
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, dout, dsink, n, 1.25);
    hipDeviceSynchronize();
  }
  long long o[16]; hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  const char* names[] = {"LDS b32 dependent read", "fp64 fma dependent", "readlane(dyn)x2 + add + fma f64", "fp64 divide dependent",
    "fp64 sqrt dependent", "LDS write->sync->read->sync", "8 indep LDS b64 reads + adds", "readlane(const)x2 + add + fma", "fp32 fma dependent",
    "s_memtime", "LDS atomic add f64 (no ret)", "global load chain (L2)"};
  for (int i = 0; i < 12; i++) printf("%-36s %7.1f cycles/iter\n", names[i], (double)o[i] / n);
  return 0;
}
