// microbenchmark: cost of fire-and-forget LDS fp64 adds (ds_add_f64) as a function of how many lanes hit the same
// address; and of ds_bpermute / wave_shr DPP moves of an fp64 value.  One wave per SIMD and two.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(long long* out, double* sink, int n, int nways) {
  __shared__ double a[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) a[i] = 0;
  __syncthreads();
  long long t[8];
  double x = lane * 0.5 + 1.0;
  // 0: all lanes distinct addresses
  t[0] = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) __hip_atomic_fetch_add(&a[lane + 64 * u], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  __builtin_amdgcn_s_waitcnt(0);
  t[1] = __builtin_readcyclecounter();
  // 1: groups of `nways` lanes share an address
#pragma unroll 1
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) __hip_atomic_fetch_add(&a[(lane / nways) + 64 * u], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  __builtin_amdgcn_s_waitcnt(0);
  t[2] = __builtin_readcyclecounter();
  // 2: all 64 lanes one address
#pragma unroll 1
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) __hip_atomic_fetch_add(&a[64 * u], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  __builtin_amdgcn_s_waitcnt(0);
  t[3] = __builtin_readcyclecounter();
  // 3: plain ds_write_b64 to distinct addresses (reference)
#pragma unroll 1
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) { a[lane + 64 * u] = x; asm volatile("" ::: "memory"); }
  }
  __builtin_amdgcn_s_waitcnt(0);
  t[4] = __builtin_readcyclecounter();
  // 4: ds_read_b64, all lanes same address (broadcast), dependent chain
  double y = 0;
#pragma unroll 1
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) { y += a[64 * u + ((int)y & 1)]; }
  }
  t[5] = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) for (int i = 0; i < 5; i++) out[i] = (t[i + 1] - t[i]);
  if (y == -12345.0) sink[lane] = y + a[lane];
}
int main() {
  long long* dout; double* dsink;
  hipMalloc(&dout, 16 * 8); hipMalloc(&dsink, 4096 * 8);
  const int n = 64;
  for (int nways : {2, 4, 8, 22, 32}) {
    for (int waves : {1, 2}) {
      for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(1024 * waves), dim3(64), 0, 0, dout, dsink, n, nways); hipDeviceSynchronize(); }
      long long h[8]; hipMemcpy(h, dout, 64, hipMemcpyDeviceToHost);
      printf("nways %2d blocks/SIMD %d: cycles per ds_add_f64 instruction: distinct %.1f  %d-way %.1f  64-way %.1f | ds_write_b64 %.1f | dependent broadcast read+add %.1f\n",
             nways, waves, h[0] / (16.0 * n), nways, h[1] / (16.0 * n), h[2] / (16.0 * n), h[3] / (16.0 * n), h[4] / (16.0 * n));
    }
  }
  return 0;
}
