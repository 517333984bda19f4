#include <hip/hip_runtime.h>
#include <cstdio>
#define NOPS4(x) asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1" : "+v"(x))
#define BLK16(x) NOPS4(x); NOPS4(x); NOPS4(x); NOPS4(x)
#define BLK64(x) BLK16(x); BLK16(x); BLK16(x); BLK16(x)
__global__ __launch_bounds__(64) void k(long long* out, int* sink, int n, int zero) {
  const int lane = threadIdx.x;
  long long t[16];
  int x = lane;
  // 0: baseline: 16 x (4 adds) straight
  t[0] = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int u = 0; u < 16; u++) { NOPS4(x); }
  }
  t[1] = __builtin_readcyclecounter();
  // 1: uniform skip (taken) over 4-instr block, + 4 adds executed
#pragma unroll 1
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int u = 0; u < 16; u++) { if (zero > u) { NOPS4(x); } NOPS4(x); }
  }
  t[2] = __builtin_readcyclecounter();
  // 2: uniform skip (taken) over 64-instr block
#pragma unroll 1
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int u = 0; u < 16; u++) { if (zero > u) { BLK64(x); } NOPS4(x); }
  }
  t[3] = __builtin_readcyclecounter();
  // 3: uniform branch NOT taken (cond true): executes 4-instr block
#pragma unroll 1
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int u = 0; u < 16; u++) { if (zero < u + 1) { NOPS4(x); } NOPS4(x); }
  }
  t[4] = __builtin_readcyclecounter();
  // 4: divergent block, no lane active (execz skip taken), 64-instr block
#pragma unroll 1
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int u = 0; u < 16; u++) { if (lane + zero > 100 + u) { BLK64(x); } NOPS4(x); }
  }
  t[5] = __builtin_readcyclecounter();
  // 5: divergent block, some lanes active (not taken), 4-instr block
#pragma unroll 1
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int u = 0; u < 16; u++) { if (lane + zero > 10 + u) { NOPS4(x); } NOPS4(x); }
  }
  t[6] = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) for (int i = 0; i < 6; i++) out[i] = (t[i + 1] - t[i]);
  if (x == -12345) sink[lane] = x;
}
int main() {
  long long* dout; int* dsink;
  hipMalloc(&dout, 16 * 8); hipMalloc(&dsink, 4096 * 8);
  const int n = 1024;
  for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, dout, dsink, n, 0); hipDeviceSynchronize(); }
  long long o[16]; hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  const char* names[] = {"baseline 4 adds", "taken uniform skip over 4 + 4 adds", "taken uniform skip over 64 + 4 adds", "not-taken uniform + 8 adds", "execz skip taken over 64 + 4 adds", "exec branch not taken + 8 adds"};
  for (int i = 0; i < 6; i++) printf("%-40s %7.2f cycles per unit\n", names[i], (double)o[i] / n);
  return 0;
}
