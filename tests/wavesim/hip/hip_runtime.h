// TEST INFRASTRUCTURE ONLY: stand-in for <hip/hip_runtime.h> when the engine's sources are compiled
// for the CPU wave emulator (tests/wavesim).  Device memory is host memory, streams and events are
// no-ops (everything is synchronous), kernels run through wavesim::launch.
#pragma once
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>

#include "../wavesim.hpp"

#define RP_WAVESIM 1
#define RPK_SREG_CONSTRAINT "+r"
#define RPK_CONST_AS

// ---- language
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define threadIdx (wavesim::ctx().threadIdx)
#define blockIdx (wavesim::ctx().blockIdx)
#define blockDim (wavesim::ctx().blockDim)
#define gridDim (wavesim::ctx().gridDim)
using dim3 = wavesim::Dim3;
struct int4 { int x, y, z, w; };
struct int2 { int x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };

using std::max;
using std::min;

// ---- cross-lane operations (each is one rendezvous of the wave; the call site identifies the group)
#define WS_SITE __builtin_return_address(0)
__attribute__((noinline)) static int ws_readlane(int v, int l) {
  return (int)wavesim::collective(wavesim::K_READLANE, WS_SITE, (uint32_t)v, (uint32_t)l);
}
__attribute__((noinline)) static int ws_readfirstlane(int v) {
  return (int)wavesim::collective(wavesim::K_READFIRSTLANE, WS_SITE, (uint32_t)v, 0);
}
__attribute__((noinline)) static int ws_update_dpp(int old, int v, int ctrl, int rmask, int bmask, bool bound) {
  return (int)wavesim::collective(wavesim::K_DPP, WS_SITE, (uint32_t)v,
                                  (uint32_t)ctrl | ((uint32_t)rmask << 16) | ((uint32_t)bmask << 20) | ((uint32_t)bound << 24),
                                  (uint32_t)old);
}
__attribute__((noinline)) static unsigned long long ws_ballot(bool p) {
  return wavesim::collective(wavesim::K_BALLOT, WS_SITE, p ? 1u : 0u, 0);
}
__attribute__((noinline)) static void ws_wave_barrier() { wavesim::collective(wavesim::K_BARRIER, WS_SITE, 0, 0); }
__attribute__((noinline)) static void ws_syncthreads() { wavesim::collective(wavesim::K_SYNCTHREADS, WS_SITE, 0, 0); }
__attribute__((noinline)) static int ws_bpermute(int byte_addr, int v) {
  return (int)wavesim::collective(wavesim::K_BPERMUTE, WS_SITE, (uint32_t)v, (uint32_t)(byte_addr >> 2) & 63u);
}
__attribute__((noinline)) static int ws_permute(int byte_addr, int v) {
  return (int)wavesim::collective(wavesim::K_PERMUTE, WS_SITE, (uint32_t)v, (uint32_t)(byte_addr >> 2) & 63u);
}
#define __builtin_amdgcn_readlane(v, l) ws_readlane((v), (l))
#define __builtin_amdgcn_readfirstlane(v) ws_readfirstlane((v))
#define __builtin_amdgcn_update_dpp(old, v, ctrl, rm, bm, bc) ws_update_dpp((old), (v), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_mov_dpp(v, ctrl, rm, bm, bc) ws_update_dpp(0, (v), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_ballot_w64(p) ws_ballot((p))
#define __builtin_amdgcn_ds_bpermute(a, v) ws_bpermute((a), (v))
#define __builtin_amdgcn_ds_permute(a, v) ws_permute((a), (v))
#define __ballot(p) ws_ballot((p))
#define __builtin_amdgcn_wave_barrier() ws_wave_barrier()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_barrier() ws_syncthreads()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __syncthreads() ws_syncthreads()
template <typename V> static inline V ws_shfl(V v, int src) {
  static_assert(sizeof(V) == 4 || sizeof(V) == 8, "shuffle width");
  const int lane = (int)(threadIdx.x & 63);
  (void)lane;
  if constexpr (sizeof(V) == 4) {
    int b; memcpy(&b, &v, 4);
    b = ws_bpermute(src << 2, b);
    V r; memcpy(&r, &b, 4); return r;
  } else {
    int b[2]; memcpy(b, &v, 8);
    b[0] = ws_bpermute(src << 2, b[0]); b[1] = ws_bpermute(src << 2, b[1]);
    V r; memcpy(&r, b, 8); return r;
  }
}
template <typename V> static inline V __shfl_xor(V v, int mask, int width = 64) { (void)width; return ws_shfl(v, (int)((threadIdx.x & 63) ^ mask)); }
template <typename V> static inline V __shfl_up(V v, int d, int width = 64) {
  (void)width; const int l = (int)(threadIdx.x & 63);
  const V o = ws_shfl(v, l - d >= 0 ? l - d : l);
  return l - d >= 0 ? o : v;
}
template <typename V> static inline V __shfl(V v, int src, int width = 64) { (void)width; return ws_shfl(v, src & 63); }

// ---- scalar builtins
#define __builtin_amdgcn_rsq(x) (1.0 / ::sqrt((double)(x)))
#define __builtin_amdgcn_rsqf(x) (1.0f / ::sqrtf((float)(x)))
#define __builtin_amdgcn_rcp(x) (1.0 / (double)(x))
#define __builtin_amdgcn_rcpf(x) (1.0f / (float)(x))
#define __builtin_readcyclecounter() (__builtin_ia32_rdtsc())
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
// within a workgroup lanes run one at a time: plain read-modify-write; across workgroups (OS threads)
// real atomics
#define __hip_atomic_fetch_add(p, v, order, scope) ws_fetch_add((p), (v))
template <typename V> static inline V ws_fetch_add(V* p, V v) { const V o = *p; *p = o + v; return o; }
static inline float __int_as_float(int v) { float f; __builtin_memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; __builtin_memcpy(&v, &f, 4); return v; }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline int atomicMax(int* p, int v) {
  int o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return o;
}

// ---- runtime API (synchronous host memory)
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
typedef struct ws_stream_* hipStream_t;
typedef struct ws_event_* hipEvent_t;
typedef struct ws_graph_* hipGraph_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipMemoryType { hipMemoryTypeHost = 0, hipMemoryTypeDevice = 1 };
struct hipPointerAttribute_t { hipMemoryType type; };
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline const char* hipGetErrorString(hipError_t) { return "wavesim error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = ::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipFree(void* p) { ::free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = ::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipHostFree(void* p) { ::free(p); return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { ::memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { ::memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { ::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { ::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)::malloc(8); return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)::malloc(8); return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { ::free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* c) { *c = hipStreamCaptureStatusNone; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)::malloc(8); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)::malloc(8); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { ::free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { a->type = hipMemoryTypeHost; return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  wavesim::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })
